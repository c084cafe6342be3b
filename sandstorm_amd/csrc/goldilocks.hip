// goldilocks.hip — NTT / iNTT / LDE and FRI folding for the 64-bit field p = 2^64 - 2^32 + 1 on gfx950.
//
// The field the reference instantiates for its experimental claim (cli/src/main.rs:103-133:
// ministark_gpu::fields::p18446744069414584321::ark::{Fp, Fq3}; rows N1/N2/F1 of SURVEY.md 8a for BASELINE.json
// configs[4], "next" row X4).  Same conventions as the 252-bit path (ntt.hip): forward = evaluation at offset * w^k in
// natural order, w = 7^((p-1)/n); forward is a decimation-in-time network (bit-reversed in, natural out), inverse the
// exact inverse network (natural in, bit-reversed out), so an LDE needs no transpose; the coset offset is folded into the
// per-stage twiddle plan; 1/n rides on the last inverse pass.
//
// An element is 8 bytes, so unlike the 252-bit kernels this one is bound by HBM, not by the multiplier: a pass streams
// the vector once (16 B per element) and runs up to 13 stages on an 8192-element tile in LDS (64 KiB; two workgroups per
// CU) in register groups of four stages; a 2^25-point transform is 13 + 8 + 4 stages = three passes (strided passes keep
// >= 32 adjacent elements per row: 256-byte global runs).  Multiplication: 64 x 64 -> 128 by four 32-bit
// multiply-adds, then 2^64 = 2^32 - 1 and 2^96 = -1 (mod p) fold the high half back: no division, no Montgomery form.
// Every operation here is linear in the data (data times coefficients this file owns), so element images in Montgomery
// form (arkworks' Fp64 in memory) pass through unchanged.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "gl64.h"
#include "gl_ntt.h"

namespace ss {


uint64_t gl_pow_host(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) { if (e & 1) r = gl_mul(r, a); a = gl_mul(a, a); e >>= 1; }
    return r;
}
uint64_t gl_root_of_unity_host(uint32_t log_n) { return gl_pow_host(7, (GL_P - 1) >> log_n); }
uint64_t gl_inv_host(uint64_t a) { return gl_pow_host(a, GL_P - 2); }

struct GlCols { const uint64_t *src[MAX_COLS]; uint64_t *dst[MAX_COLS]; };

template <bool DIF, bool CONTIG>
__global__ __launch_bounds__(512) void gl_ntt_pass_kernel(GlCols cols, const uint64_t *__restrict__ tw, GlPassParams p) {
    extern __shared__ uint64_t gl_tile[];
    const uint32_t tile = blockIdx.x;
    const void *src_v = cols.src[0];
    void *dst_v = cols.dst[0];
#pragma unroll
    for (int c = 1; c < MAX_COLS; ++c)
        if (blockIdx.y == (unsigned)c) { src_v = cols.src[c]; dst_v = cols.dst[c]; }
    const uint64_t *__restrict__ src = reinterpret_cast<const uint64_t *>(src_v);
    uint64_t *__restrict__ dst = reinterpret_cast<uint64_t *>(dst_v);
    const uint32_t first = DIF ? 0u : p.u_first, total = p.r - first;
    uint32_t done = 0;
    while (done < total) {
        const uint32_t g = (total - done) >= 4 ? 4u : (total - done);
        const uint32_t u = DIF ? (p.r - done - g) : (first + done);
        const bool fg = done == 0, tg = done + g == total;
        switch (g) {
        case 4: gl_group_dispatch<DIF, 4, CONTIG>(gl_tile, tw, p, u, tile, fg, tg, src, dst); break;
        case 3: gl_group_dispatch<DIF, 3, CONTIG>(gl_tile, tw, p, u, tile, fg, tg, src, dst); break;
        case 2: gl_group_dispatch<DIF, 2, CONTIG>(gl_tile, tw, p, u, tile, fg, tg, src, dst); break;
        default: gl_group_dispatch<DIF, 1, CONTIG>(gl_tile, tw, p, u, tile, fg, tg, src, dst); break;
        }
        done += g;
        if (!tg) __syncthreads();
    }
    if (total == 0)
        for (uint32_t e = threadIdx.x; e < (1u << p.log_tile); e += blockDim.x) dst[gl_tile_gindex(p, tile, e)] = src[gl_tile_gindex(p, tile, e) >> p.log_expand];
}

// plan: T_s[k] = hpow[s] * r^(k n / 2^(s+1)) with r^e = pow_lo[e & 4095] * pow_hi[e >> 12]
__global__ void gl_twiddle_kernel(uint64_t *__restrict__ tw, const uint64_t *__restrict__ pow_lo, const uint64_t *__restrict__ pow_hi,
                                  const uint64_t *__restrict__ hpow, uint32_t log_n) {
    const uint64_t total = (1ull << log_n) - 1ull;
    for (uint64_t idx = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t s = 63u - (uint32_t)__clzll(idx + 1ull);
        const uint64_t k = idx + 1ull - (1ull << s);
        const uint64_t e = k << (log_n - 1u - s);
        tw[idx] = gl_mul(gl_mul(pow_lo[e & 4095ull], pow_hi[e >> 12]), hpow[s]);
    }
}

__global__ void gl_bitrev_copy_kernel(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst, uint32_t log_n) {
    const uint64_t n = 1ull << log_n;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        dst[__brevll(i) >> (64u - log_n)] = src[i];
}

// ---- Fq3 = Fp[X] / (X^3 - 2) and one FRI layer over Fq3-valued evaluations ------------------------------------------
struct Gl3FriConsts {
    uint64_t winv[16];           // w_fold^-k
    uint64_t inv_fold;           // 1 / fold (or 1 when unnormalised)
    Gl3 alpha;
    uint64_t offset_inv, wl_inv; // 1 / offset, 1 / w_L
};
// out[j] = (interpolant of {evals[j + k rows]}_k over x_j <w_fold>)(alpha): per component a size-fold inverse DFT
// (direct O(fold^2): fold <= 16), coefficient k scaled by x_j^-k, Horner in alpha
template <int LOG_FOLD>
__global__ __launch_bounds__(256) void gl3_fri_fold_kernel(const uint64_t *__restrict__ evals, uint32_t log_len, Gl3FriConsts k, uint64_t *__restrict__ out) {
    constexpr int FOLD = 1 << LOG_FOLD;
    const uint64_t rows = 1ull << (log_len - LOG_FOLD);
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < rows; j += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t v[3][FOLD];
#pragma unroll
        for (int m = 0; m < FOLD; ++m)
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c][m] = evals[3 * (j + (uint64_t)m * rows) + c];
        // x_j^-1 = offset^-1 * w_L^-j
        uint64_t xinv = k.offset_inv, wp = k.wl_inv;
        for (uint64_t e = j; e; e >>= 1) { if (e & 1) xinv = gl_mul(xinv, wp); wp = gl_mul(wp, wp); }
        Gl3 acc = {{0, 0, 0}};
        uint64_t xk = k.inv_fold;                              // x_j^-t / fold, t = FOLD-1 .. 0 handled by Horner from the top
        uint64_t xpow[FOLD];
        xpow[0] = k.inv_fold;
#pragma unroll
        for (int t = 1; t < FOLD; ++t) xpow[t] = gl_mul(xpow[t - 1], xinv);
        (void)xk;
#pragma unroll
        for (int t = FOLD - 1; t >= 0; --t) {
            Gl3 coef;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                GlWide s = glw_zero();                       // one reduction for the FOLD products (gl64.h)
#pragma unroll
                for (int m = 0; m < FOLD; ++m) glw_mac(s, v[c][m], k.winv[(t * m) & (FOLD - 1)]);
                coef.c[c] = gl_mul(glw_reduce(s), xpow[t]);
            }
            acc = gl3_mul(acc, k.alpha);
#pragma unroll
            for (int c = 0; c < 3; ++c) acc.c[c] = gl_add(acc.c[c], coef.c[c]);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) out[3 * j + c] = acc.c[c];
    }
}

// ------------------------------------------------------------------------------------------------ host launch
static constexpr uint32_t GL_LOG_TILE_MAX = 13;             // 8192 elements = 64 KiB of LDS: two workgroups per CU
// A/B knobs (read once): SS_GL_LOG_TILE = 14 (128 KiB: one workgroup per CU) with SS_GL_MIN_RUN_LOG = 3 (64-byte runs in the strided
// pass) makes a 2^24 / 2^25-point transform TWO passes instead of three; SS_GL_THREADS = lanes per workgroup
static uint32_t gl_env_u32(const char *name, uint32_t dflt, uint32_t lo, uint32_t hi) {
    const char *e = getenv(name);
    if (!e) return dflt;
    const unsigned long v = strtoul(e, nullptr, 10);
    return v < lo ? lo : v > hi ? hi : (uint32_t)v;
}
uint32_t gl_log_tile_max() { static const uint32_t v = gl_env_u32("SS_GL_LOG_TILE", GL_LOG_TILE_MAX, 10, 14); return v; }
uint32_t gl_log_min_run() { static const uint32_t v = gl_env_u32("SS_GL_MIN_RUN_LOG", 5, 2, 6); return v; }
static uint32_t gl_threads() { static const uint32_t v = gl_env_u32("SS_GL_THREADS", 256, 64, 512); return v; }

hipError_t gl_set_func_attributes() {
    const int bytes = 8 << 14;
    const void *kernels[] = {reinterpret_cast<const void *>(&gl_ntt_pass_kernel<true, true>), reinterpret_cast<const void *>(&gl_ntt_pass_kernel<true, false>),
                             reinterpret_cast<const void *>(&gl_ntt_pass_kernel<false, true>), reinterpret_cast<const void *>(&gl_ntt_pass_kernel<false, false>)};
    for (const void *k : kernels) {
        const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_gl_ntt_pass(hipStream_t st, bool dif, const void *const *src, void *const *dst, uint32_t ncols, const uint64_t *tw,
                              uint32_t log_n, uint32_t s0, uint32_t r, uint32_t log_tile, uint32_t u_first, uint32_t log_expand,
                              uint64_t scale) {
    GlCols cols;
    for (int c = 0; c < MAX_COLS; ++c) { cols.src[c] = c < (int)ncols ? (const uint64_t *)src[c] : nullptr; cols.dst[c] = c < (int)ncols ? (uint64_t *)dst[c] : nullptr; }
    GlPassParams p;
    p.log_n = log_n; p.s0 = s0; p.r = r; p.log_tile = log_tile; p.u_first = u_first; p.log_expand = log_expand; p.contig = (s0 == 0); p.scale = scale;
    dim3 grid(1u << (log_n - log_tile), ncols), block(gl_threads());
    const size_t lds = (size_t)8 << log_tile;
    if (p.contig) {
        if (dif) hipLaunchKernelGGL((gl_ntt_pass_kernel<true, true>), grid, block, lds, st, cols, tw, p);
        else hipLaunchKernelGGL((gl_ntt_pass_kernel<false, true>), grid, block, lds, st, cols, tw, p);
    } else {
        if (dif) hipLaunchKernelGGL((gl_ntt_pass_kernel<true, false>), grid, block, lds, st, cols, tw, p);
        else hipLaunchKernelGGL((gl_ntt_pass_kernel<false, false>), grid, block, lds, st, cols, tw, p);
    }
    return hipGetLastError();
}

hipError_t launch_gl_twiddles(hipStream_t st, uint64_t *tw, const uint64_t *pow_lo, const uint64_t *pow_hi, const uint64_t *hpow, uint32_t log_n) {
    const uint64_t total = (1ull << log_n) - 1ull;
    uint32_t blocks = (uint32_t)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(gl_twiddle_kernel, dim3(blocks), dim3(256), 0, st, tw, pow_lo, pow_hi, hpow, log_n);
    return hipGetLastError();
}

hipError_t launch_gl_bitrev_copy(hipStream_t st, const uint64_t *src, uint64_t *dst, uint32_t log_n) {
    const uint64_t n = 1ull << log_n;
    uint32_t blocks = (uint32_t)((n + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gl_bitrev_copy_kernel, dim3(blocks), dim3(256), 0, st, src, dst, log_n);
    return hipGetLastError();
}

hipError_t launch_gl3_fri_fold(hipStream_t st, const uint64_t *evals, uint32_t log_len, uint32_t fold, const uint64_t alpha[3], uint64_t offset,
                               bool unnormalised, uint64_t *out) {
    uint32_t log_fold = 0;
    while ((1u << log_fold) < fold) ++log_fold;
    Gl3FriConsts k;
    const uint64_t wf_inv = gl_inv_host(gl_root_of_unity_host(log_fold));
    k.winv[0] = 1;
    for (uint32_t i = 1; i < 16; ++i) k.winv[i] = i < fold ? gl_mul(k.winv[i - 1], wf_inv) : 0;
    k.inv_fold = unnormalised ? 1ull : gl_inv_host(fold);
    for (int c = 0; c < 3; ++c) k.alpha.c[c] = alpha[c] % GL_P;
    k.offset_inv = gl_inv_host(offset);
    k.wl_inv = gl_inv_host(gl_root_of_unity_host(log_len));
    const uint64_t rows = 1ull << (log_len - log_fold);
    uint32_t blocks = (uint32_t)((rows + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    if (blocks == 0) blocks = 1;
    switch (log_fold) {
    case 1: hipLaunchKernelGGL(gl3_fri_fold_kernel<1>, dim3(blocks), dim3(256), 0, st, evals, log_len, k, out); break;
    case 2: hipLaunchKernelGGL(gl3_fri_fold_kernel<2>, dim3(blocks), dim3(256), 0, st, evals, log_len, k, out); break;
    case 3: hipLaunchKernelGGL(gl3_fri_fold_kernel<3>, dim3(blocks), dim3(256), 0, st, evals, log_len, k, out); break;
    default: hipLaunchKernelGGL(gl3_fri_fold_kernel<4>, dim3(blocks), dim3(256), 0, st, evals, log_len, k, out); break;
    }
    return hipGetLastError();
}


// ---- DEEP over the cubic extension ------------------------------------------------------------------------------------
// The out-of-domain point z and every coefficient live in Fq3, the trace in Fp.  Same structure as csrc/deep.hip: since
// 1/(x - z w_n^k) = w_n^-k / (x w_n^-k - z), every denominator is ONE table D[m] = 1/(x_m - z) over the n-point sub-coset
// read at a shifted index; a cell's trace value does not depend on its offset, so the sum runs column by column:
//   out[m] = sum_c T_c[i] * S_c[m] + S_K[m] + Dc[m] * (sum_k cc_k H_k[i] - Kc),    S_c[m] = sum_taps c' * D[m - shift]
// with Fq3 x Fq3 products inside S and Fp x Fq3 products outside.  An Fq3-valued column (extension trace, composition) is
// three Fp columns whose cells carry the coefficients c, c X, c X^2 - the caller's expansion, the sum is linear.
__device__ __forceinline__ uint64_t gl_pow_dev(uint64_t a, uint64_t e) { return gl_pow(a, e); }

// D[m] = 1 / (x0 * w^m - z), m < len, interleaved [len][3].  a = (x - z0, -z1, -z2); a^-1 = adj(a) / N(a) with
// adj = (a0^2 - 2 a1 a2, 2 a2^2 - a0 a1, a1^2 - a0 a2) and N = a0 adj0 + 2 a2 adj1 + 2 a1 adj2 in Fp; the norms of a
// lane's GL3_INV_CHUNK consecutive entries are inverted together (Montgomery's trick, one Fermat inversion per lane).
// x = z0 with z1 = z2 = 0 (z on the domain) has no inverse: the entry is 0, as the 252-bit table's is.
static constexpr int GL3_INV_CHUNK = 8;
__global__ __launch_bounds__(256) void gl3_inverse_table_kernel(uint64_t *__restrict__ D, uint64_t len, uint64_t x0, uint64_t w, Gl3 z) {
    const uint64_t lane = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, m0 = lane * GL3_INV_CHUNK;
    if (m0 >= len) return;
    uint64_t x = gl_mul(x0, gl_pow_dev(w, m0));
    Gl3 adj[GL3_INV_CHUNK];
    uint64_t norm[GL3_INV_CHUNK], prefix[GL3_INV_CHUNK];
    uint64_t run = 1;
#pragma unroll
    for (int k = 0; k < GL3_INV_CHUNK; ++k) {
        const uint64_t a0 = gl_sub(x, z.c[0]), a1 = gl_sub(0, z.c[1]), a2 = gl_sub(0, z.c[2]);
        const uint64_t a12 = gl_mul(a1, a2), a22 = gl_mul(a2, a2);
        adj[k].c[0] = gl_sub(gl_mul(a0, a0), gl_add(a12, a12));
        adj[k].c[1] = gl_sub(gl_add(a22, a22), gl_mul(a0, a1));
        adj[k].c[2] = gl_sub(gl_mul(a1, a1), gl_mul(a0, a2));
        const uint64_t t1 = gl_mul(a2, adj[k].c[1]), t2 = gl_mul(a1, adj[k].c[2]);
        norm[k] = gl_add(gl_mul(a0, adj[k].c[0]), gl_add(gl_add(t1, t1), gl_add(t2, t2)));
        prefix[k] = run;
        if (norm[k]) run = gl_mul(run, norm[k]);
        x = gl_mul(x, w);
    }
    uint64_t inv = gl_pow_dev(run, GL_P - 2);
#pragma unroll
    for (int k = GL3_INV_CHUNK - 1; k >= 0; --k) {
        const uint64_t ni = norm[k] ? gl_mul(inv, prefix[k]) : 0;
        if (norm[k]) inv = gl_mul(inv, norm[k]);
        if (m0 + k < len) {
            const Gl3 r = gl3_scale(adj[k], ni);
            D[3 * (m0 + k)] = r.c[0]; D[3 * (m0 + k) + 1] = r.c[1]; D[3 * (m0 + k) + 2] = r.c[2];
        }
    }
}

struct Gl3DeepArgs {
    const uint64_t *trace[MAX_COLS];
    const uint64_t *comp[12];
    const uint64_t *D, *Dc;          // [n][3]
    const uint32_t *tap_shift;       // [ntaps] by column, then by shift (sub-coset units)
    const uint64_t *tap_rhs;         // [ntaps][5]: c = coeff * w_n^-off (the constant column: -K_off) as c0, c1, c2, 2 c1, 2 c2
    const uint32_t *col_desc;        // [ncoldesc][3]: trace column (0xffffffff: constants), first tap, count
    const uint64_t *comp_coef;       // [ncomp][3]
    Gl3 comp_k;
    uint32_t ncoldesc, ncomp, log_stride, mask;
    uint64_t count;
};
__global__ __launch_bounds__(256) void gl3_deep_kernel(Gl3DeepArgs a, uint64_t *__restrict__ out0, uint64_t *__restrict__ out1, uint64_t *__restrict__ out2) {
    // every sum below is a wide accumulator (gl64.h): a column's S pays three reductions for all its taps instead of nine per tap,
    // and the sum over the columns three more
    for (uint64_t m = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; m < a.count; m += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = m << a.log_stride;
        Gl3Wide acc = gl3w_zero();
        for (uint32_t k = 0; k < a.ncoldesc; ++k) {
            const uint32_t col = a.col_desc[3 * k], first = a.col_desc[3 * k + 1], cnt = a.col_desc[3 * k + 2];
            Gl3Wide Sw = gl3w_zero();
            for (uint32_t j = first; j < first + cnt; ++j)
                gl3w_mac(Sw, gl3_load(a.D + 3 * (size_t)(((uint32_t)m - a.tap_shift[j]) & a.mask)), gl3_load_rhs(a.tap_rhs + 5 * (size_t)j));
            const Gl3 S = gl3w_reduce(Sw);
            if (col == 0xffffffffu) gl3w_add(acc, S);
            else {
                const uint64_t *tp = a.trace[0];
#pragma unroll
                for (int c = 1; c < MAX_COLS; ++c) if (col == (uint32_t)c) tp = a.trace[c];
                gl3w_mac_base(acc, S, tp[i]);
            }
        }
        if (a.ncomp) {
            Gl3Wide inner = gl3w_zero();
            for (uint32_t k = 0; k < a.ncomp; ++k) {
                const uint64_t *hp = a.comp[0];
#pragma unroll
                for (int c = 1; c < 12; ++c) if (k == (uint32_t)c) hp = a.comp[c];
                gl3w_mac_base(inner, gl3_load(a.comp_coef + 3 * (size_t)k), hp[i]);
            }
            gl3w_mac(acc, gl3_sub(gl3w_reduce(inner), a.comp_k), gl3_rhs(gl3_load(a.Dc + 3 * m)));
        }
        const Gl3 r = gl3w_reduce(acc);
        out0[m] = r.c[0]; out1[m] = r.c[1]; out2[m] = r.c[2];
    }
}

// three planar component columns -> interleaved [len][3]
__global__ void gl3_interleave_kernel(const uint64_t *__restrict__ c0, const uint64_t *__restrict__ c1, const uint64_t *__restrict__ c2,
                                      uint64_t len, uint64_t *__restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < len; i += (uint64_t)gridDim.x * blockDim.x) {
        out[3 * i] = c0[i]; out[3 * i + 1] = c1[i]; out[3 * i + 2] = c2[i];
    }
}

// zp[q] = z^bitrev(q), q < 2^log_n, planar [3][n]: what a bit-reversed coefficient array is scaled by before the transform
// that evaluates it on z * <w_n>.  A lane walks 64 consecutive q: bitrev(q) = bitrev6(q & 63) << (log_n - 6) | bitrev(q >> 6),
// so z^bitrev(q) = z^bitrev(q >> 6) * (z^(2^(log_n - 6)))^bitrev6(q & 63): one square-and-multiply per lane, then a 64-entry table.
__global__ __launch_bounds__(256) void gl3_zpow_bitrev_kernel(uint64_t *__restrict__ zp0, uint64_t *__restrict__ zp1, uint64_t *__restrict__ zp2,
                                                              uint32_t log_n, Gl3 z) {
    __shared__ uint64_t tab[64][3];                       // (z^(2^(log_n - lo)))^e, e < 2^lo
    const uint32_t lo = log_n < 6 ? log_n : 6;
    if (threadIdx.x == 0) {
        Gl3 base = z;
        for (uint32_t b = 0; b + lo < log_n; ++b) base = gl3_mul(base, base);
        Gl3 p = {{1, 0, 0}};
        for (uint32_t e = 0; e < (1u << lo); ++e) { tab[e][0] = p.c[0]; tab[e][1] = p.c[1]; tab[e][2] = p.c[2]; p = gl3_mul(p, base); }
    }
    __syncthreads();
    const uint64_t groups = 1ull << (log_n - lo);
    for (uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; g < groups; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t e_hi = log_n > lo ? (__brevll(g) >> (64u - (log_n - lo))) : 0;     // bitrev of the high part: the exponent's low bits
        Gl3 hi = {{1, 0, 0}}, sq = z;
        for (uint64_t e = e_hi; e; e >>= 1) { if (e & 1) hi = gl3_mul(hi, sq); sq = gl3_mul(sq, sq); }
        for (uint32_t t = 0; t < (1u << lo); ++t) {
            const uint32_t e_lo = __brev(t) >> (32u - lo);
            const Gl3 v = gl3_mul(hi, Gl3{{tab[e_lo][0], tab[e_lo][1], tab[e_lo][2]}});
            const uint64_t q = (g << lo) | t;
            zp0[q] = v.c[0]; zp1[q] = v.c[1]; zp2[q] = v.c[2];
        }
    }
}
// out_k[q] = coef[q] * zp_k[q]
__global__ void gl3_scale_columns_kernel(const uint64_t *__restrict__ coef, const uint64_t *__restrict__ zp0, const uint64_t *__restrict__ zp1,
                                         const uint64_t *__restrict__ zp2, uint64_t n, uint64_t *__restrict__ o0, uint64_t *__restrict__ o1, uint64_t *__restrict__ o2) {
    for (uint64_t q = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; q < n; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t c = coef[q];
        o0[q] = gl_mul(c, zp0[q]); o1[q] = gl_mul(c, zp1[q]); o2[q] = gl_mul(c, zp2[q]);
    }
}
// out[3 j + k] = comp_k[idx[j]]
__global__ void gl3_gather_kernel(const uint64_t *__restrict__ c0, const uint64_t *__restrict__ c1, const uint64_t *__restrict__ c2,
                                  const uint64_t *__restrict__ idx, uint32_t count, uint64_t *__restrict__ out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    out[3 * j] = c0[idx[j]]; out[3 * j + 1] = c1[idx[j]]; out[3 * j + 2] = c2[idx[j]];
}

// partial[b][t] = sum over block b's share of coef[q] * zp_t[q]: P(z) = sum_j c_j z^j for a bit-reversed coefficient array and the
// power table z^bitrev(q) - the out-of-domain value of a column that is read at a single point
__global__ __launch_bounds__(256) void gl3_dot_kernel(const uint64_t *__restrict__ coef, const uint64_t *__restrict__ zp0, const uint64_t *__restrict__ zp1,
                                                      const uint64_t *__restrict__ zp2, uint64_t n, uint64_t *__restrict__ partial) {
    __shared__ uint64_t red[3][256];
    GlWide a0 = glw_zero(), a1 = glw_zero(), a2 = glw_zero();     // a lane's share is far below the 2^32 terms a wide sum holds
    for (uint64_t q = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; q < n; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t c = coef[q];
        glw_mac(a0, c, zp0[q]); glw_mac(a1, c, zp1[q]); glw_mac(a2, c, zp2[q]);
    }
    red[0][threadIdx.x] = glw_reduce(a0); red[1][threadIdx.x] = glw_reduce(a1); red[2][threadIdx.x] = glw_reduce(a2);
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int t = 0; t < 3; ++t) red[t][threadIdx.x] = gl_add(red[t][threadIdx.x], red[t][threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x < 3) partial[3 * blockIdx.x + threadIdx.x] = red[threadIdx.x][0];
}

static uint32_t gl_blocks(uint64_t items, uint32_t per_block = 256, uint32_t cap = 16384) {
    uint64_t b = (items + per_block - 1) / per_block;
    return (uint32_t)(b == 0 ? 1 : b > cap ? cap : b);
}
hipError_t launch_gl3_inverse_table(hipStream_t st, uint64_t *D, uint64_t len, uint64_t x0, uint64_t w, const uint64_t z[3]) {
    const uint64_t lanes = (len + GL3_INV_CHUNK - 1) / GL3_INV_CHUNK;
    hipLaunchKernelGGL(gl3_inverse_table_kernel, dim3((uint32_t)((lanes + 255) / 256)), dim3(256), 0, st, D, len, x0, w, Gl3{{z[0], z[1], z[2]}});
    return hipGetLastError();
}
hipError_t launch_gl3_deep(hipStream_t st, const uint64_t *const *trace, uint32_t ntrace, const uint64_t *const *comp, uint32_t ncomp,
                           const uint64_t *D, const uint64_t *Dc, const uint32_t *tap_shift, const uint64_t *tap_coef, const uint32_t *col_desc,
                           uint32_t ncoldesc, const uint64_t *comp_coef, const uint64_t comp_k[3], uint64_t count, uint32_t log_stride,
                           uint64_t *out0, uint64_t *out1, uint64_t *out2) {
    Gl3DeepArgs a;
    for (int c = 0; c < MAX_COLS; ++c) a.trace[c] = c < (int)ntrace ? trace[c] : nullptr;
    for (int c = 0; c < 12; ++c) a.comp[c] = c < (int)ncomp ? comp[c] : nullptr;
    a.D = D; a.Dc = Dc; a.tap_shift = tap_shift; a.tap_rhs = tap_coef; a.col_desc = col_desc; a.ncoldesc = ncoldesc;
    a.comp_coef = comp_coef; a.comp_k = Gl3{{comp_k[0], comp_k[1], comp_k[2]}}; a.ncomp = ncomp; a.log_stride = log_stride;
    a.mask = (uint32_t)(count - 1); a.count = count;
    hipLaunchKernelGGL(gl3_deep_kernel, dim3(gl_blocks(count, 256, 8192)), dim3(256), 0, st, a, out0, out1, out2);
    return hipGetLastError();
}
hipError_t launch_gl3_interleave(hipStream_t st, const uint64_t *c0, const uint64_t *c1, const uint64_t *c2, uint64_t len, uint64_t *out) {
    hipLaunchKernelGGL(gl3_interleave_kernel, dim3(gl_blocks(len)), dim3(256), 0, st, c0, c1, c2, len, out);
    return hipGetLastError();
}
hipError_t launch_gl3_zpow_bitrev(hipStream_t st, uint64_t *zp0, uint64_t *zp1, uint64_t *zp2, uint32_t log_n, const uint64_t z[3]) {
    const uint64_t groups = 1ull << (log_n < 6 ? 0 : log_n - 6);
    hipLaunchKernelGGL(gl3_zpow_bitrev_kernel, dim3(gl_blocks(groups, 256, 4096)), dim3(256), 0, st, zp0, zp1, zp2, log_n, Gl3{{z[0], z[1], z[2]}});
    return hipGetLastError();
}
hipError_t launch_gl3_scale_columns(hipStream_t st, const uint64_t *coef, const uint64_t *zp0, const uint64_t *zp1, const uint64_t *zp2, uint64_t n,
                                    uint64_t *o0, uint64_t *o1, uint64_t *o2) {
    hipLaunchKernelGGL(gl3_scale_columns_kernel, dim3(gl_blocks(n)), dim3(256), 0, st, coef, zp0, zp1, zp2, n, o0, o1, o2);
    return hipGetLastError();
}
uint32_t gl3_dot_blocks(uint64_t n) { return gl_blocks(n, 256, 1024); }
hipError_t launch_gl3_dot(hipStream_t st, const uint64_t *coef, const uint64_t *zp0, const uint64_t *zp1, const uint64_t *zp2, uint64_t n, uint64_t *partial) {
    hipLaunchKernelGGL(gl3_dot_kernel, dim3(gl3_dot_blocks(n)), dim3(256), 0, st, coef, zp0, zp1, zp2, n, partial);
    return hipGetLastError();
}
hipError_t launch_gl3_gather(hipStream_t st, const uint64_t *c0, const uint64_t *c1, const uint64_t *c2, const uint64_t *idx, uint32_t count, uint64_t *out) {
    if (!count) return hipSuccess;
    hipLaunchKernelGGL(gl3_gather_kernel, dim3((count + 127) / 128), dim3(128), 0, st, c0, c1, c2, idx, count, out);
    return hipGetLastError();
}


// ---- the constraint program over the cubic extension (row Q1 for this field) ---------------------------------------------
// The same 4-accumulator register machine as csrc/quotient.hip (include/sandstorm_hip.h ss_air_program), interpreted:
// accumulators, constants (challenges, alpha^k, hints) and scratch slots are elements of Fq3, trace cells / periodic tables /
// x are elements of Fp read into the first coordinate.  One lane = one LDE point; the program counter and operand selectors are
// wave-uniform (scalar loads and branches).  Slots live in a file [slot][coordinate][lane] in HBM.
__device__ __forceinline__ Gl3 gl3_inv_dev(const Gl3 &a) { return gl3_inv(a); }            // gl64.h: adjugate over the norm, 0 -> 0
// Most values of a constraint program are base-field values in extension clothes (trace cells, their sums and products: every
// CPU constraint before its alpha^k): a factor whose upper coordinates are zero costs one or three multiplications instead of
// nine.  The test is on the data, per lane; which values are base-field is a property of the program, so the lanes agree.
__device__ __forceinline__ Gl3 gl3_mul_typed(const Gl3 &a, const Gl3 &b) {
    const bool a_base = (a.c[1] | a.c[2]) == 0, b_base = (b.c[1] | b.c[2]) == 0;
    if (a_base && b_base) return Gl3{{gl_mul(a.c[0], b.c[0]), 0, 0}};
    if (b_base) return gl3_scale(a, b.c[0]);
    if (a_base) return gl3_scale(b, a.c[0]);
    return gl3_mul(a, b);
}
__device__ __forceinline__ Gl3 gl3_inv_typed(const Gl3 &a) {
    if ((a.c[1] | a.c[2]) == 0) return Gl3{{gl_pow_dev(a.c[0], GL_P - 2), 0, 0}};
    return gl3_inv_dev(a);
}
struct Gl3VmArgs {
    const uint32_t *code;            // 2 words per instruction
    const uint64_t *consts;          // [n_consts][3]
    const uint64_t *tables;          // concatenated Fp tables
    const uint32_t *tdesc;           // [n_tables][2]: offset, length - 1
    const uint64_t *cols[MAX_COLS];
    uint64_t *slots;                 // [n_slots][3][lanes]
    uint64_t *out;                   // [N][3]
    uint64_t offset, w;
    uint32_t n_instr, log_blowup;
    uint64_t N;
};
__global__ __launch_bounds__(256) void gl3_vm_kernel(Gl3VmArgs a) {
    typedef const uint32_t __attribute__((address_space(4))) *const_u32;
    const_u32 code = (const_u32)(uintptr_t)a.code, tdesc = (const_u32)(uintptr_t)a.tdesc;
    const uint64_t lanes = (uint64_t)gridDim.x * blockDim.x, lane = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t wstep = gl_pow_dev(a.w, lanes);
    uint64_t x = gl_mul(a.offset, gl_pow_dev(a.w, lane));
    for (uint64_t i = lane; i < a.N; i += lanes, x = gl_mul(x, wstep)) {
        Gl3 acc0 = {{0, 0, 0}}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
        for (uint32_t pc = 0; pc < a.n_instr; ++pc) {
            const uint32_t w0 = code[2 * pc], w1 = code[2 * pc + 1];
            const uint32_t op = w0 & 0xffu, d = (w0 >> 8) & 0xfu, kind = (w0 >> 12) & 0xfu;
            Gl3 src = {{0, 0, 0}};
            if (op <= 4u) {
                switch (kind) {
                case 0: { const uint32_t k = w1 & 3u; src = k == 0 ? acc0 : k == 1 ? acc1 : k == 2 ? acc2 : acc3; } break;
                case 1: for (int c = 0; c < 3; ++c) src.c[c] = a.slots[((uint64_t)w1 * 3 + c) * lanes + lane]; break;
                case 2: src = gl3_load(a.consts + 3 * (size_t)w1); break;
                case 3: {
                    const uint32_t col = w1 >> 24, ro = w1 & 0xffffffu;
                    const uint64_t *cp = a.cols[0];
#pragma unroll
                    for (int c = 1; c < MAX_COLS; ++c) if (col == (uint32_t)c) cp = a.cols[c];
                    src.c[0] = cp[(i + ((uint64_t)ro << a.log_blowup)) & (a.N - 1)];
                } break;
                case 4: src.c[0] = a.tables[tdesc[2 * w1] + (i & tdesc[2 * w1 + 1])]; break;
                default: src.c[0] = x; break;
                }
            }
#define SS_GL3_VM_ON(v)                                                                                          \
            switch (op) {                                                                                        \
            case 0: v = src; break;                                                                              \
            case 1: v = gl3_add(v, src); break;                                                                  \
            case 2: v = gl3_sub(v, src); break;                                                                  \
            case 3: v = gl3_sub(src, v); break;                                                                  \
            case 4: v = gl3_mul_typed(v, src); break;                                                            \
            case 5: v = gl3_inv_typed(v); break;                                                                 \
            case 6: for (int c = 0; c < 3; ++c) a.slots[((uint64_t)w1 * 3 + c) * lanes + lane] = v.c[c]; break;  \
            case 7: a.out[3 * i] = v.c[0]; a.out[3 * i + 1] = v.c[1]; a.out[3 * i + 2] = v.c[2]; break;          \
            default: break;                                                                                      \
            }
            switch (d) {
            case 0: SS_GL3_VM_ON(acc0) break;
            case 1: SS_GL3_VM_ON(acc1) break;
            case 2: SS_GL3_VM_ON(acc2) break;
            default: SS_GL3_VM_ON(acc3) break;
            }
#undef SS_GL3_VM_ON
        }
    }
}
uint32_t gl3_vm_lanes(uint64_t N) { return gl_blocks(N, 256, 4096) * 256u; }
hipError_t launch_gl3_vm(hipStream_t st, const uint32_t *d_code, uint32_t n_instr, const uint64_t *d_consts, const uint64_t *d_tables,
                         const uint32_t *d_tdesc, const uint64_t *const *cols, uint32_t ncols, uint64_t *d_slots, uint64_t *d_out, uint64_t offset,
                         uint64_t w, uint32_t log_blowup, uint64_t N) {
    Gl3VmArgs a;
    a.code = d_code; a.consts = d_consts; a.tables = d_tables; a.tdesc = d_tdesc; a.slots = d_slots; a.out = d_out;
    for (int c = 0; c < MAX_COLS; ++c) a.cols[c] = c < (int)ncols ? cols[c] : nullptr;
    a.offset = offset; a.w = w; a.n_instr = n_instr; a.log_blowup = log_blowup; a.N = N;
    hipLaunchKernelGGL(gl3_vm_kernel, dim3(gl3_vm_lanes(N) / 256), dim3(256), 0, st, a);
    return hipGetLastError();
}


// ---- running products over Fq3: the permutation arguments' extension column (plain layout: trace.rs:274-330) ----------------------
// out[k] = prod_{i <= k} (z - (alpha nv_i + na_i)) / prod_{i <= k} (z - (alpha dv_i + da_i)), written as three coordinate columns at
// row out_offset + k out_stride.  The state of the scan is the PAIR (numerator product, denominator product) in Fq3; levels of 8
// items per lane: reduce (8 states -> 1) down to a handful, scan those serially, then apply back up (exclusive prefixes of the
// groups), and the last apply writes numerator * denominator^-1 (one norm-based inverse per output).
struct Gl3Pair { Gl3 n, d; };
__device__ __forceinline__ Gl3Pair gl3p_mul(const Gl3Pair &a, const Gl3Pair &b) { return Gl3Pair{gl3_mul(a.n, b.n), gl3_mul(a.d, b.d)}; }
__device__ __forceinline__ Gl3Pair gl3p_load(const uint64_t *p) { return Gl3Pair{gl3_load(p), gl3_load(p + 3)}; }
__device__ __forceinline__ void gl3p_store(uint64_t *p, const Gl3Pair &v) {
    p[0] = v.n.c[0]; p[1] = v.n.c[1]; p[2] = v.n.c[2]; p[3] = v.d.c[0]; p[4] = v.d.c[1]; p[5] = v.d.c[2];
}
struct Gl3PermArgs {
    const uint64_t *na, *nv, *da, *dv;        // nv / dv may be NULL (single-column argument: terms z - a)
    uint64_t stride;                          // item i reads element i * stride of each
    uint64_t count;
    Gl3 z, alpha;
};
__device__ __forceinline__ Gl3Pair gl3_perm_term(const Gl3PermArgs &a, uint64_t i) {
    Gl3Pair t;
    Gl3 n = a.nv ? gl3_scale(a.alpha, a.nv[i * a.stride]) : Gl3{{0, 0, 0}};
    n.c[0] = gl_add(n.c[0], a.na[i * a.stride]);
    Gl3 d = a.dv ? gl3_scale(a.alpha, a.dv[i * a.stride]) : Gl3{{0, 0, 0}};
    d.c[0] = gl_add(d.c[0], a.da[i * a.stride]);
    t.n = gl3_sub(a.z, n);
    t.d = gl3_sub(a.z, d);
    return t;
}
static constexpr int GL3_SCAN_GROUP = 8;
// level 0: out[j] = product of the terms of items 8 j .. 8 j + 7
__global__ void gl3_perm_reduce0_kernel(Gl3PermArgs a, uint64_t groups, uint64_t *__restrict__ out) {
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < groups; j += (uint64_t)gridDim.x * blockDim.x) {
        Gl3Pair run{{{1, 0, 0}}, {{1, 0, 0}}};
        for (int i = 0; i < GL3_SCAN_GROUP; ++i) { const uint64_t k = j * GL3_SCAN_GROUP + i; if (k < a.count) run = gl3p_mul(run, gl3_perm_term(a, k)); }
        gl3p_store(out + 6 * j, run);
    }
}
__global__ void gl3_pair_reduce_kernel(const uint64_t *__restrict__ in, uint64_t m, uint64_t groups, uint64_t *__restrict__ out) {
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < groups; j += (uint64_t)gridDim.x * blockDim.x) {
        Gl3Pair run{{{1, 0, 0}}, {{1, 0, 0}}};
        for (int i = 0; i < GL3_SCAN_GROUP; ++i) { const uint64_t k = j * GL3_SCAN_GROUP + i; if (k < m) run = gl3p_mul(run, gl3p_load(in + 6 * k)); }
        gl3p_store(out + 6 * j, run);
    }
}
// top: exclusive scan of a handful of states by one lane (in place)
__global__ void gl3_pair_scan_top_kernel(uint64_t *__restrict__ st, uint64_t m) {
    if (blockIdx.x || threadIdx.x) return;
    Gl3Pair run{{{1, 0, 0}}, {{1, 0, 0}}};
    for (uint64_t k = 0; k < m; ++k) { const Gl3Pair v = gl3p_load(st + 6 * k); gl3p_store(st + 6 * k, run); run = gl3p_mul(run, v); }
}
// apply: st[k] (the group products of this level) become their exclusive prefixes, seeded by the level above (prefix[j] for group j)
__global__ void gl3_pair_apply_kernel(uint64_t *__restrict__ st, uint64_t m, const uint64_t *__restrict__ prefix, uint64_t groups) {
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < groups; j += (uint64_t)gridDim.x * blockDim.x) {
        Gl3Pair run = gl3p_load(prefix + 6 * j);
        for (int i = 0; i < GL3_SCAN_GROUP; ++i) {
            const uint64_t k = j * GL3_SCAN_GROUP + i;
            if (k < m) { const Gl3Pair v = gl3p_load(st + 6 * k); gl3p_store(st + 6 * k, run); run = gl3p_mul(run, v); }
        }
    }
}
// level 0 apply: the running quotient of every item, into the coordinate columns
__global__ void gl3_perm_apply0_kernel(Gl3PermArgs a, const uint64_t *__restrict__ prefix, uint64_t groups, uint64_t *__restrict__ o0,
                                       uint64_t *__restrict__ o1, uint64_t *__restrict__ o2, uint64_t out_stride, uint64_t out_offset,
                                       uint64_t *__restrict__ last) {
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < groups; j += (uint64_t)gridDim.x * blockDim.x) {
        Gl3Pair run = gl3p_load(prefix + 6 * j);
        for (int i = 0; i < GL3_SCAN_GROUP; ++i) {
            const uint64_t k = j * GL3_SCAN_GROUP + i;
            if (k >= a.count) break;
            run = gl3p_mul(run, gl3_perm_term(a, k));
            const Gl3 q = gl3_mul(run.n, gl3_inv_dev(run.d));
            const uint64_t row = out_offset + k * out_stride;
            o0[row] = q.c[0]; o1[row] = q.c[1]; o2[row] = q.c[2];
            if (k == a.count - 1 && last) { last[0] = q.c[0]; last[1] = q.c[1]; last[2] = q.c[2]; }
        }
    }
}
hipError_t launch_gl3_running_product(hipStream_t st, const uint64_t *na, const uint64_t *nv, const uint64_t *da, const uint64_t *dv, uint64_t stride,
                                      uint64_t count, const uint64_t z[3], const uint64_t alpha[3], uint64_t *scratch /* >= 6 * (count / 7 + 64) u64 */,
                                      uint64_t *o0, uint64_t *o1, uint64_t *o2, uint64_t out_stride, uint64_t out_offset, uint64_t *d_last) {
    Gl3PermArgs a{na, nv, da, dv, stride, count, Gl3{{z[0], z[1], z[2]}}, Gl3{{alpha[0], alpha[1], alpha[2]}}};
    // level sizes: m[0] = groups of items, m[l + 1] = groups of m[l]
    uint64_t m[16], off[16];
    int levels = 0;
    uint64_t cur = (count + GL3_SCAN_GROUP - 1) / GL3_SCAN_GROUP, at = 0;
    m[0] = cur; off[0] = 0; at = 6 * cur; levels = 1;
    while (cur > 16) {
        cur = (cur + GL3_SCAN_GROUP - 1) / GL3_SCAN_GROUP;
        m[levels] = cur; off[levels] = at; at += 6 * cur; ++levels;
    }
    hipLaunchKernelGGL(gl3_perm_reduce0_kernel, dim3(gl_blocks(m[0])), dim3(256), 0, st, a, m[0], scratch + off[0]);
    for (int l = 1; l < levels; ++l)
        hipLaunchKernelGGL(gl3_pair_reduce_kernel, dim3(gl_blocks(m[l])), dim3(256), 0, st, scratch + off[l - 1], m[l - 1], m[l], scratch + off[l]);
    hipLaunchKernelGGL(gl3_pair_scan_top_kernel, dim3(1), dim3(64), 0, st, scratch + off[levels - 1], m[levels - 1]);
    for (int l = levels - 1; l >= 1; --l)
        hipLaunchKernelGGL(gl3_pair_apply_kernel, dim3(gl_blocks(m[l])), dim3(256), 0, st, scratch + off[l - 1], m[l - 1], scratch + off[l], m[l]);
    hipLaunchKernelGGL(gl3_perm_apply0_kernel, dim3(gl_blocks(m[0])), dim3(256), 0, st, a, scratch + off[0], m[0], o0, o1, o2, out_stride, out_offset, d_last);
    return hipGetLastError();
}


// ---- the plain layout's composition, compiled (tools/gen_quotient_gl.py) ---------------------------------------------------------
#include "quotient_gen_plain_gl.inc"

// the compiled kernel for this program, if it is the plain layout's: the code words hash to the generated kernel's, the shapes agree,
// and every constant the generator typed base-field is one in this table
bool gl3_compiled_matches(const uint32_t *code, uint32_t n_instr, const uint64_t *consts3, uint32_t n_consts, uint32_t n_tables) {
    if (n_instr != GL3_PLAIN_N_INSTR || n_consts != GL3_PLAIN_N_CONSTS || n_tables != GL3_PLAIN_N_TABLES) return false;
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t k = 0; k < 2 * (size_t)n_instr; ++k)
        for (int b = 0; b < 4; ++b) h = (h ^ ((code[k] >> (8 * b)) & 0xffu)) * 0x100000001b3ull;
    if (h != GL3_PLAIN_CODE_HASH) return false;
    for (uint16_t k : GL3_PLAIN_BASE_CONSTS)
        if (consts3[3 * (size_t)k + 1] | consts3[3 * (size_t)k + 2]) return false;
    return true;
}
hipError_t launch_gl3_plain(hipStream_t st, const uint64_t *d_consts, const uint64_t *d_tables, const uint32_t *d_tdesc, const uint64_t *const *cols,
                            uint32_t ncols, uint64_t *d_out, uint64_t offset, uint64_t w, uint32_t log_blowup, uint64_t N) {
    Gl3VmArgs a;
    a.code = nullptr; a.consts = d_consts; a.tables = d_tables; a.tdesc = d_tdesc; a.slots = nullptr; a.out = d_out;
    for (int c = 0; c < MAX_COLS; ++c) a.cols[c] = c < (int)ncols ? cols[c] : nullptr;
    a.offset = offset; a.w = w; a.n_instr = 0; a.log_blowup = log_blowup; a.N = N;
    hipLaunchKernelGGL(gl3_plain_kernel, dim3(gl_blocks(N, 256, 8192)), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace ss
