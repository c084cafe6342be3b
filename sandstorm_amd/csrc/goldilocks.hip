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

namespace ss {

static constexpr uint64_t GL_P = 0xFFFFFFFF00000001ull;
static constexpr uint64_t GL_EPS = 0xFFFFFFFFull;          // 2^64 mod p

__host__ __device__ __forceinline__ uint64_t gl_add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    if (s < a) s += GL_EPS;                                  // wrapped: + 2^64 = + EPS (cannot wrap again: a, b < p)
    return s >= GL_P ? s - GL_P : s;
}
__host__ __device__ __forceinline__ uint64_t gl_sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a + (GL_P - b); }
__host__ __device__ __forceinline__ uint64_t gl_reduce128(uint64_t lo, uint64_t hi) {
    const uint64_t hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
    uint64_t t = lo - hi_hi;                                 // 2^96 = -1
    if (lo < hi_hi) t -= GL_EPS;                             // borrowed 2^64 = EPS
    const uint64_t m = hi_lo * GL_EPS;                       // 2^64 = EPS; < 2^64
    uint64_t r = t + m;
    if (r < t) r += GL_EPS;
    return r >= GL_P ? r - GL_P : r;
}
__host__ __device__ __forceinline__ uint64_t gl_mul(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t lo = a * b, hi = __umul64hi(a, b);
#else
    const unsigned __int128 pr = (unsigned __int128)a * b;
    const uint64_t lo = (uint64_t)pr, hi = (uint64_t)(pr >> 64);
#endif
    return gl_reduce128(lo, hi);
}
uint64_t gl_pow_host(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) { if (e & 1) r = gl_mul(r, a); a = gl_mul(a, a); e >>= 1; }
    return r;
}
uint64_t gl_root_of_unity_host(uint32_t log_n) { return gl_pow_host(7, (GL_P - 1) >> log_n); }
uint64_t gl_inv_host(uint64_t a) { return gl_pow_host(a, GL_P - 2); }

struct GlPassParams {
    uint32_t log_n, s0, r, log_tile, u_first, log_expand, contig;
    uint64_t scale;              // DIF last pass: multiply outputs by this (1/n); 1 = off
};
struct GlCols { const uint64_t *src[MAX_COLS]; uint64_t *dst[MAX_COLS]; };

// global element index of tile-local element e (ntt.hip tile_gindex: contiguous block, or 2^r rows x T adjacent elements)
__device__ __forceinline__ uint64_t gl_tile_gindex(const GlPassParams &p, uint32_t tile, uint32_t e) {
    if (p.contig) return ((uint64_t)tile << p.log_tile) + e;
    const uint32_t log_t = p.log_tile - p.r;
    const uint32_t dq = e & ((1u << log_t) - 1u), j = e >> log_t;
    const uint64_t q = ((uint64_t)tile << log_t) + dq;
    return ((q >> p.s0) << (p.s0 + p.r)) | ((uint64_t)j << p.s0) | (q & ((1ull << p.s0) - 1ull));
}

// LDS slot of tile element e: XOR swizzle (no padding).  A register group at shift sh makes 32 consecutive lanes walk
// either 32 consecutive elements (sh >= 5) or elements 16 apart (sh = 0, 4); e ^ ((e >> 4) & 31) maps both onto 32 distinct
// 8-byte slots modulo 32, i.e. all 64 banks once per ds_*_b64.
__device__ __forceinline__ uint32_t gl_slot(uint32_t e) { return e ^ ((e >> 4) & 31u); }

// One radix-2^G register group on local stages [u, u + G): each thread holds 2^G elements (element m at ebase + (m << sh))
// and runs G butterfly stages on them before the tile is touched again - ceil(13 / 4) = 4 LDS round trips per pass instead
// of 13.  The first / last group of a pass exchange with HBM directly (from_global / to_global).
// Twiddle plan as in the 252-bit path: T_s[k] at (2^s - 1) + k, k < 2^s, T_s[k] = h^(n / 2^(s+1)) * r^(k n / 2^(s+1)).
template <bool DIF, int G>
__device__ __forceinline__ void gl_group(uint64_t *tile_lds, const uint64_t *__restrict__ tw, const GlPassParams &p, uint32_t u, uint32_t tile,
                                         bool from_global, bool to_global, const uint64_t *__restrict__ src, uint64_t *__restrict__ dst) {
    const uint32_t log_t = p.log_tile - p.r, eshift = p.contig ? 0u : log_t;
    const uint32_t items = (1u << p.log_tile) >> G, sh = eshift + u;
    for (uint32_t tau = threadIdx.x; tau < items; tau += blockDim.x) {
        const uint32_t low = tau & ((1u << sh) - 1u), high = tau >> sh;
        const uint32_t ebase = (high << (sh + G)) | low;
        uint32_t jbase, lbits;
        if (p.contig) { jbase = ebase & ((1u << p.r) - 1u); lbits = 0; }
        else {
            jbase = ebase >> log_t;
            const uint32_t q = (tile << log_t) + (ebase & ((1u << log_t) - 1u));
            lbits = q & ((1u << p.s0) - 1u);
        }
        const uint32_t jlow = jbase & ((1u << u) - 1u);
        uint64_t x[1 << G];
#pragma unroll
        for (int m = 0; m < (1 << G); ++m) {
            const uint32_t e = ebase + ((uint32_t)m << sh);
            x[m] = from_global ? src[gl_tile_gindex(p, tile, e) >> p.log_expand] : tile_lds[gl_slot(e)];
        }
#pragma unroll
        for (int step = 0; step < G; ++step) {
            const int ST = DIF ? (G - 1 - step) : step;
            const uint32_t s = p.s0 + u + ST;
            const uint64_t *tws = tw + ((1ull << s) - 1ull);
#pragma unroll
            for (int pr = 0; pr < (1 << G) / 2; ++pr) {
                const int m = ((pr >> ST) << (ST + 1)) | (pr & ((1 << ST) - 1));
                const uint32_t k = ((jlow + ((uint32_t)(m & ((1 << ST) - 1)) << u)) << p.s0) | lbits;
                const uint64_t w = tws[k];
                const uint64_t a = x[m], b = x[m | (1 << ST)];
                if (DIF) {
                    x[m] = gl_add(a, b);
                    x[m | (1 << ST)] = gl_mul(gl_sub(a, b), w);
                } else {
                    const uint64_t bt = gl_mul(b, w);
                    x[m] = gl_add(a, bt);
                    x[m | (1 << ST)] = gl_sub(a, bt);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < (1 << G); ++m) {
            const uint32_t e = ebase + ((uint32_t)m << sh);
            if (to_global) {
                uint64_t v = x[m];
                if (DIF && p.scale != 1ull) v = gl_mul(v, p.scale);
                dst[gl_tile_gindex(p, tile, e)] = v;
            } else {
                tile_lds[gl_slot(e)] = x[m];
            }
        }
    }
}

template <bool DIF>
__global__ __launch_bounds__(256) void gl_ntt_pass_kernel(GlCols cols, const uint64_t *__restrict__ tw, GlPassParams p) {
    extern __shared__ uint64_t gl_tile[];
    const uint32_t tile = blockIdx.x;
    const void *src_v = cols.src[0];
    void *dst_v = cols.dst[0];
#pragma unroll
    for (int c = 1; c < MAX_COLS; ++c)
        if (blockIdx.y == (unsigned)c) { src_v = cols.src[c]; dst_v = cols.dst[c]; }
    const uint64_t *__restrict__ src = reinterpret_cast<const uint64_t *>(src_v);
    uint64_t *__restrict__ dst = reinterpret_cast<uint64_t *>(dst_v);
    // local stages [first, p.r) in groups of <= 4, ascending (DIT) or descending (DIF); the first group reads HBM, the last writes it
    const uint32_t first = DIF ? 0u : p.u_first, total = p.r - first;
    uint32_t done = 0;
    while (done < total) {
        const uint32_t g = (total - done) >= 4 ? 4u : (total - done);
        const uint32_t u = DIF ? (p.r - done - g) : (first + done);
        const bool fg = done == 0, tg = done + g == total;
        switch (g) {
        case 4: gl_group<DIF, 4>(gl_tile, tw, p, u, tile, fg, tg, src, dst); break;
        case 3: gl_group<DIF, 3>(gl_tile, tw, p, u, tile, fg, tg, src, dst); break;
        case 2: gl_group<DIF, 2>(gl_tile, tw, p, u, tile, fg, tg, src, dst); break;
        default: gl_group<DIF, 1>(gl_tile, tw, p, u, tile, fg, tg, src, dst); break;
        }
        done += g;
        if (!tg) __syncthreads();
    }
    if (total == 0)              // nothing to do in this pass (cannot happen for log_n >= 1): copy through
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
struct Gl3 { uint64_t c[3]; };
__device__ __forceinline__ Gl3 gl3_mul(const Gl3 &a, const Gl3 &b) {
    const uint64_t d0 = gl_mul(a.c[0], b.c[0]), d1 = gl_add(gl_mul(a.c[0], b.c[1]), gl_mul(a.c[1], b.c[0]));
    const uint64_t d2 = gl_add(gl_add(gl_mul(a.c[0], b.c[2]), gl_mul(a.c[1], b.c[1])), gl_mul(a.c[2], b.c[0]));
    const uint64_t d3 = gl_add(gl_mul(a.c[1], b.c[2]), gl_mul(a.c[2], b.c[1])), d4 = gl_mul(a.c[2], b.c[2]);
    Gl3 r;
    r.c[0] = gl_add(d0, gl_add(d3, d3));
    r.c[1] = gl_add(d1, gl_add(d4, d4));
    r.c[2] = d2;
    return r;
}
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
                uint64_t s = 0;
#pragma unroll
                for (int m = 0; m < FOLD; ++m) s = gl_add(s, gl_mul(v[c][m], k.winv[(t * m) & (FOLD - 1)]));
                coef.c[c] = gl_mul(s, xpow[t]);
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
uint32_t gl_log_tile_max() { return GL_LOG_TILE_MAX; }

hipError_t gl_set_func_attributes() {
    const int bytes = 8 << GL_LOG_TILE_MAX;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gl_ntt_pass_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&gl_ntt_pass_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

hipError_t launch_gl_ntt_pass(hipStream_t st, bool dif, const void *const *src, void *const *dst, uint32_t ncols, const uint64_t *tw,
                              uint32_t log_n, uint32_t s0, uint32_t r, uint32_t log_tile, uint32_t u_first, uint32_t log_expand,
                              uint64_t scale) {
    GlCols cols;
    for (int c = 0; c < MAX_COLS; ++c) { cols.src[c] = c < (int)ncols ? (const uint64_t *)src[c] : nullptr; cols.dst[c] = c < (int)ncols ? (uint64_t *)dst[c] : nullptr; }
    GlPassParams p;
    p.log_n = log_n; p.s0 = s0; p.r = r; p.log_tile = log_tile; p.u_first = u_first; p.log_expand = log_expand; p.contig = (s0 == 0); p.scale = scale;
    dim3 grid(1u << (log_n - log_tile), ncols), block(256);
    const size_t lds = (size_t)8 << log_tile;
    if (dif) hipLaunchKernelGGL(gl_ntt_pass_kernel<true>, grid, block, lds, st, cols, tw, p);
    else hipLaunchKernelGGL(gl_ntt_pass_kernel<false>, grid, block, lds, st, cols, tw, p);
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

}  // namespace ss
