// deep.hip — out-of-domain evaluation and DEEP composition on gfx950.
//
// Replaces ministark's DeepPolyComposer (un-vendored; coefficient rule
// src/lib.rs:102-116; row D1 of SURVEY.md §8a).
//
//  * poly_reduce_kernel: P(x) for bit-reversed coefficient arrays.  In
//    bit-reversed order position q holds coefficient bitrev(q), so
//    P(x) = sum_q c[q] * prod_j (x^(2^(L-1-j)))^(q_j): a pairwise tree whose
//    level-l multiplier is x^(2^(L-1-l)).  One launch folds 3 levels (8 adjacent
//    elements per lane -> 1), so evaluating a 2^24-coefficient column streams it
//    once (1.14 x n x 32 B in total).
//  * batch_inverse_kernel: D[i] = 1/(x_i - z) over the whole LDE domain with
//    Montgomery's trick on per-lane chunks (5 mulmods per point + one inversion
//    per chunk).
//  * deep_kernel: since 1/(x_i - z w_n^k) = w_n^-k * D[i - k*blowup], every DEEP
//    denominator is one table D read at a shifted index; the host folds w_n^-k
//    into the coefficients and groups mask cells by offset:
//      out[i] = sum_g D[i - off_g b] * (sum_{j in g} c'_j T_{col_j}[i] - K_g)
//             + Dc[i] * (sum_k cc_k H_k[i] - Kc)
//    = nmask + ngroups + ncomp + 1 multiplications per point, all operands
//    coalesced (lane i reads element i, or i - shift, of each array).  The DEEP
//    polynomial has degree < n, so it is composed on the n-point sub-coset
//    offset*<w_n> only (every blowup-th LDE row) and then interpolated and
//    re-expanded by two NTTs — half the pointwise work for blowup 2.  Products are
//    accumulated in the lazy 9 x 28-bit form with one weak reduction per 12 terms.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "fp252.h"
#include "fl252.h"
#include "inv252.h"
#include "kernels.h"

namespace ss {

__device__ __forceinline__ Fp dload(const Fp *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    Fp r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__device__ __forceinline__ void dstore(Fp *p, const Fp &x) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    q[1] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
}

// ---- bit-reversed polynomial evaluation: out[col][m] = fold of in[col][m*2^levels ..] ----
struct ReduceArgs {
    const Fp *in[MAX_COLS];
    Fp *out[MAX_COLS];
    Fp mult[3];          // multipliers of the 1st, 2nd, 3rd level folded by this launch
};

template <int LEVELS>
__global__ __launch_bounds__(256) void poly_reduce_kernel(ReduceArgs a, uint64_t out_len) {
    const void *in_v = a.in[0];
    void *out_v = a.out[0];
#pragma unroll
    for (int c = 1; c < MAX_COLS; ++c)
        if (blockIdx.y == (unsigned)c) { in_v = a.in[c]; out_v = a.out[c]; }
    const Fp *in = reinterpret_cast<const Fp *>(in_v);
    Fp *out = reinterpret_cast<Fp *>(out_v);
    for (uint64_t m = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; m < out_len;
         m += (uint64_t)gridDim.x * blockDim.x) {
        Fp v[1 << LEVELS];
#pragma unroll
        for (int k = 0; k < (1 << LEVELS); ++k) v[k] = dload(in + (m << LEVELS) + k);
        // level 0 pairs adjacent elements
#pragma unroll
        for (int k = 0; k < (1 << LEVELS); k += 2) v[k] = fp_add(v[k], fp_mul(v[k + 1], a.mult[0]));
        if (LEVELS >= 2) {
#pragma unroll
            for (int k = 0; k < (1 << LEVELS); k += 4) v[k] = fp_add(v[k], fp_mul(v[k + 2 < (1 << LEVELS) ? k + 2 : 0], a.mult[1]));
        }
        if (LEVELS >= 3) v[0] = fp_add(v[0], fp_mul(v[4 < (1 << LEVELS) ? 4 : 0], a.mult[2]));
        dstore(out + m, v[0]);
    }
}

hipError_t launch_poly_reduce(hipStream_t st, const void *const *in, void *const *out, uint32_t ncols,
                              uint64_t in_len, uint32_t levels, const Fp *mult) {
    ReduceArgs a;
    for (int c = 0; c < MAX_COLS; ++c) { a.in[c] = c < (int)ncols ? (const Fp *)in[c] : nullptr; a.out[c] = c < (int)ncols ? (Fp *)out[c] : nullptr; }
    for (int k = 0; k < 3; ++k) a.mult[k] = k < (int)levels ? mult[k] : fp_zero();
    const uint64_t out_len = in_len >> levels;
    uint32_t gx = (uint32_t)((out_len + 255) / 256);
    if (gx > 65536) gx = 65536;
    if (gx == 0) gx = 1;
    dim3 grid(gx, ncols), block(256);
    if (levels == 3) hipLaunchKernelGGL(poly_reduce_kernel<3>, grid, block, 0, st, a, out_len);
    else if (levels == 2) hipLaunchKernelGGL(poly_reduce_kernel<2>, grid, block, 0, st, a, out_len);
    else hipLaunchKernelGGL(poly_reduce_kernel<1>, grid, block, 0, st, a, out_len);
    return hipGetLastError();
}

// ---- out-of-domain evaluation at FEW points: T_c(z w^k) for the k of a column's mask cells ------------------------
// A column's cells need its polynomial at |S| << n points x_k = z w_n^k.  A coset transform (round 2: one size-n transform
// per column, 12 n butterflies) computes all n of them.  Instead: split P(x) = sum_{r < R} x^r P_r(x^R), R = n / m, m = 2^S.
// In the bit-reversed coefficient array block b = [b m, (b+1) m) IS P_r for r = bitrev(b) (its coefficients, bit-reversed),
// and x_k^R = z^R w_m^k only depends on k mod m, so
//   (A) ood_blocks_kernel: one lane per block runs the first S stages of the forward network on its m coefficients in
//       registers (the size-m coset transform, offset z^R: S m / 2 butterflies with wave-uniform twiddles) and keeps the
//       outputs whose residue some cell needs: Y[res][b] = P_bitrev(b)(z^R w_m^res);
//   (B) ood_fold_kernel: X[k] = sum_b x_k^bitrev(b) Y[k mod m][b] is the bit-reversed Horner tree of poly_reduce_kernel over
//       the block index, 2^LEVELS elements per lane and launch, as ONE fused dot product per point (fl252.h FlWide: the
//       2^LEVELS coefficients prod_i mult_i^(t_i) come from the host in R280 form); points that read the same array share
//       the loads.
// Work per column: S n / 2 butterflies + 1.07 |S| n / m fused terms, against 12 n butterflies: 5.7x less over the starknet
// mask (269 cells: |S| = 16, 5, 4, 9, 2, 60, 4, 56, 105, 8 per column), and no per-proof twiddle plan of n entries.
// butterfly IDX = (stage u, pair pr) of the 2^S-point network, unrolled by recursion: the element indices must be compile-time
// constants for x[] to live in registers (a `#pragma unroll` loop around 32 inlined multiplications is not unrolled - the array
// then sits in scratch, 5x slower)
template <int S, int IDX>
__device__ __forceinline__ void ood_butterflies(Fl (&x)[1 << S], const OodBlockArgs &a) {
    constexpr int HALF = (1 << S) / 2;
    if constexpr (IDX < S * HALF) {
        constexpr int u = IDX / HALF, pr = IDX % HALF;
        constexpr int j = pr & ((1 << u) - 1), lo = ((pr >> u) << (u + 1)) | j, hi = lo | (1 << u);
        Fl t;
#pragma unroll
        for (int i = 0; i < 9; ++i) t.l[i] = a.tw[(1 << u) - 1 + j][i];
        const Fl bt = fl_mul_r280(x[hi], t);              // DIT: a' = a + b t, b' = a - b t + 2p (ntt.hip radix_stage: <= 11 stages lazily)
        const Fl lo_v = x[lo];
        x[lo] = fl_add(lo_v, bt);
        x[hi] = fl_sub_c<2, 1>(lo_v, bt);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);                // one multiplication at a time: interleaved, their 19-column accumulators cost 130 registers
        ood_butterflies<S, IDX + 1>(x, a);
    }
}
template <int S, int J>
__device__ __forceinline__ void ood_store_residues(const Fl (&x)[1 << S], const OodBlockArgs &a, uint64_t b, uint32_t slot) {
    if constexpr (J < (1 << S)) {
        if ((a.res_mask >> J) & 1u) {
            dstore(a.out + (uint64_t)slot * a.blocks + b, fl_pack(fl_weak_reduce(x[J])));     // < 2^252, what the fold re-limbs
            ++slot;
        }
        ood_store_residues<S, J + 1>(x, a, b, slot);
    }
}
template <int S, int T>
__device__ __forceinline__ void ood_load_block(Fl (&x)[1 << S], const Fp *p) {
    if constexpr (T < (1 << S)) {
        x[T] = fl_from_fp(dload(p + T));
        ood_load_block<S, T + 1>(x, p);
    }
}

template <int S>
__global__ __launch_bounds__(128, 2) void ood_blocks_kernel(OodBlockArgs a) {       // two waves per SIMD: <= 256 registers
    constexpr int M = 1 << S;
    const uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (b >= a.blocks) return;
    Fl x[M];
    ood_load_block<S, 0>(x, a.coeffs + b * M);
    ood_butterflies<S, 0>(x, a);
    ood_store_residues<S, 0>(x, a, b, 0u);
}

hipError_t launch_ood_blocks(hipStream_t st, int S, const OodBlockArgs &a) {
    const dim3 grid((uint32_t)((a.blocks + 127) / 128)), block(128);
    if (S == 4) hipLaunchKernelGGL(ood_blocks_kernel<4>, grid, block, 0, st, a);
    else if (S == 3) hipLaunchKernelGGL(ood_blocks_kernel<3>, grid, block, 0, st, a);
    else if (S == 2) hipLaunchKernelGGL(ood_blocks_kernel<2>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(ood_blocks_kernel<1>, grid, block, 0, st, a);
    return hipGetLastError();
}

// one array (grid.y) folded for each of its points: out_p[q] = sum_{t < 2^LEVELS} in[(q << LEVELS) + t] * coef_p[t].  The points'
// coefficients sit in LDS (wave-uniform reads: broadcasts) - read from global memory inside the loop every term waited on its
// own load; the 2^LEVELS values of a lane stay in registers across the array's points.
template <int LEVELS, int T>
__device__ __forceinline__ void ood_load_group(Fl (&v)[1 << LEVELS], const Fp *p) {
    if constexpr (T < (1 << LEVELS)) {
        v[T] = fl_from_fp(dload(p + T));
        ood_load_group<LEVELS, T + 1>(v, p);
    }
}
typedef u32 __attribute__((address_space(3))) *ood_lds_u32;
template <int LEVELS, int T>
__device__ __forceinline__ void ood_dot(FlWide &w, const Fl (&v)[1 << LEVELS], const ood_lds_u32 coef) {
    if constexpr (T < (1 << LEVELS)) {
        Fl c;
#pragma unroll
        for (int i = 0; i < 9; ++i) c.l[i] = coef[9 * T + i];
        fl_wide_mad(w, v[T], c);
        asm volatile("" ::: "memory");                    // keep the coefficient reads next to their use (hoisted, they are 144 registers)
        __builtin_amdgcn_sched_barrier(0);
        ood_dot<LEVELS, T + 1>(w, v, coef);
    }
}
template <int LEVELS>
__global__ __launch_bounds__(128, 2) void ood_fold_kernel(const OodFoldArray *__restrict__ arrays, const OodFoldPoint *__restrict__ points,
                                                       uint64_t out_len, int canonical) {
    constexpr int T = 1 << LEVELS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const ood_lds_u32 lds = (ood_lds_u32)smem;                                    // [npoints][T][9]
    const OodFoldArray arr = arrays[blockIdx.y];
    for (uint32_t idx = threadIdx.x; idx < arr.npoints * (uint32_t)(T * 9); idx += blockDim.x) {
        const uint32_t k = idx / (uint32_t)(T * 9), rem = idx % (uint32_t)(T * 9);
        lds[idx] = points[arr.first_point + k].coef[rem / 9u][rem % 9u];
    }
    __syncthreads();
    for (uint64_t q = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; q < out_len; q += (uint64_t)gridDim.x * blockDim.x) {
        Fl v[T];
        ood_load_group<LEVELS, 0>(v, arr.in + (q << LEVELS));
#pragma unroll 1
        for (uint32_t k = 0; k < arr.npoints; ++k) {
            FlWide w;
            fl_wide_zero(w);
            ood_dot<LEVELS, 0>(w, v, lds + k * (uint32_t)(T * 9));
            const Fl r = fl_wide_reduce(w);                  // normalised, < 1.01 p
            dstore(points[arr.first_point + k].out + q, canonical ? fl_to_fp(r) : fl_pack(r));
        }
    }
}

hipError_t launch_ood_fold(hipStream_t st, const OodFoldArray *d_arrays, uint32_t narrays, uint32_t max_points, const OodFoldPoint *d_points,
                           uint64_t in_len, uint32_t levels, bool canonical) {
    const uint64_t out_len = in_len >> levels;
    uint32_t gx = (uint32_t)((out_len + 127) / 128);
    if (gx > 16384) gx = 16384;
    if (gx == 0) gx = 1;
    const dim3 grid(gx, narrays), block(128);
    const int cn = canonical ? 1 : 0;
    const size_t lds = (size_t)max_points * (36u << levels);                          // the coefficients of one array's points
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    if (levels == 4) hipLaunchKernelGGL(ood_fold_kernel<4>, grid, block, lds, st, d_arrays, d_points, out_len, cn);
    else if (levels == 3) hipLaunchKernelGGL(ood_fold_kernel<3>, grid, block, lds, st, d_arrays, d_points, out_len, cn);
    else if (levels == 2) hipLaunchKernelGGL(ood_fold_kernel<2>, grid, block, lds, st, d_arrays, d_points, out_len, cn);
    else hipLaunchKernelGGL(ood_fold_kernel<1>, grid, block, lds, st, d_arrays, d_points, out_len, cn);
    return hipGetLastError();
}

// ---- D[i] = 1 / (offset * w^i - z), i < 2^log_N ----------------------------------------
// One lane per chunk of CH elements: a forward pass stores the prefix products in D, one inversion, a backward pass overwrites
// them with the inverses.  Chunk c is the elements c, c + m, c + 2 m, ... (m chunks): the lanes of a wave touch 64 neighbouring
// elements in every step (2 KiB runs; with contiguous chunks every lane had a cache line of its own - 4 KiB apart - and the
// kernel ran at the memory system's pace, not the multiplier's), and the lane's points are a geometric sequence of ratio w^m.
// r280 != 0: the table is written in R280 form (fl252.h: entries times 2^24, for fl_mul_r280 consumers);
// the factor rides on the running inverse, so it costs one multiplication per chunk.
__global__ __launch_bounds__(128) void batch_inverse_kernel(Fp *__restrict__ D, uint64_t nchunks, uint32_t log_chunk,
                                                            Fp offset, Fp w, Fp step, Fp step_inv, Fp z, int r280) {
    const uint64_t c = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    const uint64_t CH = 1ull << log_chunk;
    const Fl sl = fl_from_fp(step), sil = fl_from_fp(step_inv), zl = fl_from_fp(z);
    Fl x = fl_from_fp(fp_mul(offset, fp_pow_u64(w, c)));
    Fl run = fl_one();
    for (uint64_t k = 0; k < CH; ++k) {
        dstore(D + c + k * nchunks, fl_pack(run));   // prefix product of the chunk's d_0 .. d_{k-1} (weakly reduced image)
        run = fl_mul(fl_sub_c<2, 1>(x, zl), run);
        x = fl_mul(x, sl);
    }
    Fl inv = fl_from_fp(fp_inv_safegcd(fl_to_fp(fl_weak_reduce(run))));   // 1 / (d_0 ... d_{CH-1}); 0 stays 0 if some x_i == z
    if (r280) { Fp two24 = fp_zero(); two24.v[0] = 1u << 24; inv = fl_mul(inv, fl_from_fp(fp_to_mont(two24))); }
    for (uint64_t k = CH; k-- > 0;) {
        x = fl_mul(x, sil);                          // the chunk's point k
        const Fl pre = fl_from_fp(dload(D + c + k * nchunks));
        dstore(D + c + k * nchunks, fl_to_fp(fl_mul(inv, pre)));
        inv = fl_mul(fl_sub_c<2, 1>(x, zl), inv);
    }
}

static uint32_t batch_inverse_log_chunk(uint32_t log_N) {
    // enough lanes to fill the chip, chunks long enough to amortise the inversion (SS_BATCH_INV_LOG_CHUNK: tuning only)
    uint32_t log_chunk = log_N > 17 ? log_N - 17 : 0;
    if (log_chunk < 4) log_chunk = log_N < 4 ? log_N : 4;
    if (log_chunk > 7) log_chunk = 7;
    static const char *env = getenv("SS_BATCH_INV_LOG_CHUNK");
    if (env) { const uint32_t v = (uint32_t)atoi(env); if (v >= 1 && v <= 10 && v <= log_N) log_chunk = v; }
    return log_chunk;
}

hipError_t launch_batch_inverse(hipStream_t st, Fp *D, uint32_t log_N, const Fp &offset, const Fp &w,
                                const Fp &w_inv, const Fp &z, bool r280) {
    const uint32_t log_chunk = batch_inverse_log_chunk(log_N);
    const uint64_t nchunks = 1ull << (log_N - log_chunk);
    hipLaunchKernelGGL(batch_inverse_kernel, dim3((uint32_t)((nchunks + 127) / 128)), dim3(128), 0, st, D, nchunks,
                       log_chunk, offset, w, fp_pow_u64(w, nchunks), fp_pow_u64(w_inv, nchunks), z, r280 ? 1 : 0);
    return hipGetLastError();
}

// D[j] = 1 / (x0 * w^j - z) for j < len: a RANGE of the domain (the row-block form of the sharded prover).  len must be a
// multiple of 2^BATCH_INVERSE_RANGE_LOG_CHUNK; the entries are the whole-domain table's, bit for bit (every entry is
// the fully reduced inverse, whatever the chunking).
hipError_t launch_batch_inverse_range(hipStream_t st, Fp *D, uint64_t len, const Fp &x0, const Fp &w, const Fp &w_inv,
                                      const Fp &z, bool r280) {
    const uint32_t log_chunk = BATCH_INVERSE_RANGE_LOG_CHUNK;
    const uint64_t nchunks = len >> log_chunk;
    if (nchunks == 0) return hipSuccess;
    hipLaunchKernelGGL(batch_inverse_kernel, dim3((uint32_t)((nchunks + 127) / 128)), dim3(128), 0, st, D, nchunks,
                       log_chunk, x0, w, fp_pow_u64(w, nchunks), fp_pow_u64(w_inv, nchunks), z, r280 ? 1 : 0);
    return hipGetLastError();
}

// ---- DEEP composition -----------------------------------------------------------------
struct DeepArgs {
    const Fp *trace[MAX_COLS];
    const Fp *comp[4];
    const Fp *D;            // 1/(x_i - z), R280 form
    const Fp *Dc;           // 1/(x_i - z^ncomp), R280 form
    const uint32_t *tap_shift;   // [ntaps] sorted by column, then by shift (= off * blowup / stride: sub-coset units)
    const Fp *tap_coef;          // [ntaps] coeff_j * w_n^-off_j in R280 form (times 2^24); the constant column: -K_g, Montgomery form
    const uint32_t *col_desc;    // [ncoldesc][3]: trace column (0xffffffff: the constant column), first tap, tap count
    const Fp *comp_coef;         // [ncomp], R280 form
    Fp comp_k;                   // sum_k cc_k * ood_comp_k
    uint32_t ncoldesc, ncomp, log_stride;
    uint64_t count;              // points evaluated: sub-coset points m0 .. m0 + count (m0 is folded into the pointers)
    uint32_t d_bias, d_mask;     // D is read at (m + d_bias - shift) & d_mask: whole table (0, M - 1) or a local range with
                                 // `d_bias` entries in front of the block's first point (d_mask = ~0)
    uint32_t xcd_map;            // 1: workgroup b sweeps the (b % 8)-th eighth of the points (see the kernel)
};

// Evaluates the DEEP sum at LDE indices i = m * stride, m < count: with stride = blowup this is the trace-size sub-coset
// offset*<w_n>, enough to pin the degree < n DEEP polynomial, which the caller then interpolates and re-expands (halves
// the pointwise work).  D / Dc are tables over that same sub-coset; shifts are in sub-coset units.
//
// A cell's trace value is T_col[i] whatever its offset - only the denominator moves - so the sum is taken column by column:
//   out[m] = sum_c T_c[i] * S_c[m] + S_K[m] + Dc[m] * (sum_k cc_k H_k[i] - Kc),
//   S_c[m] = sum_{cells (c, off)} c'_(c,off) D[m - shift_off],     S_K[m] = sum_off (-K_off) D[m - shift_off]
// (K_off = sum of c' * ood over the cells of that offset).  Every S is ONE fused dot product (fl252.h FlWide: 81 partial
// products per term into 64-bit columns, one Montgomery reduction per <= 16 terms), and so is the outer sum over the
// columns.  Grouping by offset instead - one inner sum and one reduction per distinct offset - paid 191 + 12 reductions per
// point for the starknet mask (269 cells at 191 offsets) against 269 + 191 products; this order pays ~ 40.
// Wave-uniform, read-only tables (shifts, coefficients, descriptors) are read through the constant address space:
// scalar loads into SGPRs, and the coefficient's re-limbing runs on the scalar unit.  The denominator table is read
// at ~ 460 shifted positions per point; workgroup b runs on XCD b % 8 (observed; speed only) and sweeps that XCD's own
// eighth of the points front to back, so those reads stay in one L2.
typedef uint32_t dk_u32x4 __attribute__((ext_vector_type(4)));
typedef const uint32_t __attribute__((address_space(4))) *dk_const_u32;
typedef const dk_u32x4 __attribute__((address_space(4))) *dk_const_u32x4;
__device__ __forceinline__ Fp dload_uniform(const Fp *p) {
    dk_const_u32x4 q = (dk_const_u32x4)(uintptr_t)p;
    const dk_u32x4 a = q[0], b = q[1];
    Fp r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}

__global__ __launch_bounds__(256) void deep_kernel(DeepArgs a, Fp *__restrict__ out) {
    const uint64_t M = a.count;                      // points evaluated
    dk_const_u32 col_desc = (dk_const_u32)(uintptr_t)a.col_desc, tap_shift = (dk_const_u32)(uintptr_t)a.tap_shift;
    // point range of this workgroup: XCD-contiguous when the points split evenly over 8 x gridDim/8 workgroups
    uint64_t m_first, m_end, m_step;
    if (a.xcd_map) {
        const uint64_t chunk = M >> 3;
        m_first = (uint64_t)(blockIdx.x & 7u) * chunk + (uint64_t)(blockIdx.x >> 3) * blockDim.x;
        m_end = (uint64_t)((blockIdx.x & 7u) + 1u) * chunk;
        m_step = (uint64_t)(gridDim.x >> 3) * blockDim.x;
    } else {
        m_first = blockIdx.x * (uint64_t)blockDim.x;
        m_end = M;
        m_step = (uint64_t)gridDim.x * blockDim.x;
    }
    for (uint64_t m = m_first + threadIdx.x; m < m_end; m += m_step) {
        const uint64_t i = m << a.log_stride;        // LDE row of this point
        const uint32_t dbase = (uint32_t)m + a.d_bias;
        Fl acc = fl_zero();                          // reduced partial sums (lazy adds): < 32 p at the end for fl_to_fp
        uint32_t acc_terms = 0;
        FlWide wo;                                   // outer: sum_c T_c * S_c
        fl_wide_zero(wo);
        uint32_t outer_terms = 0;
        for (uint32_t k = 0; k < a.ncoldesc; ++k) {
            const uint32_t col = col_desc[3 * k], first = col_desc[3 * k + 1], cnt = col_desc[3 * k + 2];
            Fl part = fl_zero();                     // reduced partial sums of S (each < 1.01 p)
            uint32_t parts = 0, terms = 0;
            FlWide wi;
            fl_wide_zero(wi);
            Fp d = dload(a.D + ((dbase - tap_shift[first]) & a.d_mask));
            for (uint32_t j = first; j < first + cnt; ++j) {
                const Fl dl = fl_from_fp(d);         // canonical image from memory: normalised limbs
                if (j + 1 < first + cnt) d = dload(a.D + ((dbase - tap_shift[j + 1]) & a.d_mask));     // next tap in flight
                fl_wide_mad(wi, dl, fl_from_fp(dload_uniform(a.tap_coef + j)));
                if (++terms == (uint32_t)FL_WIDE_MAX_TERMS) {
                    part = fl_add(part, fl_wide_reduce(wi));
                    fl_wide_zero(wi);
                    terms = 0;
                    if (++parts == 6) { part = fl_weak_reduce(part); parts = 1; }
                }
            }
            if (terms) part = fl_add(part, fl_wide_reduce(wi));
            const Fl S = fl_weak_reduce(part);       // <= 7 reduced values -> < 2p, normalised
            if (col == 0xffffffffu) {                // the constant column: S_K is a Montgomery image, added as it is
                acc = fl_add(acc, S);
                ++acc_terms;
            } else {
                const Fp *tp = a.trace[0];
#pragma unroll
                for (int c = 1; c < MAX_COLS; ++c) if (col == (uint32_t)c) tp = a.trace[c];
                fl_wide_mad(wo, fl_from_fp(dload(tp + i)), S);
                if (++outer_terms == (uint32_t)FL_WIDE_MAX_TERMS) {
                    acc = fl_add(acc, fl_wide_reduce(wo));
                    fl_wide_zero(wo);
                    outer_terms = 0;
                    ++acc_terms;
                }
            }
            if (acc_terms >= 6) { acc = fl_weak_reduce(acc); acc_terms = 1; }
        }
        if (a.ncomp) {
            FlWide wi;
            fl_wide_zero(wi);
            for (uint32_t k = 0; k < a.ncomp; ++k) {
                const Fp *hp = a.comp[0];
#pragma unroll
                for (int c = 1; c < 4; ++c) if (k == (uint32_t)c) hp = a.comp[c];
                fl_wide_mad(wi, fl_from_fp(dload(hp + i)), fl_from_fp(dload_uniform(a.comp_coef + k)));
            }
            const Fl inner = fl_weak_reduce(fl_sub_c<2, 1>(fl_wide_reduce(wi), fl_from_fp(a.comp_k)));
            fl_wide_mad(wo, inner, fl_from_fp(dload(a.Dc + m)));
            ++outer_terms;
        }
        if (outer_terms) acc = fl_add(acc, fl_wide_reduce(wo));      // <= 6 + 1 values < 2p each: < 32 p for fl_to_fp
        dstore(out + m, fl_to_fp(acc));
    }
}

hipError_t launch_deep(hipStream_t st, const void *const *trace, uint32_t ntrace, const void *const *comp,
                       uint32_t ncomp, const Fp *D, const Fp *Dc, const uint32_t *tap_shift, const Fp *tap_coef,
                       const uint32_t *col_desc, uint32_t ncoldesc, const Fp *comp_coef,
                       const Fp &comp_k, uint64_t count, uint32_t d_bias, uint32_t d_mask, uint32_t log_stride, Fp *out) {
    DeepArgs a;
    for (int c = 0; c < MAX_COLS; ++c) a.trace[c] = c < (int)ntrace ? (const Fp *)trace[c] : nullptr;
    for (int c = 0; c < 4; ++c) a.comp[c] = c < (int)ncomp ? (const Fp *)comp[c] : nullptr;
    a.D = D; a.Dc = Dc; a.tap_shift = tap_shift; a.tap_coef = tap_coef; a.col_desc = col_desc; a.ncoldesc = ncoldesc;
    a.comp_coef = comp_coef; a.comp_k = comp_k; a.ncomp = ncomp;
    a.count = count; a.d_bias = d_bias; a.d_mask = d_mask; a.log_stride = log_stride;
    const uint64_t N = count;
    uint32_t gx = (uint32_t)((N + 255) / 256);
    static const uint32_t wgs = getenv("SS_DEEP_WGS") ? (uint32_t)atoi(getenv("SS_DEEP_WGS")) : 256 * 16;    // measured: 512 -> 44 ms, 1024 -> 36, 4096 -> 32 (starknet_2p20)
    if (gx > wgs) gx = wgs;
    static const bool no_xcd_map = getenv("SS_DEEP_NO_XCD_MAP") != nullptr;                                // A/B switch for profiling
    a.xcd_map = (!no_xcd_map && gx >= 8 && gx % 8 == 0 && N % (8 * 256) == 0) ? 1u : 0u;
    hipLaunchKernelGGL(deep_kernel, dim3(gx), dim3(256), 0, st, a, out);
    return hipGetLastError();
}

// ---- the mask's LARGE columns as rational functions ------------------------------------------------------------------------
// A column with many cells (starknet: 105, 60 and 56 of the 269) and the constants' column (one tap per distinct offset: 191)
// pay a tap each per point above.  But sum_{cells j of c} c_j / (x - z w^off_j) = A_c(x) / B(x) with B = prod_off (x - z w^off)
// over the mask's distinct offsets and deg A_c < deg B <= a few hundred: the host builds the coefficients (O(offsets x cells)
// products), ONE pruned transform evaluates each polynomial on the sub-coset (its first stages see zeros: log_expand), B is
// inverted in batches, and a point pays ONE product per such column:
//     out[m] += (sum_c T_c[i] A_c[m] - A_K[m]) / B[m],      A_K = sum_off K_off B / (x - z w^off)
// A_c comes scaled by 2^24 and B by 2^-24 from the host, so both multipliers are in R280 form as they leave the transform / the
// inversion.  The field arithmetic is exact: the same values as the tap sums, bit for bit.
__global__ __launch_bounds__(128) void batch_inverse_values_kernel(Fp *__restrict__ V, Fp *__restrict__ tmp, uint64_t nchunks, uint32_t log_chunk) {
    const uint64_t c = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    // chunk c = elements c, c + nchunks, c + 2 nchunks, ...: the lanes of a wave read neighbours
    const uint64_t CH = 1ull << log_chunk;
    Fl run = fl_one();
    for (uint64_t k = 0; k < CH; ++k) {
        const uint64_t i = c + k * nchunks;
        Fl v = fl_from_fp(dload(V + i));
        if (fn_is_zero(v)) v = fl_one();
        dstore(tmp + i, fl_pack(run));
        run = fn_mul(run, v);
    }
    Fl inv = fl_from_fp(fp_inv_safegcd(fl_to_fp(run)));
    for (uint64_t k = CH; k-- > 0;) {
        const uint64_t i = c + k * nchunks;
        Fl v = fl_from_fp(dload(V + i));
        const bool zero = fn_is_zero(v);
        if (zero) v = fl_one();
        const Fl r = fn_mul(inv, fl_from_fp(dload(tmp + i)));
        inv = fn_mul(inv, v);
        dstore(V + i, zero ? fp_zero() : fl_to_fp(r));
    }
}
hipError_t launch_batch_inverse_values(hipStream_t st, Fp *V, Fp *tmp, uint64_t len) {
    uint32_t log_chunk = 5;
    while (log_chunk > 0 && (len >> log_chunk) < 4096 ) --log_chunk;
    while ((len & ((1ull << log_chunk) - 1)) != 0) --log_chunk;          // len is a power of two here; any len works with chunk 1
    const uint64_t nchunks = len >> log_chunk;
    if (nchunks == 0) return hipSuccess;
    hipLaunchKernelGGL(batch_inverse_values_kernel, dim3((uint32_t)((nchunks + 127) / 128)), dim3(128), 0, st, V, tmp, nchunks, log_chunk);
    return hipGetLastError();
}

struct DeepRationalArgs {
    const Fp *trace[DEEP_RATIONAL_MAX_COLS];
    const Fp *A[DEEP_RATIONAL_MAX_COLS];
    const Fp *AK, *Binv;
    uint32_t ncols, log_stride;
    uint64_t count;
};
__global__ __launch_bounds__(256) void deep_rational_kernel(DeepRationalArgs a, Fp *__restrict__ out) {
    for (uint64_t m = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; m < a.count; m += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = m << a.log_stride;
        FlWide w;
        fl_wide_zero(w);
#pragma unroll
        for (uint32_t k = 0; k < DEEP_RATIONAL_MAX_COLS; ++k)
            if (k < a.ncols) fl_wide_mad(w, fl_from_fp(dload(a.trace[k] + i)), fl_from_fp(dload(a.A[k] + m)));
        const Fl s = fl_wide_reduce(w);                                      // normalised, < 1.01 p (0 for no columns)
        const Fl d = fl_sub_c<2, 1>(s, fl_from_fp(dload(a.AK + m)));         // lazy: < 4p
        const Fl v = fl_mul_r280(d, fl_from_fp(dload(a.Binv + m)));         // normalised, < 1.01 p
        dstore(out + m, fl_to_fp(fl_add(fl_from_fp(dload(out + m)), v)));
    }
}
hipError_t launch_deep_rational(hipStream_t st, const void *const *trace, const Fp *const *A, uint32_t ncols, const Fp *AK, const Fp *Binv,
                                uint64_t count, uint32_t log_stride, Fp *out) {
    if (ncols > DEEP_RATIONAL_MAX_COLS) return hipErrorInvalidValue;
    DeepRationalArgs a;
    for (uint32_t k = 0; k < DEEP_RATIONAL_MAX_COLS; ++k) { a.trace[k] = k < ncols ? (const Fp *)trace[k] : nullptr; a.A[k] = k < ncols ? A[k] : nullptr; }
    a.AK = AK; a.Binv = Binv; a.ncols = ncols; a.log_stride = log_stride; a.count = count;
    uint32_t gx = (uint32_t)((count + 255) / 256);
    if (gx > 256 * 16) gx = 256 * 16;
    if (gx == 0) gx = 1;
    hipLaunchKernelGGL(deep_rational_kernel, dim3(gx), dim3(256), 0, st, a, out);
    return hipGetLastError();
}

// gather with a column selector: out[j] = cols[col[j]][idx[j]]
struct GatherArgs { const Fp *cols[MAX_COLS]; };
__global__ void gather_cells_kernel(GatherArgs a, const uint32_t *__restrict__ col, const uint64_t *__restrict__ idx,
                                    uint32_t n, Fp *__restrict__ out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t cc = col[j];
    const Fp *p = a.cols[0];
#pragma unroll
    for (int c = 1; c < MAX_COLS; ++c) if (cc == (uint32_t)c) p = a.cols[c];
    dstore(out + j, dload(p + idx[j]));
}
hipError_t launch_gather_cells(hipStream_t st, const void *const *cols, uint32_t ncols, const uint32_t *col,
                               const uint64_t *idx, uint32_t n, Fp *out) {
    if (n == 0) return hipSuccess;
    GatherArgs a;
    for (int c = 0; c < MAX_COLS; ++c) a.cols[c] = c < (int)ncols ? (const Fp *)cols[c] : nullptr;
    hipLaunchKernelGGL(gather_cells_kernel, dim3((n + 127) / 128), dim3(128), 0, st, a, col, idx, n, out);
    return hipGetLastError();
}

}  // namespace ss
