// fri.hip — one coset-FRI layer fold on gfx950.
//
// Replaces the per-layer fold of ministark's FriProver::build_layers
// (un-vendored; options fri_folding_factor = 8 by default, cli/src/main.rs:57-58;
// row F1 of SURVEY.md §8a).  For row j of the layer (the `fold` evaluations at
// x_j * w_fold^k, stored at evals[j + k * len/fold]) the folded value is the
// degree < fold interpolant evaluated at alpha:
//     out[j] = (1/fold) * sum_m (alpha / x_j)^m * sum_k evals[j + k*rows] * w_fold^(-k m)
// i.e. a size-`fold` inverse NTT held in registers followed by Horner.
// One lane = one row; lane j reads evals[j + k*rows] for each k, so every load
// instruction of a wave is a contiguous 2 KiB run.
//
// Two conventions, both pinned by proof files the reference ships (tests/golden/make_fri_golden.py):
//   flags = 0                    natural order, normalised - the older code path's proofs
//   SS_FRI_BITREV_ROWS           the vector is in bit-reversed order: row r = evals[fold r .. fold r + fold),
//                                entry j of a row at x_r w_fold^bitrev(j), x_r = offset w^bitrev(r); the output is
//                                the next layer in bit-reversed order again
//   SS_FRI_UNNORMALISED          out = fold * interpolant(alpha): StarkWare's fold, no 1/2 per halving
// BITREV_ROWS | UNNORMALISED is what the current code path's proofs contain.
#include <hip/hip_runtime.h>
#include "../../include/sandstorm_hip.h"
#include "fp252.h"
#include "fl252.h"
#include "kernels.h"

namespace ss {

struct FriConsts {
    Fp alpha_over_offset, w_inv;        // alpha / offset; w_len^-1
    Fp tw_inv[8];  // w_fold^(-k), k < fold/2, in R280 form (times 2^24)
    // w_len^-e from two small tables (a lane's 1 / x_j is one product instead of a 22-bit power: 33 products, twice the fold's own):
    // pow_lo[k] = w^-k, k < 2^lo_bits; pow_hi[k] = w^-(k << lo_bits)
    const Fp *pow_lo, *pow_hi;
    uint32_t lo_bits;
    uint32_t flags;
};

__device__ __forceinline__ Fp fri_load(const Fp *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    Fp r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__device__ __forceinline__ void fri_store(Fp *p, const Fp &x) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    q[1] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
}

__host__ __device__ constexpr int brev_c(int m, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((m >> i) & 1) << (bits - 1 - i);
    return r;
}

// The fold runs in the hot-loop limb form (fl252.h: a row's 14 products at the 9 x 28-bit multiplier's rate, twice the 8 x 32
// interchange form's - round 6); sums and differences are weakly reduced (normalised, < 2p), the stage twiddles come in R280 form.
template <int LOGF, int ST>
__device__ __forceinline__ void fri_stage(Fl (&v)[1 << LOGF], const Fl (&tw)[8]) {
    if (ST >= LOGF) return;
    constexpr int STC = ST < LOGF ? ST : 0;
    constexpr int half = 1 << STC;
#pragma unroll
    for (int pr = 0; pr < (1 << LOGF) / 2; ++pr) {
        const int i = ((pr >> STC) << (STC + 1)) | (pr & (half - 1));
        const int t = (pr & (half - 1)) << (LOGF - 1 - STC);   // exponent of w_fold^-1
        const Fl a = v[i], b = v[i | half];
        v[i] = fn_add(a, b);
        const Fl d = fl_sub_c<2, 1>(a, b);                      // lazy (< 4p): a multiplier's operand, or reduced below
        v[i | half] = t == 0 ? fl_weak_reduce(d) : fl_mul_r280(d, tw[t]);
    }
}

template <int LOGF>
__global__ __launch_bounds__(128) void fri_fold_kernel(const Fp *__restrict__ evals, uint32_t log_len,
                                                       FriConsts c, Fp *__restrict__ out, uint64_t row0, uint64_t rows) {
    // rows row0 .. row0 + rows of the layer, entry k of row row0 + i at evals[i + k * rows] (the whole layer: row0 = 0, rows = len / fold)
    constexpr int F = 1 << LOGF;
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const uint64_t j = row0 + i;
    const bool bitrev_rows = (c.flags & SS_FRI_BITREV_ROWS) != 0;
    Fl v[F];                                         // v[k] = f(x_j w_fold^k)
    if (bitrev_rows) {
#pragma unroll
        for (int k = 0; k < F; ++k) v[k] = fl_from_fp(fri_load(evals + i * F + brev_c(k, LOGF)));
    } else {
#pragma unroll
        for (int k = 0; k < F; ++k) v[k] = fl_from_fp(fri_load(evals + i + (uint64_t)k * rows));
    }
    Fl tw[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) tw[k] = fl_from_fp(c.tw_inv[k]);
    // unnormalised inverse NTT, DIF: natural in, bit-reversed out
    if (LOGF >= 4) fri_stage<LOGF, 3>(v, tw);
    if (LOGF >= 3) fri_stage<LOGF, 2>(v, tw);
    if (LOGF >= 2) fri_stage<LOGF, 1>(v, tw);
    fri_stage<LOGF, 0>(v, tw);
    // t = alpha / x_j,  1/x_j = offset^-1 * w^-e,  e = j (natural) or bitrev(j) over the log2(rows) row bits
    const uint32_t row_bits = log_len - LOGF;
    const uint64_t e = bitrev_rows ? (row_bits ? (uint64_t)(__brevll(j) >> (64u - row_bits)) : 0ull) : j;
    const Fl ao = fl_from_fp(c.alpha_over_offset);
    const Fl t = c.pow_lo ? fl_mul(fl_mul(ao, fl_from_fp(fri_load(c.pow_lo + (e & ((1ull << c.lo_bits) - 1ull))))), fl_from_fp(fri_load(c.pow_hi + (e >> c.lo_bits))))
                          : fl_mul(ao, fl_from_fp(fp_pow_u64(c.w_inv, e)));
    // Horner over natural-order coefficients c_m = v[bitrev(m)] (t: a product, normalised; the running sum one addition lazy)
    Fl acc = v[brev_c(F - 1, LOGF)];
#pragma unroll
    for (int m = F - 2; m >= 0; --m) acc = fl_add(fl_mul(acc, t), v[brev_c(m, LOGF)]);
    // the butterflies above are an unnormalised inverse NTT: acc = fold * interpolant(alpha)
    const Fp r = fl_to_fp(acc);
    fri_store(out + i, (c.flags & SS_FRI_UNNORMALISED) ? r : fp_div_pow2(r, LOGF));
}

// tab[k] = w_inv^k for k < 2^lo_bits, then tab[2^lo_bits + k] = w_inv^(k << lo_bits) for k < 2^hi_bits (once per layer size and
// context: the powers depend on the domain alone)
__global__ __launch_bounds__(256) void fri_pow_table_kernel(Fp *__restrict__ tab, Fp w_inv, uint32_t lo_bits, uint32_t hi_bits) {
    const uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, nlo = 1ull << lo_bits, nhi = 1ull << hi_bits;
    if (k >= nlo + nhi) return;
    fri_store(tab + k, fp_pow_u64(w_inv, k < nlo ? k : (k - nlo) << lo_bits));
}
hipError_t launch_fri_pow_table(hipStream_t st, Fp *tab, const Fp &w_inv, uint32_t lo_bits, uint32_t hi_bits) {
    const uint64_t n = (1ull << lo_bits) + (1ull << hi_bits);
    hipLaunchKernelGGL(fri_pow_table_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, tab, w_inv, lo_bits, hi_bits);
    return hipGetLastError();
}

hipError_t launch_fri_fold(hipStream_t st, const Fp *evals, uint32_t log_len, uint32_t log_fold,
                           const Fp &alpha, const Fp &offset_inv, const Fp &w_inv, const Fp *fold_tw_inv,
                           uint32_t flags, Fp *out, uint64_t row0, uint64_t count, const Fp *pow_tab, uint32_t lo_bits) {
    FriConsts c;
    c.flags = flags;
    c.alpha_over_offset = fp_mul(alpha, offset_inv); c.w_inv = w_inv;
    c.pow_lo = pow_tab; c.pow_hi = pow_tab ? pow_tab + (1ull << lo_bits) : nullptr; c.lo_bits = lo_bits;
    // the stage twiddles in R280 form (times 2^24: the butterflies' fl_mul_r280)
    Fp two24 = fp_zero(); two24.v[0] = 1u << 24;
    const Fp k24 = fp_to_mont(two24);
    for (int k = 0; k < 8; ++k) c.tw_inv[k] = k < (1 << log_fold) / 2 ? fp_mul(fold_tw_inv[k], k24) : fp_zero();
    const uint64_t rows = count;
    dim3 grid((uint32_t)((rows + 127) / 128)), block(128);
    switch (log_fold) {
    case 1: hipLaunchKernelGGL(fri_fold_kernel<1>, grid, block, 0, st, evals, log_len, c, out, row0, rows); break;
    case 2: hipLaunchKernelGGL(fri_fold_kernel<2>, grid, block, 0, st, evals, log_len, c, out, row0, rows); break;
    case 3: hipLaunchKernelGGL(fri_fold_kernel<3>, grid, block, 0, st, evals, log_len, c, out, row0, rows); break;
    case 4: hipLaunchKernelGGL(fri_fold_kernel<4>, grid, block, 0, st, evals, log_len, c, out, row0, rows); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace ss
