// ext.hip — extension-trace scans on gfx950 ("next" row X1).
//
// Trace::build_extension_columns (layouts/src/recursive/trace.rs:699-814,
// layouts/src/starknet/trace.rs:997-1100) is three running permutation products
//     out_i = prod_{k<=i} (z - (alpha v_k + a_k)) / prod_{k<=i} (z - (alpha v'_k + a'_k))
// (memory; range check and diluted check have single-value terms z - x_k) and one affine
// recurrence (diluted-check aggregate)
//     acc_0 = 1,  acc_i = acc_{i-1} (1 + z u_i) + alpha u_i^2,  u_i = x_i - x_{i-1}.
// The reference runs them as sequential loops on the host between two device phases
// ("TODO: multithread", trace.rs:700).  Here both are inclusive scans over a monoid
// (field multiplication; composition of affine maps), done in three phases per level:
//   reduce : one lane folds a chunk of 2^k consecutive items into its aggregate
//   (recurse on the aggregates until one lane holds them all)
//   apply  : the lane replays its chunk starting from the scanned aggregate before it
// A running product takes TWO kinds of passes: perm_quotients (the terms n_k, d_k, the batched inversion of the d_k by Montgomery's
// trick - one safegcd inversion per lane's chunk, zero-preserving like ark-ff's batch_inversion - and q_k = n_k / d_k, one launch),
// then the scan of the q_k whose last level writes the strided column.  The arithmetic is exact, so the order of association
// does not change a single bit.  Cost: ~10 multiplications and ~11 x 32 B of HBM traffic per item (17 before the passes were
// fused: round 6) - microseconds next to the LDE; the point is that the columns never leave HBM between the two phases.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "ext_scan.h"
#include "kernels.h"

namespace ss {

namespace {

template <class Op>
__global__ __launch_bounds__(128) void scan_reduce_kernel(const Fp *__restrict__ data, uint64_t n, Fp *__restrict__ agg, uint32_t lc) {
    scan_reduce_lane<Op>(blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, data, n, agg, lc);
}
template <class Op>
__global__ __launch_bounds__(128) void scan_apply_kernel(Fp *__restrict__ data, uint64_t n, const Fp *__restrict__ aggscan, uint32_t lc) {
    scan_apply_lane<Op>(blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, data, n, aggscan, lc);
}
template <class Op>
__global__ __launch_bounds__(128) void scan_apply_out_kernel(const Fp *__restrict__ data, uint64_t n, const Fp *__restrict__ aggscan, uint32_t lc,
                                                             Fp *out, uint64_t out_stride, uint64_t out_off) {
    scan_apply_out_lane<Op>(blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, data, n, aggscan, lc, out, out_stride, out_off);
}
__global__ __launch_bounds__(128) void perm_quotients_kernel(PermOperand num, PermOperand den, uint64_t count, Fp z, Fp alpha,
                                                             Fp *__restrict__ q, Fp *__restrict__ tmp, uint32_t lc) {
    perm_quotients_lane(blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, num, den, count, z, alpha, q, tmp, lc);
}
__global__ __launch_bounds__(256) void dil_terms_kernel(const Fp *x, uint64_t stride, uint64_t off, uint64_t count, Fp z, Fp alpha,
                                                        Fp *__restrict__ mc) {
    dil_terms_lane(blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, x, stride, off, count, z, alpha, mc);
}

__global__ __launch_bounds__(256) void scale_strided_kernel(Fp *data, uint64_t stride, uint64_t off, uint64_t count, Fp factor) {
    scale_strided_lane(blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, data, stride, off, count, factor);
}
__global__ __launch_bounds__(256) void dil_terms_block_kernel(const Fp *x, uint64_t stride, uint64_t off, uint64_t count, bool starts_column,
                                                              Fp z, Fp alpha, Fp *__restrict__ mc) {
    dil_terms_block_lane(blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, x, stride, off, count, starts_column, z, alpha, mc);
}
__global__ __launch_bounds__(256) void affine_apply_kernel(const Fp *__restrict__ mc, uint64_t count, Fp start, Fp *out, uint64_t out_stride,
                                                           uint64_t out_off) {
    affine_apply_lane(blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, mc, count, start, out, out_stride, out_off);
}

inline dim3 grid_for(uint64_t lanes, uint32_t block) { return dim3((uint32_t)((lanes + block - 1) / block)); }

// SS_SCAN_LOG_CHUNK / SS_INV_LOG_CHUNK override the chunk sizes (tuning only)
ScanShape shape_from_env() {
    ScanShape s;
    if (const char *e = getenv("SS_SCAN_LOG_CHUNK")) { const int v = atoi(e); if (v >= (int)SCAN_MIN_LOG_CHUNK && v <= 10) s.log_scan = (uint32_t)v; }
    if (const char *e = getenv("SS_INV_LOG_CHUNK")) { const int v = atoi(e); if (v >= 1 && v <= 10) s.log_inv = (uint32_t)v; }
    return s;
}

struct HipExec {                 // a lane body = a kernel launch on the context's stream
    hipStream_t st;
    ScanShape shape;
    static int done() { return (int)hipGetLastError(); }
    template <class Op> int reduce(uint64_t lanes, const Fp *data, uint64_t n, Fp *agg, uint32_t lc) {
        hipLaunchKernelGGL(scan_reduce_kernel<Op>, grid_for(lanes, 128), dim3(128), 0, st, data, n, agg, lc);
        return done();
    }
    template <class Op> int apply(uint64_t lanes, Fp *data, uint64_t n, const Fp *aggscan, uint32_t lc) {
        hipLaunchKernelGGL(scan_apply_kernel<Op>, grid_for(lanes, 128), dim3(128), 0, st, data, n, aggscan, lc);
        return done();
    }
    template <class Op> int apply_out(uint64_t lanes, const Fp *data, uint64_t n, const Fp *aggscan, uint32_t lc, Fp *out, uint64_t out_stride, uint64_t out_off) {
        hipLaunchKernelGGL(scan_apply_out_kernel<Op>, grid_for(lanes, 128), dim3(128), 0, st, data, n, aggscan, lc, out, out_stride, out_off);
        return done();
    }
    int perm_quotients(const PermOperand &num, const PermOperand &den, uint64_t count, const Fp &z, const Fp &alpha, Fp *q, Fp *tmp, uint32_t lc) {
        hipLaunchKernelGGL(perm_quotients_kernel, grid_for(scan_chunks(count, lc), 128), dim3(128), 0, st, num, den, count, z, alpha, q, tmp, lc);
        return done();
    }
    int dil_terms(const Fp *x, uint64_t stride, uint64_t off, uint64_t count, const Fp &z, const Fp &alpha, Fp *mc) {
        hipLaunchKernelGGL(dil_terms_kernel, grid_for(count, 256), dim3(256), 0, st, x, stride, off, count, z, alpha, mc);
        return done();
    }
    int dil_terms_block(const Fp *x, uint64_t stride, uint64_t off, uint64_t count, bool starts_column, const Fp &z, const Fp &alpha, Fp *mc) {
        hipLaunchKernelGGL(dil_terms_block_kernel, grid_for(count, 256), dim3(256), 0, st, x, stride, off, count, starts_column, z, alpha, mc);
        return done();
    }
};

}  // namespace

hipError_t launch_permutation_product(hipStream_t st, const PermOperand &num, const PermOperand &den, uint64_t count,
                                      const Fp &z, const Fp &alpha, Fp *out, uint64_t out_stride, uint64_t out_off, Fp *scratch) {
    static const ScanShape shape = shape_from_env();
    HipExec ex{st, shape};
    return (hipError_t)permutation_product(ex, num, den, count, z, alpha, out, out_stride, out_off, scratch);
}

hipError_t launch_diluted_aggregate(hipStream_t st, const Fp *x, uint64_t stride, uint64_t off, uint64_t count, const Fp &z,
                                    const Fp &alpha, Fp *out, uint64_t out_stride, uint64_t out_off, Fp *scratch) {
    static const ScanShape shape = shape_from_env();
    HipExec ex{st, shape};
    return (hipError_t)diluted_aggregate(ex, x, stride, off, count, z, alpha, out, out_stride, out_off, scratch);
}

hipError_t launch_scale_strided(hipStream_t st, Fp *data, uint64_t stride, uint64_t off, uint64_t count, const Fp &factor) {
    hipLaunchKernelGGL(scale_strided_kernel, grid_for(count, 256), dim3(256), 0, st, data, stride, off, count, factor);
    return hipGetLastError();
}

hipError_t launch_diluted_aggregate_maps(hipStream_t st, const Fp *x, uint64_t stride, uint64_t off, uint64_t count, bool starts_column,
                                         const Fp &z, const Fp &alpha, Fp *mc, Fp *scratch) {
    static const ScanShape shape = shape_from_env();
    HipExec ex{st, shape};
    return (hipError_t)diluted_aggregate_maps(ex, x, stride, off, count, starts_column, z, alpha, mc, scratch);
}

hipError_t launch_affine_apply(hipStream_t st, const Fp *mc, uint64_t count, const Fp &start, Fp *out, uint64_t out_stride, uint64_t out_off) {
    hipLaunchKernelGGL(affine_apply_kernel, grid_for(count, 256), dim3(256), 0, st, mc, count, start, out, out_stride, out_off);
    return hipGetLastError();
}

}  // namespace ss
