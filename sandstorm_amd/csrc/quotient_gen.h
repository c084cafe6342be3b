// quotient_gen.h — what the generated constraint kernels (quotient_gen_<layout>.hip, written by tools/gen_quotient.py)
// share: the argument block, the operand macros and the point loop.  One lane = one point of the evaluation domain
// x_i = offset * w^i; everything that is uniform over the wave (column bases, table descriptors, constants) comes
// through kernel arguments and scalar loads, everything per point is a coalesced 32-byte load (lane i reads element
// (i + shift) & mask of a column).
#pragma once
#include <hip/hip_runtime.h>
#include "fp252.h"
#include "fl252.h"

namespace ss {

static constexpr int QG_THREADS = 256;
static constexpr int QG_MAX_COLS = 16;
static constexpr int QG_CONST_STRIDE = 24;       // dwords per constant: 9 R256 limbs at 0, 9 R280 limbs at 12

struct QGenArgs {
    const Fp *cols[QG_MAX_COLS];
    const Fp *tables;                            // all tables, concatenated
    const uint32_t *tdesc;                       // per table: first element, index mask (period - 1)
    const uint32_t *consts;                      // per constant QG_CONST_STRIDE dwords, limb form
    Fp *out;
    Fp offset, w, wstep;                         // x_i = offset * w^i; wstep = w^(lanes of the grid)
    uint32_t log_N, log_blowup;
};

struct QGenKernel {
    const char *layout;
    uint64_t code_hash;                          // FNV-1a of the program's code words
    uint32_t n_instr, n_consts, n_tables, ncols;
    hipError_t (*launch)(hipStream_t, const QGenArgs &, uint32_t blocks);
};

const QGenKernel &quotient_gen_starknet();       // quotient_gen_starknet.hip
const QGenKernel &quotient_gen_recursive();      // quotient_gen_recursive.hip

typedef uint32_t qg_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ Fl qg_load(const Fp *base, uint32_t index) {
    const qg_u32x4 *q = reinterpret_cast<const qg_u32x4 *>(base + index);
    const qg_u32x4 lo = q[0], hi = q[1];
    Fp r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return fl_from_fp(r);
}
__device__ __forceinline__ Fl qg_const(const uint32_t *__restrict__ c) {      // wave-uniform address: scalar loads
    Fl r;
#pragma unroll
    for (int k = 0; k < 9; ++k) r.l[k] = c[k];
    return r;
}
__device__ __forceinline__ void qg_store(Fp *p, const Fp &x) {
    qg_u32x4 *q = reinterpret_cast<qg_u32x4 *>(p);
    q[0] = qg_u32x4{x.v[0], x.v[1], x.v[2], x.v[3]};
    q[1] = qg_u32x4{x.v[4], x.v[5], x.v[6], x.v[7]};
}
#define qstore qg_store

// operands (the generator writes these with immediates)
#define QG_TRACE(col, off) qg_load(a.cols[col], (i32 + ((off) << lb)) & maskN)
#define QG_TABLE(t) qg_load(a.tables + tdesc[2 * (t)], i32 & tdesc[2 * (t) + 1])
#define QG_CONST(k) qg_const(consts + QG_CONST_STRIDE * (k))
#define QG_CONST_R280(k) qg_const(consts + QG_CONST_STRIDE * (k) + 12)

#define QG_PROLOGUE                                                                                   \
    const uint64_t N = 1ull << a.log_N;                                                               \
    const uint64_t lanes = (uint64_t)gridDim.x * blockDim.x;                                          \
    const uint64_t lane = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;                            \
    const uint32_t lb = a.log_blowup, maskN = (uint32_t)(N - 1);                                      \
    const uint32_t *consts = a.consts;                                                                \
    const uint32_t *tdesc = a.tdesc;                                                                  \
    Fl x = fl_from_fp(fp_mul(a.offset, fp_pow_u64(a.w, lane)));                                       \
    const Fl wstep = fl_from_fp(a.wstep);

// the constant pointers are made opaque once per point: the loads stay inside the loop as scalar loads that the
// scheduler places ahead of their use, instead of ~2000 hoisted SGPRs spilled to VGPR lanes
#define QG_POINT_LOOP_BEGIN                                                                           \
    for (uint64_t i = lane; i < N; i += lanes) {                                                      \
        const uint32_t i32 = (uint32_t)i;                                                             \
        asm volatile("" : "+s"(consts), "+s"(tdesc));

#define QG_POINT_LOOP_END                                                                             \
        x = fl_mul(x, wstep);                                                                         \
    }

}  // namespace ss
