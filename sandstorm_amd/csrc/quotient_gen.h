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
    Fp offset, w, wstep;                         // x of local point k = offset * w^k (offset carries w^row0); wstep = w^(lanes of the grid)
    uint64_t npoints;                            // points evaluated: the whole domain, or one row block of it
    uint32_t row0;                               // global row of local point 0 (tables are indexed by the global row)
    uint32_t trace_mask;                         // trace cells are read at (k + shift) & trace_mask: N - 1 on whole columns, ~0 on a
    uint32_t log_blowup;                         // row block that carries the rows behind it (ss_eval_quotient_rows)
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

// a 32-byte element as it lies in memory (re-limbed at its use, so that the load can be issued far ahead at 8 registers)
__device__ __forceinline__ Fp qg_load_raw(const Fp *base, uint32_t index) {
    const qg_u32x4 *q = reinterpret_cast<const qg_u32x4 *>(base + index);
    const qg_u32x4 lo = q[0], hi = q[1];
    Fp r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
__device__ __forceinline__ void qg_store(Fp *p, const Fp &x) {
    qg_u32x4 *q = reinterpret_cast<qg_u32x4 *>(p);
    q[0] = qg_u32x4{x.v[0], x.v[1], x.v[2], x.v[3]};
    q[1] = qg_u32x4{x.v[4], x.v[5], x.v[6], x.v[7]};
}
#define qstore qg_store

// constants live in LDS for the kernel's lifetime (18 dwords each: 9 R256 limbs, 9 R280 limbs): a wave-uniform LDS read
// is a broadcast, its latency is short and known to the scheduler - unlike ~25 KB of scalar loads that miss the 16 KB
// scalar cache in front of every multiplication
static constexpr int QG_CONST_LDS_STRIDE = 18;
__device__ __forceinline__ Fl qg_const_lds(const uint32_t *c) {
    Fl r;
#pragma unroll
    for (int k = 0; k < 9; ++k) r.l[k] = c[k];
    return r;
}

// ---- fused dot products: sum_k a_k * t_k with ONE Montgomery reduction.  The 81 partial products of every term go
// straight into 19 64-bit columns (normalised 28-bit limbs on both sides: a column grows by < 9 * 2^56 per term, so 16
// terms stay below 2^64 with room for the reduction's own terms); qg_dot_reduce is the ten-step R280 reduction of
// fl_mul_r280.  Used for sum_k alpha^k C_k, where it replaces a reduction and an addition per constraint by 1 / 16 of one.
struct QgWide { u64 c[19]; };
__device__ __forceinline__ void qg_dot_zero(QgWide &w) {
#pragma unroll
    for (int k = 0; k < 19; ++k) w.c[k] = 0;
}
__device__ __forceinline__ void qg_dot_mad(QgWide &w, const Fl &a, const Fl &t) {
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) w.c[i + j] += (u64)a.l[i] * t.l[j];
}
__device__ __forceinline__ Fl qg_dot_reduce(QgWide &w) {
    const u32 k24 = fl_opaque(1u << 24), k27 = fl_opaque(1u << 27), k1 = fl_opaque(1u);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const u32 m = (0u - (u32)w.c[i]) & FL_MASK;
        w.c[i + 1] += ((u64)m * k1 + w.c[i]) >> 28;
        w.c[i + 6] += (u64)m * k24;
        w.c[i + 7] += (u64)m * k1;
        w.c[i + 8] += (u64)m * k27;
    }
    Fl r;
    u64 carry = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const u64 v = w.c[10 + j] + carry;
        r.l[j] = (u32)v & FL_MASK;
        carry = v >> 28;
    }
    r.l[8] = (u32)(w.c[18] + carry);
    return r;
}

// operands (the generator writes these with immediates).  `idx` is the point's local index: this point's, or the next
// point's for the loads issued across the loop edge.
#define QG_TRACE_RAW(col, off, idx) qg_load_raw(a.cols[col], ((idx) + ((off) << lb)) & maskN)
#define QG_TABLE_RAW(t, idx) qg_load_raw(a.tables + tdesc[2 * (t)], ((idx) + row0) & tdesc[2 * (t) + 1])
#define QG_CONST(k) qg_const_lds(lds_consts + QG_CONST_LDS_STRIDE * (k))
#define QG_CONST_R280(k) qg_const_lds(lds_consts + QG_CONST_LDS_STRIDE * (k) + 9)
// keep a load where the generator put it: everything but vector-memory instructions may still be scheduled across
#define QG_PIN_LOADS __builtin_amdgcn_sched_barrier(0x1 | 0x2 | 0x4 | 0x8 | 0x80 | 0x100 | 0x200);

#define QG_PROLOGUE(NCONSTS)                                                                          \
    __shared__ uint32_t lds_consts[(NCONSTS) * QG_CONST_LDS_STRIDE];                                  \
    for (uint32_t k = threadIdx.x; k < (uint32_t)(NCONSTS) * QG_CONST_LDS_STRIDE; k += blockDim.x)    \
        lds_consts[k] = a.consts[(k / QG_CONST_LDS_STRIDE) * QG_CONST_STRIDE + (k % QG_CONST_LDS_STRIDE < 9 ? k % QG_CONST_LDS_STRIDE : k % QG_CONST_LDS_STRIDE + 3)]; \
    __syncthreads();                                                                                  \
    const uint64_t N = a.npoints;                                                                     \
    const uint64_t lanes = (uint64_t)gridDim.x * blockDim.x;                                          \
    const uint64_t lane = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;                            \
    const uint32_t lb = a.log_blowup, maskN = a.trace_mask, row0 = a.row0;                            \
    const uint32_t *tdesc = a.tdesc;                                                                  \
    Fl x = fl_from_fp(fp_mul(a.offset, fp_pow_u64(a.w, lane)));                                       \
    const Fl wstep = fl_from_fp(a.wstep);

// lanes past the end (a grid larger than the block of points) still run the loads with a wrapped index, never the store
#define QG_POINT_LOOP_BEGIN                                                                           \
    for (uint64_t i = lane; i < N; i += lanes) {                                                      \
        i32 = (uint32_t)i;                                                                            \
        const uint32_t inext = (uint32_t)(i + lanes < N ? i + lanes : i);

#define QG_POINT_LOOP_END                                                                             \
        x = fl_mul(x, wstep);                                                                         \
    }

}  // namespace ss
