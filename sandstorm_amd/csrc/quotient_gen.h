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

#ifdef QG_THREADS_PER_WG                       // a wrapper's own workgroup shape (tools/gen_quotient.py VARIANTS)
static constexpr int QG_THREADS = QG_THREADS_PER_WG;
#else
static constexpr int QG_THREADS = 256;
#endif
static constexpr int QG_MAX_COLS = 16;
static constexpr int QG_CONST_STRIDE = 48;       // dwords per constant, four limb forms of 9 (+ 3 pad): R256 limbs at 0 (add / sub / mov), R280
                                                 // limbs at 12 (fl_mul_r280), R280 limbs of c 2^24 at 24 and of -c 2^24 at 36 (the alpha powers
                                                 // that take a constraint's own wide sum g C 2^-24 back: tools/gen_quotient.py plan_wide_constraints)

struct QGenArgs {
    const Fp *cols[QG_MAX_COLS];
    const Fp *tables;                            // all tables, concatenated
    const Fp *tables_scaled;                     // the multiplier-only tables times 2^24, concatenated (QGenKernel::scaled; made per launch)
    const uint32_t *tdesc;                       // per table: first element, index mask (period - 1); then the same per scaled copy
    const uint32_t *consts;                      // per constant QG_CONST_STRIDE dwords, limb form
    Fp *out;
    Fp *sink;                                    // 32 bytes nobody reads: where lanes past the end of the points store
    Fp offset, w;                                // x of local point k = offset * w^k (offset carries w^row0)
    const Fp *wstep_ptr;                         // -> w^(lanes of this launch's grid), in device memory (read at the end of every point)
    uint64_t npoints;                            // points evaluated: the whole domain, or one row block of it
    uint32_t row0;                               // global row of local point 0 (tables are indexed by the global row)
    uint32_t trace_mask;                         // trace cells are read at (k + shift) & trace_mask: N - 1 on whole columns, ~0 on a
    uint32_t log_blowup;                         // row block that carries the rows behind it (ss_eval_quotient_rows)
};

// One compiled program = one or more kernels ("parts", tools/gen_quotient.py split_program) whose outputs sum to the
// program's: part 0 stores its sum, every further part adds its own into the output.  A part is sized to the registers and
// LDS of its workgroup shape, so that two workgroups per CU are resident wherever its scratch values allow.
static constexpr int QG_MAX_PARTS = 8;
struct QGenPart {
    uint32_t wgs_per_cu;                         // workgroups per CU its register / LDS budget allows: the grid is 256 CUs times this
    uint32_t threads;                            // lanes per workgroup (QG_THREADS of that kernel's translation unit)
    uint32_t n_instr;                            // program instructions in this part
    hipError_t (*launch)(hipStream_t, const QGenArgs &, uint32_t blocks);
};
struct QGenKernel {
    const char *layout;
    uint64_t code_hash;                          // FNV-1a of the (whole) program's code words
    uint32_t n_instr, n_consts, n_tables, ncols;
    uint32_t variant;                            // tools/gen_quotient.py VARIANTS; ss_eval_quotient takes 0 (SS_QG_VARIANT overrides)
    uint32_t n_parts;
    QGenPart parts[QG_MAX_PARTS];
    uint32_t n_scaled;                           // tables the kernels read from a 2^24-fold copy (multiplier-only tables): their numbers;
    const uint32_t *scaled;                      // copy j is addressed by descriptor n_tables + j
};

// the launch's copy of the multiplier-only tables times 2^24 (csrc/quotient.hip qg_scale_tables_kernel)
static constexpr int QG_MAX_SCALED = 48;
struct QgScaleArgs {
    const Fp *tables;
    Fp *out;
    uint32_t n, src[QG_MAX_SCALED], dst[QG_MAX_SCALED], len[QG_MAX_SCALED];      // per table: first element in `tables`, in `out`, elements
};
hipError_t launch_qg_scale_tables(hipStream_t st, const QgScaleArgs &a, uint64_t total_felts);

#define QG_VARIANT(entry) const QGenKernel &entry();
#include "quotient_gen_variants.inc"             // quotient_gen_starknet(), quotient_gen_recursive() [, the A/B variants]
#undef QG_VARIANT

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
// (a lane past the end of the points stores into a.sink instead: a select of the address, not a branch around the store -
// control flow here costs the register allocator the whole kernel, 6 KB of scratch per lane)
#define QG_OUT(v) qg_store(qg_live ? a.out + i : a.sink, fl_to_fp(v))
// parts after the first: out += v (canonical before and after; out[i] was written by the previous part's launch)
#define QG_OUT_ACC(v) qg_store(qg_live ? a.out + i : a.sink, fl_to_fp(fl_add(fl_weak_reduce(v), fl_from_fp(qg_load_raw(a.out, (uint32_t)i)))))
// QG_SYNC (between program instructions, where a part's configuration asks for it): the waves of a workgroup run the same
// straight-line code; kept within a few instructions of each other they share every instruction-cache line they fetch
// (a part is 100-900 KB of code against 64 KB of instruction cache per two CUs)
#ifdef QG_SYNC_WAVES
#define QG_SYNC __syncthreads();
#else
#define QG_SYNC
#endif

// constants live in LDS for the kernel's lifetime (36 dwords each: the four limb forms of QG_CONST_STRIDE): a wave-uniform LDS read
// is a broadcast, its latency is short and known to the scheduler - unlike ~25 KB of scalar loads that miss the 16 KB
// scalar cache in front of every multiplication
static constexpr int QG_CONST_LDS_STRIDE = 36;
// LDS pointers carry their address space (32-bit, ds_* instructions) also through the per-point laundering below
typedef uint32_t __attribute__((address_space(3))) *qg_lds_u32;
typedef qg_u32x4 __attribute__((address_space(3))) *qg_lds_u32x4;
__device__ __forceinline__ Fl qg_const_lds(const qg_lds_u32 c) {
    Fl r;
#pragma unroll
    for (int k = 0; k < 9; ++k) r.l[k] = c[k];
    return r;
}

// fused dot products (fl252.h FlWide): used for sum_k alpha^k C_k, where they replace a reduction and an addition per
// constraint by 1 / 16 of one
typedef FlWide QgWide;
#define qg_dot_zero fl_wide_zero
#define qg_dot_mad fl_wide_mad
#define qg_dot_reduce fl_wide_reduce
// + L 2^256 into a wide sum that the ten-step reduction divides by 2^280: 2^256 = 2^4 2^(9 * 28), so limb j goes to column 9 + j
// shifted by four bits (a lazy limb < 2^32: far below a column's 2^64)
__device__ __forceinline__ void qg_wide_tail(QgWide &w, const Fl &l) {
#pragma unroll
    for (int j = 0; j < 9; ++j) w.c[9 + j] += (u64)l.l[j] << 4;
}

// operands (the generator writes these with immediates).  `idx` is the point's local index: this point's, or the next
// point's for the loads issued across the loop edge.
#define QG_TRACE_RAW(col, off, idx) qg_load_raw(a.cols[col], ((idx) + ((off) << lb)) & maskN)
#define QG_TABLE_RAW(t, idx) qg_load_raw(a.tables + tdesc[2 * (t)], ((idx) + row0) & tdesc[2 * (t) + 1])
#define QG_TABLE_SCALED_RAW(t, idx) qg_load_raw(a.tables_scaled + tdesc[2 * (t)], ((idx) + row0) & tdesc[2 * (t) + 1])
#define QG_CONST(k) qg_const_lds(lds_consts + QG_CONST_LDS_STRIDE * (k))
#define QG_CONST_R280(k) qg_const_lds(lds_consts + QG_CONST_LDS_STRIDE * (k) + 9)
#define QG_CONST_R280_UP(k) qg_const_lds(lds_consts + QG_CONST_LDS_STRIDE * (k) + 18)
#define QG_CONST_R280_UPN(k) qg_const_lds(lds_consts + QG_CONST_LDS_STRIDE * (k) + 27)
// keep a load where the generator put it: ALU instructions and LDS writes may still be scheduled across, vector-memory
// instructions and LDS reads may not (an LDS read of a constant depends on nothing: unpinned, hundreds of them float
// to the top of the point and sit in registers until used)
#define QG_PIN_LOADS __builtin_amdgcn_sched_barrier(0x1 | 0x2 | 0x4 | 0x8 | 0x200);
// between two program instructions.  QG_FENCE_EVERY_INSTRUCTION: nothing is scheduled across, so the machine scheduler works
// on one program instruction at a time instead of interleaving several multiplications (each 19 64-bit columns) for ILP the
// 81 independent multiply-adds of ONE product already provide
#ifdef QG_FENCE_EVERY_INSTRUCTION
#define QG_FENCE __builtin_amdgcn_sched_barrier(0);
#else
#define QG_FENCE
#endif

// scratch slots of the program: per point, in LDS as 32-byte images in two 16-byte planes [slot][lane] (a lane's 16 bytes
// next to its neighbour's: conflict-free b128 accesses), so that the registers hold only what is being computed on
#ifdef QG_SLOTS_IN_REGISTERS
// variant: the slots are plain lazy values the compiler keeps in (or spills from) registers; no LDS slot area
#define QG_SLOT_STORE(k, v) qg_s##k = (v)
#define QG_SLOT(k) qg_s##k
#define QG_DECLARE_SLOTS Fl qg_s0, qg_s1, qg_s2, qg_s3, qg_s4, qg_s5, qg_s6, qg_s7, qg_s8, qg_s9, qg_s10, qg_s11, qg_s12, qg_s13, qg_s14, \
    qg_s15, qg_s16, qg_s17, qg_s18, qg_s19, qg_s20, qg_s21, qg_s22, qg_s23;
#else
#define QG_SLOT_STORE(k, v) qg_slot_store(lds_slots, (k), fl_pack(v))
#define QG_SLOT(k) fl_from_fp(qg_slot_load(lds_slots, (k)))
#define QG_DECLARE_SLOTS
#endif
__device__ __forceinline__ void qg_slot_store(qg_lds_u32x4 slots, int k, const Fp &x) {
    slots[(2 * k) * QG_THREADS + threadIdx.x] = qg_u32x4{x.v[0], x.v[1], x.v[2], x.v[3]};
    slots[(2 * k + 1) * QG_THREADS + threadIdx.x] = qg_u32x4{x.v[4], x.v[5], x.v[6], x.v[7]};
}
__device__ __forceinline__ Fp qg_slot_load(const qg_lds_u32x4 slots, int k) {
    const qg_u32x4 lo = slots[(2 * k) * QG_THREADS + threadIdx.x], hi = slots[(2 * k + 1) * QG_THREADS + threadIdx.x];
    Fp r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
static inline size_t qg_lds_bytes(int nconsts, int nslots) {
    return (size_t)nslots * 2 * QG_THREADS * 16 + (size_t)nconsts * QG_CONST_LDS_STRIDE * 4;
}

#define QG_PROLOGUE(NCONSTS, NSLOTS)                                                                  \
    extern __shared__ __attribute__((aligned(16))) unsigned char qg_smem[];                           \
    qg_lds_u32x4 lds_slots = (qg_lds_u32x4)qg_smem;                                                   \
    qg_lds_u32 lds_consts = (qg_lds_u32)(qg_smem + (size_t)(NSLOTS) * 2 * QG_THREADS * 16);           \
    for (uint32_t k = threadIdx.x; k < (uint32_t)(NCONSTS) * QG_CONST_LDS_STRIDE; k += blockDim.x)    \
        lds_consts[k] = a.consts[(k / QG_CONST_LDS_STRIDE) * QG_CONST_STRIDE + ((k % QG_CONST_LDS_STRIDE) / 9) * 12 + (k % QG_CONST_LDS_STRIDE) % 9]; \
    __syncthreads();                                                                                  \
    const uint64_t N = a.npoints;                                                                     \
    const uint64_t lanes = (uint64_t)gridDim.x * blockDim.x;                                          \
    const uint64_t lane = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;                            \
    const uint32_t lb = a.log_blowup, maskN = a.trace_mask, row0 = a.row0;                            \
    const uint32_t *tdesc = a.tdesc;                                                                  \
    Fl x = fl_from_fp(fp_mul(a.offset, fp_pow_u64(a.w, lane)));                                       \
    QG_DECLARE_SLOTS

// Every lane of the grid runs the same number of iterations (the parts may hold workgroup barriers - QG_SYNC - inside a point):
// a lane past the end of the points works on the last point again and does not store.
#define QG_POINT_LOOP_BEGIN                                                                           \
    const uint64_t qg_iters = (N + lanes - 1) / lanes;                                                \
    for (uint64_t qg_it = 0; qg_it < qg_iters; ++qg_it) {                                             \
        const uint64_t qg_raw = lane + qg_it * lanes;                                                 \
        const bool qg_live = qg_raw < N;                                                              \
        const uint64_t i = qg_live ? qg_raw : N - 1;                                                  \
        i32 = (uint32_t)i;                                                                            \
        const uint32_t inext = (uint32_t)(qg_raw + lanes < N ? qg_raw + lanes : N - 1);               \
        /* the constants do not change, but their loads must not be hoisted out of the loop (thousands of registers) */ \
        asm volatile("" : "+v"(lds_consts), "+v"(lds_slots));

// The step to the lane's next point is read from memory at the end of every point (two 16-byte loads + 16 vector instructions):
// held in registers across the point it cost nine of the 256 that the straight-line code fights over.
#define QG_POINT_LOOP_END                                                                             \
        {                                                                                             \
            const Fp *qg_ws = a.wstep_ptr;                                                            \
            asm volatile("" : "+v"(qg_ws));                                                           \
            x = fl_mul(x, fl_from_fp(qg_load_raw(qg_ws, 0)));                                         \
        }                                                                                             \
    }

}  // namespace ss
