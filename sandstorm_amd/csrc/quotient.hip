// quotient.hip — constraint-program evaluation over the LDE domain on gfx950.
//
// Replaces ministark's AirConfig::eval_constraint (un-vendored default body) applied
// to the composition constraint that layouts/src/{recursive,starknet}/air.rs build
// (recursive air.rs:61-1200; row Q1 of SURVEY.md §8a).  The host lowers the `Expr`
// DAG once per (layout, trace length) into the 4-accumulator program described in
// include/sandstorm_hip.h; this kernel interprets it.
//
// One lane = one LDE point; the program counter, opcodes and operand selectors are
// wave-uniform (scalar loads, scalar branches — no divergence), the four
// accumulators are VGPR-resident, constants arrive through the scalar cache, trace
// cells are coalesced (lane i reads element i + shift of a column) and scratch slots
// are [slot][lane] so a slot access is one contiguous 2 KiB run per wave.
#include <hip/hip_runtime.h>
#include "../../include/sandstorm_hip.h"
#include "fp252.h"
#include "fl252.h"
#include "kernels.h"

namespace ss {

__device__ __forceinline__ Fp qload(const Fp *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    Fp r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__device__ __forceinline__ void qstore(Fp *p, const Fp &x) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    q[1] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
}

struct VmArgs {
    const Fp *cols[MAX_COLS];
    const uint32_t *code;
    const Fp *consts;
    const Fp *tables;
    const uint32_t *table_desc;
    Fp *slots;               // [n_slots][total_lanes]
    Fp *out;
    Fp offset, w, wstep;     // x_i = offset * w^i; wstep = w^(total lanes)
    uint32_t n_instr, log_N, log_blowup, xcd_split;
};

// Accumulators live in the lazy 9 x 28-bit form (fl252.h).  Because the program is
// wave-uniform, a per-accumulator bound b (value < 2 b p, limbs < b 2^28) is tracked in
// scalar registers: ADD / SUB are nine carry-less 32-bit adds that bump the bound, and a
// weak reduction is issued (a uniform branch) only when an operation's precondition
// needs it:  fl_mul   a-side any b <= 8, b-side b == 1;   fl_sub_c<2,1> subtrahend b == 1;
// ST / table-free operands from memory are always b == 1.
static constexpr uint32_t VM_MAX_BOUND = 8;       // 8 * 2p * 2p / 2^256 + p < 2p: products stay < 2^252

// One instruction on a NAMED accumulator (v, vb): the destination index is wave-uniform,
// so the caller dispatches with a scalar switch to one of four inlined copies of this body
// instead of selecting registers with v_cndmask chains.
__device__ __forceinline__ void vm_exec(uint32_t op, Fl &v, uint32_t &vb, Fl src, uint32_t sb, const VmArgs &a,
                                        uint32_t w1, uint64_t lanes, uint64_t lane, uint64_t i) {
    switch (op) {
    case SS_OP_MOV: v = src; vb = sb; break;
    case SS_OP_ADD:
        if (vb + sb > VM_MAX_BOUND) { v = fl_weak_reduce(v); vb = 1; }
        if (vb + sb > VM_MAX_BOUND) { src = fl_weak_reduce(src); sb = 1; }
        v = fl_add(v, src); vb += sb;
        break;
    case SS_OP_SUB:
        if (sb > 1) { src = fl_weak_reduce(src); sb = 1; }
        if (vb + 1 > VM_MAX_BOUND) { v = fl_weak_reduce(v); vb = 1; }
        v = fl_sub_c<2, 1>(v, src); vb += 1;
        break;
    case SS_OP_RSUB:
        if (vb > 1) { v = fl_weak_reduce(v); vb = 1; }
        if (sb + 1 > VM_MAX_BOUND) { src = fl_weak_reduce(src); sb = 1; }
        v = fl_sub_c<2, 1>(src, v); vb = sb + 1;
        break;
    case SS_OP_MUL:
        if (sb > 1) { src = fl_weak_reduce(src); sb = 1; }
        v = fl_mul(v, src); vb = 1;                // vb <= VM_MAX_BOUND by construction
        break;
    case SS_OP_INV: v = fn_inv(fl_weak_reduce(v)); vb = 1; break;
    case SS_OP_ST:
        if (vb > 1) { v = fl_weak_reduce(v); vb = 1; }      // slots hold weakly reduced images
        qstore(a.slots + (uint64_t)w1 * lanes + lane, fl_pack(v));
        break;
    case SS_OP_OUT: qstore(a.out + i, fl_to_fp(v)); break;
    default: break;
    }
}

__global__ __launch_bounds__(256, 4) void quotient_vm_kernel(VmArgs a) {
    const uint64_t N = 1ull << a.log_N;
    const uint64_t lanes = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t lane = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;       // slot-file index
    // XCD-aware point mapping.  Workgroup b runs on XCD b % 8 (observed dispatch order; used
    // for speed only): every XCD sweeps its own contiguous 1/8 of the LDE domain, so the rows
    // that different mask offsets re-read stay inside that XCD's 4 MiB L2 instead of being
    // spread over all eight.  a.xcd_split == 0 falls back to the plain grid-stride order.
    uint64_t i0, stride, count;
    if (a.xcd_split) {
        const uint64_t per_xcd = N >> 3, lanes_xcd = lanes >> 3;
        i0 = (blockIdx.x & 7u) * per_xcd + (uint64_t)(blockIdx.x >> 3) * blockDim.x + threadIdx.x;
        stride = lanes_xcd;
        count = per_xcd / lanes_xcd;
    } else {
        i0 = lane; stride = lanes; count = (N + lanes - 1 - lane) / lanes;
    }
    Fl x = fl_from_fp(fp_mul(a.offset, fp_pow_u64(a.w, i0)));
    const Fl wstep = fl_from_fp(a.wstep);                                          // w^stride
    for (uint64_t it = 0; it < count; ++it) {
        const uint64_t i = i0 + it * stride;
        Fl acc0 = fl_zero(), acc1 = fl_zero(), acc2 = fl_zero(), acc3 = fl_zero();
        uint32_t bnd0 = 1, bnd1 = 1, bnd2 = 1, bnd3 = 1;
        for (uint32_t pc = 0; pc < a.n_instr; ++pc) {
            const uint32_t w0 = a.code[2 * pc], w1 = a.code[2 * pc + 1];
            const uint32_t op = w0 & 0xffu, d = (w0 >> 8) & 0xfu, kind = (w0 >> 12) & 0xfu;
            Fl src = fl_zero();
            uint32_t sb = 1;                               // bound of src
            if (op <= SS_OP_MUL) {
                if (kind == SS_SRC_ACC) {
                    switch (w1 & 3u) {
                    case 0: src = acc0; sb = bnd0; break;
                    case 1: src = acc1; sb = bnd1; break;
                    case 2: src = acc2; sb = bnd2; break;
                    default: src = acc3; sb = bnd3; break;
                    }
                } else if (kind == SS_SRC_X) {
                    src = x;
                } else {
                    const Fp *ptr;
                    if (kind == SS_SRC_SLOT) {
                        ptr = a.slots + (uint64_t)w1 * lanes + lane;
                    } else if (kind == SS_SRC_CONST) {
                        ptr = a.consts + w1;
                    } else if (kind == SS_SRC_TRACE) {
                        const uint32_t col = w1 >> 24;
                        const uint64_t row = (i + ((uint64_t)(w1 & 0xffffffu) << a.log_blowup)) & (N - 1);
                        const Fp *cp = a.cols[0];
#pragma unroll
                        for (int c = 1; c < MAX_COLS; ++c) if (col == (uint32_t)c) cp = a.cols[c];
                        ptr = cp + row;
                    } else {
                        const uint32_t off = a.table_desc[2 * w1], ll = a.table_desc[2 * w1 + 1];
                        ptr = a.tables + off + (i & ((1ull << ll) - 1ull));
                    }
                    src = fl_from_fp(qload(ptr));       // canonical or weakly reduced 256-bit image
                }
            }
            switch (d) {
            case 0: vm_exec(op, acc0, bnd0, src, sb, a, w1, lanes, lane, i); break;
            case 1: vm_exec(op, acc1, bnd1, src, sb, a, w1, lanes, lane, i); break;
            case 2: vm_exec(op, acc2, bnd2, src, sb, a, w1, lanes, lane, i); break;
            default: vm_exec(op, acc3, bnd3, src, sb, a, w1, lanes, lane, i); break;
            }
        }
        x = fl_mul(x, wstep);
    }
}

hipError_t launch_quotient_vm(hipStream_t st, const void *const *cols, uint32_t ncols, const uint32_t *d_code,
                              uint32_t n_instr, const Fp *d_consts, const Fp *d_tables, const uint32_t *d_table_desc,
                              Fp *d_slots, uint64_t lanes, const Fp &offset, const Fp &w, const Fp &wstep,
                              uint32_t log_N, uint32_t log_blowup, uint32_t xcd_split, Fp *out) {
    VmArgs a;
    for (int c = 0; c < MAX_COLS; ++c) a.cols[c] = c < (int)ncols ? (const Fp *)cols[c] : nullptr;
    a.code = d_code; a.consts = d_consts; a.tables = d_tables; a.table_desc = d_table_desc; a.slots = d_slots;
    a.out = out; a.offset = offset; a.w = w; a.wstep = wstep; a.n_instr = n_instr; a.log_N = log_N;
    a.log_blowup = log_blowup; a.xcd_split = xcd_split;
    hipLaunchKernelGGL(quotient_vm_kernel, dim3((uint32_t)(lanes / 256)), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace ss
