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

// Device instruction stream (built by ss_eval_quotient from the caller's 2-word program):
// 4 words per entry  { w0 | pk << 16,  w1,  pw,  0 },  entry 0 is a no-op that primes the pipeline.
// (pk, pw) names the memory operand (kind, payload) that THIS entry starts loading: the operand
// of the next memory-consuming instruction, so that its L2/HBM round trip overlaps this
// instruction's arithmetic.  Measured (profiles/r01_quotient_vm_experiments.txt): the kernel is
// VALU-bound (295 k VALU instructions per 64-point wave-row), so one operand of lead is all
// that pays; a deeper LDS-DMA operand ring (2 to 8 operands in flight) gave nothing more.
static constexpr uint32_t VM_PK_NONE = 0xfu;
static constexpr uint32_t VM_OP_NOP = 0xffu;

__device__ __forceinline__ const Fp *vm_operand_ptr(const VmArgs &a, uint32_t kind, uint32_t w, uint64_t N,
                                                    uint64_t lanes, uint64_t lane, uint64_t i) {
    if (kind == SS_SRC_SLOT) return a.slots + (uint64_t)w * lanes + lane;
    if (kind == SS_SRC_CONST) return a.consts + w;
    if (kind == SS_SRC_TRACE) {
        const uint32_t col = w >> 24;
        const uint64_t row = (i + ((uint64_t)(w & 0xffffffu) << a.log_blowup)) & (N - 1);
        const Fp *cp = a.cols[0];
#pragma unroll
        for (int c = 1; c < MAX_COLS; ++c) if (col == (uint32_t)c) cp = a.cols[c];
        return cp + row;
    }
    const uint32_t off = a.table_desc[2 * w], ll = a.table_desc[2 * w + 1];
    return a.tables + off + (i & ((1ull << ll) - 1ull));
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
    const uint4 *code = reinterpret_cast<const uint4 *>(a.code);
    for (uint64_t it = 0; it < count; ++it) {
        const uint64_t i = i0 + it * stride;
        Fl acc0 = fl_zero(), acc1 = fl_zero(), acc2 = fl_zero(), acc3 = fl_zero();
        uint32_t bnd0 = 1, bnd1 = 1, bnd2 = 1, bnd3 = 1;
        Fp pre = fp_zero();                                // the operand in flight
        for (uint32_t pc = 0; pc < a.n_instr; ++pc) {
            const uint4 ins = code[pc];
            const uint32_t w0 = ins.x, w1 = ins.y, pw = ins.z;
            const uint32_t op = w0 & 0xffu, d = (w0 >> 8) & 0xfu, kind = (w0 >> 12) & 0xfu, pk = (w0 >> 16) & 0xfu;
            const bool mem = op <= SS_OP_MUL && kind >= SS_SRC_SLOT && kind <= SS_SRC_TABLE;
            const Fp cur = pre;
            // start the next operand before this instruction's arithmetic; after an ST only once
            // the store has been issued (the operand may be the slot just written)
            if (pk != VM_PK_NONE && op != SS_OP_ST) pre = qload(vm_operand_ptr(a, pk, pw, N, lanes, lane, i));
            Fl src = fl_zero();
            uint32_t sb = 1;                               // bound of src
            if (mem) {
                src = fl_from_fp(cur);                     // canonical or weakly reduced 256-bit image
            } else if (op <= SS_OP_MUL) {
                if (kind == SS_SRC_ACC) {
                    switch (w1 & 3u) {
                    case 0: src = acc0; sb = bnd0; break;
                    case 1: src = acc1; sb = bnd1; break;
                    case 2: src = acc2; sb = bnd2; break;
                    default: src = acc3; sb = bnd3; break;
                    }
                } else {
                    src = x;
                }
            }
            switch (d) {
            case 0: vm_exec(op, acc0, bnd0, src, sb, a, w1, lanes, lane, i); break;
            case 1: vm_exec(op, acc1, bnd1, src, sb, a, w1, lanes, lane, i); break;
            case 2: vm_exec(op, acc2, bnd2, src, sb, a, w1, lanes, lane, i); break;
            default: vm_exec(op, acc3, bnd3, src, sb, a, w1, lanes, lane, i); break;
            }
            if (pk != VM_PK_NONE && op == SS_OP_ST) pre = qload(vm_operand_ptr(a, pk, pw, N, lanes, lane, i));
        }
        x = fl_mul(x, wstep);
    }
}

// 2-word caller program -> the 4-word device stream described above (n_instr + 1 entries).
void quotient_build_device_code(const uint32_t *code, uint32_t n_instr, uint32_t *dev) {
    auto put = [&](uint32_t idx, uint32_t w0, uint32_t w1) {
        dev[4 * idx] = w0 | (VM_PK_NONE << 16); dev[4 * idx + 1] = w1; dev[4 * idx + 2] = 0; dev[4 * idx + 3] = 0;
    };
    put(0, VM_OP_NOP, 0);
    uint32_t last_node = 0, last_consumer = 0;     // entry indices; the priming no-op counts as both
    for (uint32_t pc = 0; pc < n_instr; ++pc) {
        const uint32_t w0 = code[2 * pc] & 0xffffu, w1 = code[2 * pc + 1], idx = pc + 1;
        const uint32_t op = w0 & 0xffu, kind = (w0 >> 12) & 0xfu;
        put(idx, w0, w1);
        if (op <= SS_OP_MUL && kind >= SS_SRC_SLOT && kind <= SS_SRC_TABLE) {
            // a slot operand may only be requested after the last ST in front of it; anything
            // else is requested by the previous memory consumer, across any STs in between
            const uint32_t issuer = kind == SS_SRC_SLOT ? last_node : last_consumer;
            dev[4 * issuer] = (dev[4 * issuer] & ~(0xfu << 16)) | (kind << 16);
            dev[4 * issuer + 2] = w1;
            last_node = last_consumer = idx;
        } else if (op == SS_OP_ST) {
            last_node = idx;
        }
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
    a.n_instr = n_instr + 1;                 // d_code is the (n_instr + 1)-entry device stream
    hipLaunchKernelGGL(quotient_vm_kernel, dim3((uint32_t)(lanes / 256)), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace ss
