// quotient.hip — constraint-program evaluation over the LDE domain on gfx950.
//
// Replaces ministark's AirConfig::eval_constraint (un-vendored default body) applied
// to the composition constraint that layouts/src/{recursive,starknet}/air.rs build
// (recursive air.rs:61-1200; row Q1 of SURVEY.md §8a).  The host lowers the `Expr`
// DAG once per (layout, trace length) into the 4-accumulator program described in
// include/sandstorm_hip.h; ss_eval_quotient resolves it (operand addresses, lazy-form
// bounds, R280 constants: "device program" below) and this kernel interprets that.
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
#include "quotient_gen.h"

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
    const uint32_t *code;    // device program, 8 words per entry (see below)
    Fp *slots;               // [n_slots][total_lanes]
    Fp *out;
    Fp offset, w, wstep;     // x_i = offset * w^i; wstep = w^(total lanes)
    uint32_t n_entries, xcd_split;
    uint64_t npoints;        // points evaluated (the whole domain or a row block of it)
};

// ---- device program -------------------------------------------------------------------------
// The interpreter's non-multiplying instructions are bound by SCALAR issue, not by the vector
// ALU (measured, profiles/r01_quotient_vm_experiments.txt: ~90 SALU per instruction = 360 cycles
// per SIMD against 68 VALU = 272): operand-kind decoding, the 12-way column select, table
// descriptor loads and the lazy-bound bookkeeping all ran on the one scalar unit a CU has.  The
// program is static, so ss_eval_quotient does that work once on the host and ships a resolved
// stream, 8 words per entry:
//   w0  control: op[0:3] d[4:5] src class[6:7] (0 memory, 1 accumulator, 2 x) src acc[8:9]
//       RV[10] weakly reduce the destination first   RS[11] weakly reduce the source first
//       P[12]  start loading the NEXT memory operand  PL[13] its index is the lane (slot file), not the point
//   w1  slot index (ST)
//   w2,w3 / w4 / w5   next operand: base pointer / index offset / index mask:
//       address = base + 32 * ((index + offset) & mask)
//       trace cell: base = column, offset = row_offset << log_blowup, mask = N - 1
//       table: base = table start, mask = period - 1     constant: mask = 0     slot: base = slot row, index = lane
// Entry 0 is a no-op that primes the pipeline.  The lazy-form bounds (value < 2 b p, limbs < b 2^28,
// b <= VM_MAX_BOUND) are tracked by the host with the rules below, which decide RV / RS:
//   ADD  needs b_v + b_s <= 8 (reduce v, then s, while it does not hold); b_v += b_s
//   SUB  subtrahend b_s == 1; b_v + 1 <= 8; b_v += 1        RSUB  b_v == 1; b_s + 1 <= 8; b_v = b_s + 1
//   MUL  b_s == 1 (the accumulator side takes any b <= 8); b_v = 1.  A MUL by a constant becomes MULR: the
//        host keeps a second copy of the constants in R280 form and points the operand there
//   INV  reduced input; ST stores a reduced image (and keeps it); memory operands and x are b = 1
static constexpr uint32_t VM_MAX_BOUND = 8;       // 8 * 2p * 2p / 2^256 + p < 2p: products stay < 2^252
static constexpr uint32_t VM_OP_NOP = 8;
static constexpr uint32_t VM_OP_MULR = 9;         // MUL by a constant kept in R280 form (fl252.h): the cheaper fl_mul_r280
static constexpr uint32_t VM_F_RV = 1u << 10, VM_F_RS = 1u << 11, VM_F_P = 1u << 12, VM_F_PL = 1u << 13;

// One instruction on a NAMED accumulator: the destination index is wave-uniform, so the caller
// dispatches with a scalar switch to one of four inlined copies of this body.
__device__ __forceinline__ void vm_exec(uint32_t w0, Fl &v, const Fl &src, const VmArgs &a, uint32_t w1,
                                        uint64_t lanes, uint64_t lane, uint64_t i) {
    if (w0 & VM_F_RV) v = fl_weak_reduce(v);
    switch (w0 & 0xfu) {
    case SS_OP_MOV: v = src; break;
    case SS_OP_ADD: v = fl_add(v, src); break;
    case SS_OP_SUB: v = fl_sub_c<2, 1>(v, src); break;
    case SS_OP_RSUB: v = fl_sub_c<2, 1>(src, v); break;
    case SS_OP_MUL: v = fl_mul(v, src); break;
    case VM_OP_MULR: v = fl_mul_r280(v, src); break;
    case SS_OP_INV: v = fn_inv(v); break;
    case SS_OP_ST: qstore(a.slots + (uint64_t)w1 * lanes + lane, fl_pack(v)); break;     // weakly reduced image
    case SS_OP_OUT: qstore(a.out + i, fl_to_fp(v)); break;
    default: break;
    }
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((address_space(4))) vm_const_u32x4;      // constant address space
typedef u32x4 __attribute__((address_space(1))) vm_global_u32x4;     // global address space
__device__ __forceinline__ Fp vm_load(const vm_global_u32x4 *p) {
    const u32x4 lo = p[0], hi = p[1];
    Fp r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}

__global__ __launch_bounds__(256, 4) void quotient_vm_kernel(VmArgs a) {
    const uint64_t N = a.npoints;
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
    // the program is read-only for the kernel's lifetime: constant address space => scalar loads
    // (s_load_dwordx8 through the scalar cache) instead of a vector load + v_readfirstlane per word
    const vm_const_u32x4 *code = (const vm_const_u32x4 *)(uintptr_t)a.code;
    for (uint64_t it = 0; it < count; ++it) {
        const uint64_t i = i0 + it * stride;
        const uint32_t i32 = (uint32_t)i, lane32 = (uint32_t)lane;
        Fl acc0 = fl_zero(), acc1 = fl_zero(), acc2 = fl_zero(), acc3 = fl_zero();
        Fp pre = fp_zero();                                // the operand in flight
        for (uint32_t pc = 0; pc < a.n_entries; ++pc) {
            const u32x4 c0 = code[2 * pc], c1 = code[2 * pc + 1];
            const uint32_t w0 = c0.x, w1 = c0.y;
            const uint32_t op = w0 & 0xfu;
            // start the next operand before this instruction's arithmetic; after an ST only once
            // the store has been issued (the operand may be the slot just written)
            const uint64_t nbase = ((uint64_t)c0.w << 32) | c0.z;
            const uint32_t nidx = (((w0 & VM_F_PL) ? lane32 : i32) + c1.x) & c1.y;
            const vm_global_u32x4 *nptr = (const vm_global_u32x4 *)(uintptr_t)(nbase + ((uint64_t)nidx << 5));   // global, not flat
            Fl src;
            switch ((w0 >> 6) & 3u) {
            case 0:                                        // canonical or weakly reduced 256-bit image:
                src = fl_from_fp(pre);                     // re-limbed BEFORE the register is reloaded
                break;
            case 1:
                switch ((w0 >> 8) & 3u) {
                case 0: src = acc0; break;
                case 1: src = acc1; break;
                case 2: src = acc2; break;
                default: src = acc3; break;
                }
                break;
            default: src = x; break;
            }
            if ((w0 & VM_F_P) && op != SS_OP_ST) pre = vm_load(nptr);
            if (w0 & VM_F_RS) src = fl_weak_reduce(src);
            switch ((w0 >> 4) & 3u) {
            case 0: vm_exec(w0, acc0, src, a, w1, lanes, lane, i); break;
            case 1: vm_exec(w0, acc1, src, a, w1, lanes, lane, i); break;
            case 2: vm_exec(w0, acc2, src, a, w1, lanes, lane, i); break;
            default: vm_exec(w0, acc3, src, a, w1, lanes, lane, i); break;
            }
            if ((w0 & VM_F_P) && op == SS_OP_ST) pre = vm_load(nptr);
        }
        x = fl_mul(x, wstep);
    }
}

// Caller's 2-word program -> resolved device program (n_instr + 1 entries of 8 words, see above).
void quotient_build_device_code(const uint32_t *code, uint32_t n_instr, const VmResolve &r, uint32_t *dev) {
    auto is_mem = [](uint32_t op, uint32_t kind) { return op <= SS_OP_MUL && kind >= SS_SRC_SLOT && kind <= SS_SRC_TABLE; };
    for (size_t k = 0; k < ((size_t)n_instr + 1) * 8; ++k) dev[k] = 0;
    dev[0] = VM_OP_NOP;
    uint32_t bound[4] = {1, 1, 1, 1};
    uint32_t last_node = 0, last_consumer = 0;     // entry indices; the priming no-op counts as both
    for (uint32_t pc = 0; pc < n_instr; ++pc) {
        const uint32_t c0 = code[2 * pc], w1 = code[2 * pc + 1], idx = pc + 1;
        const uint32_t op = c0 & 0xffu, d = (c0 >> 8) & 0x3u, kind = (c0 >> 12) & 0xfu;
        const bool mem = is_mem(op, kind);
        uint32_t w0 = op | (d << 4);
        uint32_t sb = 1;
        if (op <= SS_OP_MUL) {
            if (mem) w0 |= 0u << 6;
            else if (kind == SS_SRC_ACC) { w0 |= (1u << 6) | ((w1 & 3u) << 8); sb = bound[w1 & 3u]; }
            else w0 |= 2u << 6;                                                   // x
        }
        uint32_t &vb = bound[d];
        switch (op) {
        case SS_OP_MOV: vb = sb; break;
        case SS_OP_ADD:
            if (vb + sb > VM_MAX_BOUND) { w0 |= VM_F_RV; vb = 1; }
            if (vb + sb > VM_MAX_BOUND) { w0 |= VM_F_RS; sb = 1; }
            vb += sb;
            break;
        case SS_OP_SUB:
            if (sb > 1) { w0 |= VM_F_RS; sb = 1; }
            if (vb + 1 > VM_MAX_BOUND) { w0 |= VM_F_RV; vb = 1; }
            vb += 1;
            break;
        case SS_OP_RSUB:
            if (vb > 1) { w0 |= VM_F_RV; vb = 1; }
            if (sb + 1 > VM_MAX_BOUND) { w0 |= VM_F_RS; sb = 1; }
            vb = sb + 1;
            break;
        case SS_OP_MUL:
            if (sb > 1) { w0 |= VM_F_RS; sb = 1; }
            vb = 1;
            break;
        case SS_OP_INV: w0 |= VM_F_RV; vb = 1; break;
        case SS_OP_ST: if (vb > 1) { w0 |= VM_F_RV; vb = 1; } break;
        default: break;                                                           // OUT reduces fully itself
        }
        if (op == SS_OP_MUL && kind == SS_SRC_CONST) w0 = (w0 & ~0xfu) | VM_OP_MULR;
        dev[8 * idx] = w0;
        dev[8 * idx + 1] = w1;
        if (mem) {
            // a slot operand may only be requested after the last ST in front of it; anything
            // else is requested by the previous memory consumer, across any STs in between
            const uint32_t issuer = kind == SS_SRC_SLOT ? last_node : last_consumer;
            uint64_t base = 0;
            uint32_t off = 0, mask = 0, fl = VM_F_P;
            if (kind == SS_SRC_TRACE) {
                base = (uint64_t)r.cols[w1 >> 24]; off = (w1 & 0xffffffu) << r.log_blowup; mask = r.trace_mask;
            } else if (kind == SS_SRC_TABLE) {           // tables are indexed by the GLOBAL row
                base = (uint64_t)r.tables + 32ull * r.table_desc[2 * w1]; off = r.row0; mask = (uint32_t)((1ull << r.table_desc[2 * w1 + 1]) - 1ull);
            } else if (kind == SS_SRC_CONST) {
                base = (uint64_t)(op == SS_OP_MUL ? r.consts_r280 : r.consts) + 32ull * w1;
            } else {
                base = (uint64_t)r.slots + 32ull * w1 * r.lanes; mask = 0xffffffffu; fl |= VM_F_PL;
            }
            uint32_t *e = dev + 8 * (size_t)issuer;
            e[0] |= fl; e[2] = (uint32_t)base; e[3] = (uint32_t)(base >> 32); e[4] = off; e[5] = mask;
            last_node = last_consumer = idx;
        } else if (op == SS_OP_ST) {
            last_node = idx;
        }
    }
}

hipError_t launch_quotient_vm(hipStream_t st, const uint32_t *d_code, uint32_t n_entries, Fp *d_slots, uint64_t lanes,
                              const Fp &offset, const Fp &w, const Fp &wstep, uint64_t npoints, uint32_t xcd_split, Fp *out) {
    VmArgs a;
    a.code = d_code; a.slots = d_slots; a.out = out; a.offset = offset; a.w = w; a.wstep = wstep;
    a.n_entries = n_entries; a.npoints = npoints; a.xcd_split = xcd_split;
    hipLaunchKernelGGL(quotient_vm_kernel, dim3((uint32_t)(lanes / 256)), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ---- the compiled kernels' copy of the multiplier-only tables times 2^24 (quotient_gen.h QgScaleArgs; one launch per evaluation:
// a few thousand to a few hundred thousand entries) - "group sum x zerofier inverse" is then a product with the ten-step reduction
__global__ __launch_bounds__(256) void qg_scale_tables_kernel(QgScaleArgs a, uint64_t total) {
    Fp two24 = fp_zero(); two24.v[0] = 1u << 24;
    const Fp f = fp_to_mont(two24);
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t j = 0;
        while (j + 1 < a.n && i >= a.dst[j + 1]) ++j;          // the copies lie one after the other: table j covers [dst[j], dst[j] + len[j])
        a.out[i] = fp_mul(a.tables[a.src[j] + (i - a.dst[j])], f);
    }
}

hipError_t launch_qg_scale_tables(hipStream_t st, const QgScaleArgs &a, uint64_t total_felts) {
    if (a.n == 0 || total_felts == 0) return hipSuccess;
    uint32_t blocks = (uint32_t)((total_felts + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(qg_scale_tables_kernel, dim3(blocks), dim3(256), 0, st, a, total_felts);
    return hipGetLastError();
}

}  // namespace ss
