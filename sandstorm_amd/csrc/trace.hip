// trace.hip — the BASE trace made on the device ("next" row X1).
//
// ExecutionTrace::new (layouts/src/starknet/trace.rs:99-987, layouts/src/recursive/trace.rs:89-688,
// layouts/src/utils.rs:112-152) turns ~25 MB of `cairo-run` output (trace.bin: (ap, fp, pc) per cycle; memory.bin:
// (address, word) records) into 7 / 9 columns of 16 rows per cycle - 3.6 / 4.8 GB at 2^20 steps.  The reference makes them
// with rayon loops and sequential sorts on the host; this repo's host generator (host/trace_*.cpp) does the same, and a
// files -> proof call was then bound by the PCIe uploads of cells that are functions of the 25 MB.  Here the cells are made
// where they are used:
//   trace_cpu_kernel      one lane per Cairo cycle decodes the instruction, reads its three operands from the memory image,
//                         computes res / tmp0 / tmp1 / op0*op1 (trace.rs:177-244) into LDS; the workgroup then writes the
//                         cycle's 16 rows of the flags, memory-pool, range-check and auxiliary columns row by row
//                         (consecutive lanes = consecutive rows: whole 2 KiB runs per wave) and the pool's addresses as
//                         integers for the sort-free ordered memory below
//   trace_tile_kernel     a builtin's instances: ONE template of cells per distinct instance (the dummy instance: one for
//                         the whole run), written into every block that holds it; address cells are affine in the block
//   trace_rc_*            the range-check builtin's parts and the pool's ordered values / padding (utils.rs:357-380):
//                         runs located by binary search in a prefix array of the 65536-bin histogram
//   trace_runs_kernel     the diluted pool's ordered column the same way
//   trace_patch_kernel    the few cells that are neither (the diluted pool's padding values)
//   mem_*                 get_ordered_memory_accesses (utils.rs:112-152) without a sort: count per address through an LDS
//                         table per workgroup (an idling run reads the same handful of cells a million times), gaps by a
//                         flag scan, a prefix sum over the counts, rows by binary search in it; the reference's checks
//                         (continuous, single-valued, public memory = the address-0 cells) as error bits in a status word
// HBM-write bound: ~5 GB written once.  No floating point, no MFMA: integer decode and a handful of modular products per cell.
#include <hip/hip_runtime.h>
#include "inv252.h"
#include "kernels.h"

namespace ss {

namespace {

__device__ __forceinline__ Fp load_fp(const Fp *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    const uint4 a = q[0], b = q[1];
    Fp r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__device__ __forceinline__ void store_fp(Fp *p, const Fp &x) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    q[1] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
}
__device__ __forceinline__ Fp fp_of_words(const u64 w[4]) {          // a 256-bit integer, not yet in Montgomery form
    Fp r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.v[2 * i] = (u32)w[i]; r.v[2 * i + 1] = (u32)(w[i] >> 32); }
    return r;
}
__device__ __forceinline__ u32 sat32(u64 a) { return a > 0xffffffffull ? 0xffffffffu : (u32)a; }
__device__ __forceinline__ void status_error(u32 *status, u32 bits, u64 where) {
    atomicOr(&status[TRACE_ST_ERRORS], bits);
    atomicMax(&status[TRACE_ST_WHERE], ~sat32(where));              // the SMALLEST cycle / address that failed, as its complement
}

// ------------------------------------------------------------------------------------------------ memory image
// memory.bin: records of (u64 address, 32-byte little-endian word) -> image[address] (4 x u64); a cell no record names keeps
// the fill (all ones: not a field element, binary/src/lib.rs reads values < p)
__global__ __launch_bounds__(256) void trace_memory_image_kernel(const u64 *__restrict__ records, u64 n_records, u64 *__restrict__ image, u64 cells) {
    const u64 k = (u64)blockIdx.x * 256 + threadIdx.x;
    if (k >= n_records) return;
    const u64 a = records[5 * k];
    if (a >= cells) return;                                          // beyond every address a continuous memory of this trace can hold
#pragma unroll
    for (int i = 0; i < 4; ++i) image[4 * a + i] = records[5 * k + 1 + i];
}

// ------------------------------------------------------------------------------------------------ the CPU's cells
constexpr int CPU_CYC = 128;                 // cycles (and lanes) per workgroup
struct CpuRec {
    u64 word, ap, fp;
    u32 addr[4];                             // pc, op0, dst, op1 (saturated at 2^32 - 1: such an address fails the memory checks)
    u32 jnz, pad;
    Fp val[4];                               // the instruction word, op0, dst, op1 as Montgomery felts
    Fp res, tmp1, mul;
};
enum { REC_PC = 0, REC_OP0 = 1, REC_DST = 2, REC_OP1 = 3 };

__device__ __forceinline__ bool image_cell(const u64 *image, u64 cells, u64 a, u64 w[4]) {
    if (a >= cells) { w[0] = w[1] = w[2] = w[3] = 0; return false; }
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = image[4 * a + i];
    if (w[3] == ~0ull) { w[0] = w[1] = w[2] = w[3] = 0; return false; }
    return true;
}

__global__ __launch_bounds__(CPU_CYC) void trace_cpu_kernel(TraceLayout L, const u64 *__restrict__ states, u64 num_cycles, const u64 *__restrict__ image,
                                                            u64 cells, Fp pad_value, Fp rc_fill, Fp *__restrict__ flags, Fp *__restrict__ npc,
                                                            Fp *__restrict__ rc, Fp *__restrict__ aux, u32 *__restrict__ pool_addr, u32 *status) {
    __shared__ CpuRec rec[CPU_CYC];
    __shared__ uint8_t tab[48];              // the layout's placement tables (indexed per lane: LDS, not the kernel arguments)
    const u32 t = threadIdx.x;
    if (t < 8) tab[t] = L.npc_pair[t];
    if (t < 16) { tab[16 + t] = L.rc_cell[t]; tab[32 + t] = L.aux_cell[t]; }
    const u64 cyc0 = (u64)blockIdx.x * CPU_CYC, cyc = cyc0 + t;
    if (cyc < num_cycles) {
        CpuRec &r = rec[t];
        const u64 ap = states[3 * cyc], fp = states[3 * cyc + 1], pc = states[3 * cyc + 2];
        u32 err = 0;
        u64 iw[4], w0[4], wd[4], w1[4];
        if (!image_cell(image, cells, pc, iw)) err |= TRACE_ERR_MISSING_CELL;
        const u64 w = iw[0];
        if ((iw[1] | iw[2] | iw[3]) || (w >> 63)) err |= TRACE_ERR_NOT_INSTRUCTION;
        auto flag = [&](int f) { return (u32)((w >> (48 + f)) & 1); };
        const u64 dst_addr = (w & 0xffff) + (flag(0) ? fp : ap) - 0x8000, op0_addr = ((w >> 16) & 0xffff) + (flag(1) ? fp : ap) - 0x8000;
        if (!image_cell(image, cells, op0_addr, w0)) err |= TRACE_ERR_MISSING_CELL;
        if (!image_cell(image, cells, dst_addr, wd)) err |= TRACE_ERR_MISSING_CELL;
        const u32 src = flag(2) + 2 * flag(3) + 4 * flag(4);
        u64 base = 0;
        if (src == 0) { if (w0[1] | w0[2] | w0[3]) err |= TRACE_ERR_NOT_AN_ADDRESS; base = w0[0]; }
        else if (src == 1) base = pc;
        else if (src == 2) base = fp;
        else if (src == 4) base = ap;
        else err |= TRACE_ERR_BAD_OP1_SOURCE;
        const u64 op1_addr = ((w >> 32) & 0xffff) + base - 0x8000;
        if (!image_cell(image, cells, op1_addr, w1)) err |= TRACE_ERR_MISSING_CELL;
        const Fp inst = fp_to_mont(fp_of_words(iw)), op0 = fp_to_mont(fp_of_words(w0)), dst = fp_to_mont(fp_of_words(wd)), op1 = fp_to_mont(fp_of_words(w1));
        const u32 pc_update = flag(7) + 2 * flag(8) + 4 * flag(9), res_logic = flag(5) + 2 * flag(6);
        Fp res = fp_zero();
        if (pc_update == 4) { if (!fp_is_zero(dst)) res = fp_inv_safegcd(dst); }        // get_res of a conditional jump: dst^-1
        else if (res_logic == 0) res = op1;
        else if (res_logic == 1) res = fp_add(op0, op1);
        else if (res_logic == 2) res = fp_mul(op0, op1);
        else err |= TRACE_ERR_BAD_RES_LOGIC;
        r.word = w; r.ap = ap; r.fp = fp;
        r.addr[REC_PC] = sat32(pc); r.addr[REC_OP0] = sat32(op0_addr); r.addr[REC_DST] = sat32(dst_addr); r.addr[REC_OP1] = sat32(op1_addr);
        r.jnz = flag(9);
        r.val[REC_PC] = inst; r.val[REC_OP0] = op0; r.val[REC_DST] = dst; r.val[REC_OP1] = op1;
        r.res = res; r.tmp1 = r.jnz ? fp_mul(dst, res) : fp_zero(); r.mul = fp_mul(op0, op1);
        if (err) status_error(status, err, cyc);
    }
    __syncthreads();
    const u64 live = num_cycles - cyc0 < CPU_CYC ? num_cycles - cyc0 : CPU_CYC;       // cycles of this workgroup
    const u64 row0 = cyc0 * 16;
    for (u32 it = 0; it < 16; ++it) {
        const u32 lr = it * CPU_CYC + t, c = lr >> 4, o = lr & 15;
        if (c >= live) break;
        const CpuRec &r = rec[c];
        // flags: the instruction word's bits from bit o upwards (Flag::*.curr(): binary/src/lib.rs:740-772; bit 15 is zero)
        store_fp(&flags[row0 + lr], fp_from_u64(o == 15 ? 0 : (r.word >> (48 + o)) & ((1ull << (15 - o)) - 1)));
        // range-check column: the fill, the instruction's three offsets, zeros where the layout has other cells (the pool's cells
        // come with trace_rc_pool_kernel)
        {
            const u32 kind = tab[16 + o];
            Fp v = rc_fill;
            if (kind == TRACE_RC_ZERO) v = fp_zero();
            else if (kind != TRACE_RC_FILL) v = fp_from_u64((r.word >> (kind == TRACE_RC_OFF_DST ? 0 : kind == TRACE_RC_OFF_OP0 ? 16 : 32)) & 0xffff);
            store_fp(&rc[row0 + lr], v);
        }
        // auxiliary column
        {
            const u32 kind = tab[32 + o];
            Fp v = fp_zero();
            if (kind == TRACE_AUX_AP || kind == TRACE_AUX_FP) v = fp_from_u64(kind == TRACE_AUX_AP ? r.ap : r.fp);
            else if (kind == TRACE_AUX_TMP0) { if (r.jnz) v = r.val[REC_DST]; }
            else if (kind == TRACE_AUX_TMP1) v = r.tmp1;
            else if (kind == TRACE_AUX_MUL) v = r.mul;
            else if (kind == TRACE_AUX_RES) v = r.res;
            store_fp(&aux[row0 + lr], v);
        }
        // memory pool: (address, value) pairs - the CPU's four accesses, the public memory's (0, 0) slots, the padding pair elsewhere
        {
            const u32 kind = tab[o >> 1];
            Fp v;
            if (kind == TRACE_NPC_PAD) v = (o & 1) ? pad_value : fp_one();
            else if (kind == TRACE_NPC_PUBLIC) v = fp_zero();
            else v = (o & 1) ? r.val[kind - TRACE_NPC_PC] : fp_from_u64(r.addr[kind - TRACE_NPC_PC]);
            store_fp(&npc[row0 + lr], v);
        }
    }
    for (u32 it = 0; it < 8; ++it) {
        const u32 lp = it * CPU_CYC + t, c = lp >> 3, j = lp & 7;
        if (c >= live) break;
        const u32 kind = tab[j];
        pool_addr[cyc0 * 8 + lp] = kind == TRACE_NPC_PAD ? 1u : kind == TRACE_NPC_PUBLIC ? 0u : rec[c].addr[kind - TRACE_NPC_PC];
    }
}

// ------------------------------------------------------------------------------------------------ builtin instances
// entry e of a builtin's template: column, row offset inside the instance's block, kind (a value of the template, or the felt of
// the address addr_begin + addr_mult * block + arg); thread g writes entry g % n_e of block g / n_e
__global__ __launch_bounds__(256) void trace_tile_kernel(ColPtrs cols, u32 ncols, const TraceTileEntry *__restrict__ entries, u32 n_e, const Fp *__restrict__ values,
                                                         u32 n_templates, const u32 *__restrict__ tmpl_of_block, u64 nblocks, u64 step, u64 addr_begin,
                                                         u64 addr_mult, u32 *__restrict__ pool_addr) {
    __shared__ Fp *col_ptr[MAX_COLS];
    if (threadIdx.x < MAX_COLS) col_ptr[threadIdx.x] = (Fp *)cols.dst[threadIdx.x];
    __syncthreads();
    const u64 g = (u64)blockIdx.x * 256 + threadIdx.x;
    if (g >= nblocks * n_e) return;
    u64 i; u32 e;
    if (nblocks * n_e <= 0xffffffffull) { i = (u32)g / n_e; e = (u32)g - (u32)i * n_e; } else { i = g / n_e; e = (u32)(g - i * n_e); }
    const TraceTileEntry en = entries[e];
    if (en.col >= ncols || en.off >= step) return;                   // (the table is the caller's device memory: never write outside a column)
    const u64 row = i * step + en.off;
    Fp v;
    if (en.kind == TRACE_TILE_ADDRESS) {
        const u64 a = addr_begin + addr_mult * i + en.arg;
        v = fp_from_u64(a);
        pool_addr[row >> 1] = sat32(a);
    } else {
        const u32 tm = tmpl_of_block ? tmpl_of_block[i] : 0;
        if (tm >= n_templates) return;
        v = load_fp(&values[(u64)tm * n_e + e]);
    }
    store_fp(&col_ptr[en.col][row], v);
}

// ------------------------------------------------------------------------------------------------ range-check builtin and pool
// the pool's padding values in order (utils.rs:357-380 RangeCheckPool::get_ordered_values_with_padding's second half), then rc_hi
__device__ __forceinline__ u32 rc_padding(const TraceRcPlan &p, const uint16_t *padding, u64 j) { return j < p.n_padding ? padding[j] : p.rc_hi; }

// slot s of the range-check builtin (trace.rs:388-426 / recursive 236-284): a given instance, or a dummy whose 8 parts are the
// next padding values; the 16-bit parts go to the range-check column, (address, value) to the memory pool
__global__ __launch_bounds__(256) void trace_rc_builtin_kernel(TraceRcPlan p, const u64 *__restrict__ given /* index, lo, hi */, const uint16_t *__restrict__ padding,
                                                               Fp *__restrict__ rc, Fp *__restrict__ npc, u32 *__restrict__ pool_addr) {
    const u64 s = (u64)blockIdx.x * 256 + threadIdx.x;
    if (s >= p.n_slots) return;
    u64 lo, hi, index = s;
    if (s < p.n_given) { index = given[3 * s]; lo = given[3 * s + 1]; hi = given[3 * s + 2]; }
    else {
        lo = hi = 0;
        for (u32 k = 0; k < 8; ++k) { hi = (hi << 16) | (lo >> 48); lo = (lo << 16) | rc_padding(p, padding, 8 * (s - p.n_given) + k); }
    }
    const u64 base = s * p.slot_rows;
    for (u32 k = 0; k < 8; ++k) {
        const u32 sh = 16 * (7 - k);
        store_fp(&rc[base + (u64)p.part_stride * k + p.part_off], fp_from_u64((sh >= 64 ? hi >> (sh - 64) : lo >> sh) & 0xffff));
    }
    const u64 a = p.addr_begin + index;
    Fp v = fp_zero();
    v.v[0] = (u32)lo; v.v[1] = (u32)(lo >> 32); v.v[2] = (u32)hi; v.v[3] = (u32)(hi >> 32);
    store_fp(&npc[base + p.pair_off], fp_from_u64(a));
    store_fp(&npc[base + p.pair_off + 1], fp_to_mont(v));
    pool_addr[(base + p.pair_off) >> 1] = sat32(a);
}

// the value of slot k of an ordered pool: every value of [lo, hi] max(count, 1) times, ascending; first[j] = slots before value
// lo + j (j <= hi - lo + 1).  -> the value, or `beyond` from the pool's end on
__device__ __forceinline__ u32 run_value(const u32 *__restrict__ first, u32 n_values, u32 lo, u64 k, u32 beyond) {
    if (k >= first[n_values]) return beyond;
    u32 a = 0, b = n_values;                  // first[a] <= k < first[b]
    while (b - a > 1) { const u32 m = (a + b) >> 1; if (first[m] <= k) a = m; else b = m; }
    return lo + a;
}
// per cycle: its ordered values (every RANGE_CHECK_STEP rows) and, on odd cycles, the next unused padding value (trace.rs:263-292)
__global__ __launch_bounds__(256) void trace_rc_pool_kernel(TraceRcPlan p, const u32 *__restrict__ first, const uint16_t *__restrict__ padding, u64 num_cycles,
                                                            Fp *__restrict__ rc) {
    const u64 g = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 per = 16 / p.ordered_step;
    if (g < num_cycles * per) {
        const u64 cyc = g / per, j = g - cyc * per;
        store_fp(&rc[cyc * 16 + j * p.ordered_step + p.ordered_off], fp_from_u64(run_value(first, p.rc_hi - p.rc_lo + 1, p.rc_lo, g, p.rc_hi)));
    } else if (g < num_cycles * per + num_cycles / 2) {
        const u64 h = g - num_cycles * per;   // odd cycle 2 h + 1
        store_fp(&rc[(2 * h + 1) * 16 + p.unused_off], fp_from_u64(rc_padding(p, padding, p.pad0 + h)));
    }
}

__device__ __forceinline__ u64 dilute16(u32 v) {                     // bit i -> bit 4 i (builtins/src/bitwise/mod.rs dilute)
    u64 x = v & 0xffffu;
    x = (x | (x << 24)) & 0x000000ff000000ffull;
    x = (x | (x << 12)) & 0x000f000f000f000full;
    x = (x | (x << 6)) & 0x0303030303030303ull;
    x = (x | (x << 3)) & 0x1111111111111111ull;
    return x;
}
// slot k of an ordered pool into col[k * stride + off]: zero before the pool's first slot (first[0]), then the runs
__global__ __launch_bounds__(256) void trace_runs_kernel(Fp *__restrict__ col, u64 stride, u64 off, u64 slots, const u32 *__restrict__ first, u32 n_values,
                                                         u32 lo, u32 diluted) {
    const u64 k = (u64)blockIdx.x * 256 + threadIdx.x;
    if (k >= slots) return;
    Fp v = fp_zero();
    if (k >= first[0]) {
        const u32 x = run_value(first, n_values, lo, k, lo + n_values - 1);
        v = fp_from_u64(diluted ? dilute16(x) : x);
    }
    store_fp(&col[k * stride + off], v);
}
__global__ __launch_bounds__(256) void trace_patch_kernel(Fp *__restrict__ col, u64 col_rows, const u64 *__restrict__ rows, const u64 *__restrict__ values, u64 count) {
    const u64 k = (u64)blockIdx.x * 256 + threadIdx.x;
    if (k < count && rows[k] < col_rows) store_fp(&col[rows[k]], fp_from_u64(values[k]));
}

// ------------------------------------------------------------------------------------------------ exclusive prefix sums (u32)
constexpr u32 SCAN_LANES = 256, SCAN_PER = 16, SCAN_CHUNK = SCAN_LANES * SCAN_PER;
// exclusive scan of the 256 lane sums of a workgroup -> the lane's offset; *total = their sum
__device__ __forceinline__ u32 wg_exclusive(u32 mine, u32 *lds, u32 *total) {
    const u32 t = threadIdx.x;
    lds[t] = mine;
    __syncthreads();
    for (u32 d = 1; d < SCAN_LANES; d <<= 1) {
        const u32 add = t >= d ? lds[t - d] : 0;
        __syncthreads();
        lds[t] += add;
        __syncthreads();
    }
    const u32 incl = lds[t];
    *total = lds[SCAN_LANES - 1];
    __syncthreads();
    return incl - mine;
}
__global__ __launch_bounds__(SCAN_LANES) void scan_chunks_kernel(u32 *__restrict__ data, u64 n, u32 *__restrict__ sums) {
    __shared__ u32 lds[SCAN_LANES];
    const u64 base = (u64)blockIdx.x * SCAN_CHUNK + (u64)threadIdx.x * SCAN_PER;
    u32 v[SCAN_PER], s = 0;
#pragma unroll
    for (u32 k = 0; k < SCAN_PER; ++k) { v[k] = base + k < n ? data[base + k] : 0; s += v[k]; }
    u32 total;
    u32 run = wg_exclusive(s, lds, &total);
#pragma unroll
    for (u32 k = 0; k < SCAN_PER; ++k) { if (base + k < n) data[base + k] = run; run += v[k]; }
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
// one workgroup: the chunk sums in place (exclusive), the grand total to *total_out
__global__ __launch_bounds__(SCAN_LANES) void scan_sums_kernel(u32 *__restrict__ sums, u32 n_sums, u32 *total_out) {
    __shared__ u32 lds[SCAN_LANES];
    u32 carry = 0;
    for (u32 c0 = 0; c0 < n_sums; c0 += SCAN_CHUNK) {
        const u32 base = c0 + threadIdx.x * SCAN_PER;
        u32 v[SCAN_PER], s = 0;
#pragma unroll
        for (u32 k = 0; k < SCAN_PER; ++k) { v[k] = base + k < n_sums ? sums[base + k] : 0; s += v[k]; }
        u32 total;
        u32 run = carry + wg_exclusive(s, lds, &total);
#pragma unroll
        for (u32 k = 0; k < SCAN_PER; ++k) { if (base + k < n_sums) sums[base + k] = run; run += v[k]; }
        carry += total;
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}
__global__ __launch_bounds__(SCAN_LANES) void scan_apply_kernel_u32(u32 *__restrict__ data, u64 n, const u32 *__restrict__ sums) {
    const u32 add = sums[blockIdx.x];
    const u64 base = (u64)blockIdx.x * SCAN_CHUNK + threadIdx.x;
#pragma unroll
    for (u32 k = 0; k < SCAN_PER; ++k) { const u64 i = base + (u64)k * SCAN_LANES; if (i < n) data[i] += add; }
}
hipError_t exclusive_scan_u32(hipStream_t st, u32 *data, u64 n, u32 *sums, u32 *total_out) {
    const u32 chunks = (u32)((n + SCAN_CHUNK - 1) / SCAN_CHUNK);
    hipLaunchKernelGGL(scan_chunks_kernel, dim3(chunks), dim3(SCAN_LANES), 0, st, data, n, sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_LANES), 0, st, sums, chunks, total_out);
    hipLaunchKernelGGL(scan_apply_kernel_u32, dim3(chunks), dim3(SCAN_LANES), 0, st, data, n, (const u32 *)sums);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ ordered memory
constexpr u32 MEM_LANES = 256, MEM_PER = 16, MEM_SLOTS = 1024;
// accesses per address: count[a] (a >= 2) and rep[a] = the first access of a (its index in the pool); address 0 / 1 accesses are
// counted in the status words (most of a run: the public-memory slots and the padding pairs).  Hot addresses - an idling run reads
// its `jmp rel 0` a million times - are counted in an LDS table per workgroup and reach the global arrays once per workgroup.
__global__ __launch_bounds__(MEM_LANES) void mem_count_kernel(const u32 *__restrict__ pool_addr, u64 half, u32 cap, u32 *__restrict__ count, u32 *__restrict__ rep,
                                                              u32 *status) {
    __shared__ u32 key[MEM_SLOTS], cnt[MEM_SLOTS], first[MEM_SLOTS];
    __shared__ u32 agg[4];                    // zeros, ones, max address, ~min address
    const u32 t = threadIdx.x;
    for (u32 s = t; s < MEM_SLOTS; s += MEM_LANES) { key[s] = 0; cnt[s] = 0; first[s] = 0xffffffffu; }
    if (t < 4) agg[t] = 0;
    __syncthreads();
    u32 zeros = 0, ones = 0, top = 0, nlow = 0;
    for (u32 it = 0; it < MEM_PER; ++it) {
        const u64 j = (u64)blockIdx.x * (MEM_LANES * MEM_PER) + (u64)it * MEM_LANES + t;
        if (j >= half) break;
        const u32 a = pool_addr[j];
        top = a > top ? a : top;
        nlow = ~a > nlow ? ~a : nlow;
        if (a == 0) { ++zeros; continue; }
        if (a == 1) { ++ones; continue; }
        if (a > cap) { status_error(status, TRACE_ERR_ADDRESS_RANGE, a); continue; }
        const u32 s = a & (MEM_SLOTS - 1);
        const u32 old = atomicCAS(&key[s], 0u, a);
        if (old == 0 || old == a) { atomicAdd(&cnt[s], 1u); atomicMin(&first[s], (u32)j); }
        else { atomicAdd(&count[a], 1u); atomicMin(&rep[a], (u32)j); }
    }
    if (zeros) atomicAdd(&agg[0], zeros);
    if (ones) atomicAdd(&agg[1], ones);
    atomicMax(&agg[2], top);
    atomicMax(&agg[3], nlow);
    __syncthreads();
    for (u32 s = t; s < MEM_SLOTS; s += MEM_LANES)
        if (key[s]) { atomicAdd(&count[key[s]], cnt[s]); atomicMin(&rep[key[s]], first[s]); }
    if (t == 0) {
        if (agg[0]) atomicAdd(&status[TRACE_ST_ZEROS], agg[0]);
        if (agg[1]) atomicAdd(&status[TRACE_ST_ONES], agg[1]);
        atomicMax(&status[TRACE_ST_TOP], agg[2]);
        atomicMax(&status[TRACE_ST_NLOW], agg[3]);
    }
}
// the public memory's entries are accesses too (utils.rs:116-131): entry k is access half + k
__global__ __launch_bounds__(256) void mem_public_kernel(const u32 *__restrict__ pub_addr, u32 n_pub, u64 half, u32 cap, u32 *__restrict__ count,
                                                         u32 *__restrict__ rep, u32 *status) {
    const u32 k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_pub) return;
    const u32 a = pub_addr[k];
    atomicMax(&status[TRACE_ST_TOP], a);
    atomicMax(&status[TRACE_ST_NLOW], ~a);
    if (a == 0) { status_error(status, TRACE_ERR_PUBLIC_ZERO, 0); return; }
    if (a == 1) { atomicAdd(&status[TRACE_ST_ONES], 1u); return; }
    if (a > cap) { status_error(status, TRACE_ERR_ADDRESS_RANGE, a); return; }
    atomicAdd(&count[a], 1u);
    atomicMin(&rep[a], (u32)(half + k));
}
// gap fillers (trace.rs:594-625 / 890-925): the addresses strictly between the lowest and the highest accessed one that nothing
// accesses, ascending; gap number g takes the unused pool pair of cycle g as (address, 0)
__global__ __launch_bounds__(256) void mem_gap_flags_kernel(const u32 *__restrict__ count, u32 cap, const u32 *status, u32 *__restrict__ flag) {
    const u32 a = blockIdx.x * 256 + threadIdx.x;
    if (a > cap) return;
    const u32 top = status[TRACE_ST_TOP], low = ~status[TRACE_ST_NLOW];
    flag[a] = (a > low && a < top && a >= 2 && count[a] == 0) ? 1u : 0u;             // (address 1 always has the padding pairs; 0 is never a gap: low <= 0)
}
__global__ __launch_bounds__(256) void mem_gap_fill_kernel(const u32 *__restrict__ gap_index, u32 cap, u64 num_cycles, u32 unused_off, u32 *__restrict__ count,
                                                           u32 *__restrict__ rep, Fp *__restrict__ npc, u32 *__restrict__ pool_addr, u32 *status) {
    const u32 a = blockIdx.x * 256 + threadIdx.x;
    if (a > cap) return;
    const u32 top = status[TRACE_ST_TOP], low = ~status[TRACE_ST_NLOW];
    if (!(a > low && a < top && a >= 2 && count[a] == 0)) return;
    const u32 g = gap_index[a];
    if (g >= num_cycles) { status_error(status, TRACE_ERR_TOO_MANY_GAPS, a); return; }
    const u64 row = (u64)g * 16 + unused_off;
    store_fp(&npc[row], fp_from_u64(a));
    store_fp(&npc[row + 1], fp_zero());
    pool_addr[row >> 1] = a;
    count[a] = 1;
    rep[a] = (u32)(row >> 1);
}
__device__ __forceinline__ Fp mem_value_of(u32 idx, u64 half, const Fp *npc, const Fp *pub_value) {
    return idx < half ? load_fp(&npc[2 * (u64)idx + 1]) : load_fp(&pub_value[idx - half]);
}
// the reference's checks (utils.rs:132-150): one value per address, the pool's address-0 cells are exactly the public-memory
// cells, memory starts at address 1 and has no holes; then the counts become the prefix array's input: start[a] = count of a
// (address 1: the padding pairs that are left + the public memory's own entries at 1)
__global__ __launch_bounds__(256) void mem_check_kernel(const u32 *__restrict__ pool_addr, u64 half, const u32 *__restrict__ pub_addr, u32 n_pub,
                                                        const Fp *__restrict__ pub_value, Fp pad_value, const u32 *__restrict__ rep, const Fp *__restrict__ npc,
                                                        u32 cap, u32 *status) {
    const u64 j = (u64)blockIdx.x * 256 + threadIdx.x;
    if (j >= half + n_pub) return;
    const u32 a = j < half ? pool_addr[j] : pub_addr[j - half];
    if (a == 0 || a > cap) return;
    const Fp mine = mem_value_of((u32)j, half, npc, pub_value);
    const Fp want = a == 1 ? pad_value : mem_value_of(rep[a], half, npc, pub_value);
    if (!fp_eq(mine, want)) status_error(status, TRACE_ERR_NOT_SINGLE_VALUED, a);
}
__global__ __launch_bounds__(256) void mem_starts_kernel(const u32 *__restrict__ count, u32 cap, u64 public_cells, u32 n_pub, u32 *status, u32 *__restrict__ start) {
    const u32 a = blockIdx.x * 256 + threadIdx.x;
    if (a > cap + 1) return;
    const u32 top = status[TRACE_ST_TOP];
    // the padding pairs the gaps took are gone; the public memory's padding entries (1, pad) come on top (utils.rs:120-124)
    const u32 ones = status[TRACE_ST_ONES] - status[TRACE_ST_GAPS] + (u32)(public_cells - n_pub);
    u32 c = 0;
    if (a == 1) c = ones;
    else if (a >= 2 && a <= top && a <= cap) c = count[a];
    if (a >= 1 && a <= top && a <= cap && c == 0) status_error(status, TRACE_ERR_NOT_CONTINUOUS, a);
    if (a == 0) {
        if (status[TRACE_ST_ZEROS] != public_cells) status_error(status, TRACE_ERR_PUBLIC_CELLS, 0);
        if (top > cap) status_error(status, TRACE_ERR_ADDRESS_RANGE, top);
    }
    start[a] = c;
}
// row pair j of the ordered column: (a, value of a) for the address whose run holds j (utils.rs:138-152)
__global__ __launch_bounds__(256) void mem_fill_kernel(const u32 *__restrict__ start, u64 half, u32 cap, const u32 *__restrict__ rep, const Fp *__restrict__ npc,
                                                       const Fp *__restrict__ pub_value, Fp pad_value, Fp *__restrict__ mem_col, u32 *status) {
    const u64 j = (u64)blockIdx.x * 256 + threadIdx.x;
    if (j >= half) return;
    u32 top = status[TRACE_ST_TOP];
    if (top > cap) top = cap;
    Fp av = fp_zero(), vv = fp_zero();
    if (j >= start[top + 1]) {                                       // the runs do not fill the column
        status_error(status, TRACE_ERR_FILL, top);
    } else {
        u32 a = 0, b = top + 1;                                      // start[a] <= j < start[b]
        while (b - a > 1) { const u32 m = (a + b) >> 1; if (start[m] <= j) a = m; else b = m; }
        av = fp_from_u64(a);
        vv = a <= 1 ? pad_value : mem_value_of(rep[a], half, npc, pub_value);
    }
    store_fp(&mem_col[2 * j], av);
    store_fp(&mem_col[2 * j + 1], vv);
}
__global__ void mem_fill_total_kernel(const u32 *__restrict__ start, u64 half, u32 cap, u32 *status) {
    if (threadIdx.x || blockIdx.x) return;
    u32 top = status[TRACE_ST_TOP];
    if (top > cap) top = cap;
    if (start[top + 1] != half) status_error(status, TRACE_ERR_FILL, top);
    if (start[2] == 0) status_error(status, TRACE_ERR_NO_ONES, 1);     // start[2] = accesses of address 1 (start[0] = start[1] = 0)
}

inline dim3 grid_for(u64 lanes, u32 block) { return dim3((u32)((lanes + block - 1) / block)); }

}  // namespace

hipError_t launch_trace_memory_image(hipStream_t st, const u64 *d_records, u64 n_records, u64 *d_image, u64 cells) {
    hipError_t e = hipMemsetAsync(d_image, 0xff, cells * 32, st);
    if (e != hipSuccess || !n_records) return e;
    hipLaunchKernelGGL(trace_memory_image_kernel, grid_for(n_records, 256), dim3(256), 0, st, d_records, n_records, d_image, cells);
    return hipGetLastError();
}
hipError_t launch_trace_cpu(hipStream_t st, const TraceLayout &L, const u64 *d_states, u64 num_cycles, const u64 *d_image, u64 cells, const Fp &pad_value,
                            u64 rc_fill, Fp *flags, Fp *npc, Fp *rc, Fp *aux, u32 *d_pool_addr, u32 *d_status) {
    hipLaunchKernelGGL(trace_cpu_kernel, grid_for(num_cycles, CPU_CYC), dim3(CPU_CYC), 0, st, L, d_states, num_cycles, d_image, cells, pad_value,
                       fp_from_u64(rc_fill), flags, npc, rc, aux, d_pool_addr, d_status);
    return hipGetLastError();
}
hipError_t launch_trace_tile(hipStream_t st, const ColPtrs &cols, u32 ncols, const TraceTileEntry *d_entries, u32 n_entries, const Fp *d_values, u32 n_templates,
                             const u32 *d_tmpl_of_block, u64 nblocks, u64 step, u64 addr_begin, u64 addr_mult, u32 *d_pool_addr) {
    if (!nblocks || !n_entries) return hipSuccess;
    hipLaunchKernelGGL(trace_tile_kernel, grid_for(nblocks * n_entries, 256), dim3(256), 0, st, cols, ncols, d_entries, n_entries, d_values, n_templates,
                       d_tmpl_of_block, nblocks, step, addr_begin, addr_mult, d_pool_addr);
    return hipGetLastError();
}
hipError_t launch_trace_rc_builtin(hipStream_t st, const TraceRcPlan &p, const u64 *d_given, const uint16_t *d_padding, Fp *rc, Fp *npc, u32 *d_pool_addr) {
    if (!p.n_slots) return hipSuccess;
    hipLaunchKernelGGL(trace_rc_builtin_kernel, grid_for(p.n_slots, 256), dim3(256), 0, st, p, d_given, d_padding, rc, npc, d_pool_addr);
    return hipGetLastError();
}
hipError_t launch_trace_rc_pool(hipStream_t st, const TraceRcPlan &p, const u32 *d_first, const uint16_t *d_padding, u64 num_cycles, Fp *rc) {
    hipLaunchKernelGGL(trace_rc_pool_kernel, grid_for(num_cycles * (16 / p.ordered_step) + num_cycles / 2, 256), dim3(256), 0, st, p, d_first, d_padding, num_cycles, rc);
    return hipGetLastError();
}
hipError_t launch_trace_runs(hipStream_t st, Fp *col, u64 stride, u64 off, u64 slots, const u32 *d_first, u32 n_values, u32 lo, bool diluted) {
    if (!slots) return hipSuccess;
    hipLaunchKernelGGL(trace_runs_kernel, grid_for(slots, 256), dim3(256), 0, st, col, stride, off, slots, d_first, n_values, lo, diluted ? 1u : 0u);
    return hipGetLastError();
}
hipError_t launch_trace_patch(hipStream_t st, Fp *col, u64 col_rows, const u64 *d_rows, const u64 *d_values, u64 count) {
    if (!count) return hipSuccess;
    hipLaunchKernelGGL(trace_patch_kernel, grid_for(count, 256), dim3(256), 0, st, col, col_rows, d_rows, d_values, count);
    return hipGetLastError();
}

// scratch (u32 words): count, rep, start (cap + 2 each), the scans' chunk sums
uint64_t trace_memory_scratch_words(u64 half) {
    const u64 cap = half;
    return 3 * (cap + 2) + 2 * ((cap + 2 + SCAN_CHUNK - 1) / SCAN_CHUNK + 1) + 16;
}
hipError_t launch_trace_ordered_memory(hipStream_t st, const TraceMemoryArgs &m, u32 *scratch) {
    const u64 half = m.n / 2;
    const u32 cap = (u32)half;                // continuous memory has an access per address 1 .. top and n / 2 accesses in all
    u32 *count = scratch, *rep = count + (cap + 2), *start = rep + (cap + 2), *sums = start + (cap + 2);
    hipError_t e;
    if ((e = hipMemsetAsync(count, 0, (size_t)(cap + 2) * 4, st)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(rep, 0xff, (size_t)(cap + 2) * 4, st)) != hipSuccess) return e;
    hipLaunchKernelGGL(mem_count_kernel, grid_for(half, MEM_LANES * MEM_PER), dim3(MEM_LANES), 0, st, (const u32 *)m.d_pool_addr, half, cap, count, rep, m.d_status);
    if (m.n_public) hipLaunchKernelGGL(mem_public_kernel, grid_for(m.n_public, 256), dim3(256), 0, st, m.d_public_addr, m.n_public, half, cap, count, rep, m.d_status);
    // gaps: flags -> their rank among the gaps (start[] as the temporary) -> the pool pairs
    hipLaunchKernelGGL(mem_gap_flags_kernel, grid_for((u64)cap + 1, 256), dim3(256), 0, st, (const u32 *)count, cap, (const u32 *)m.d_status, start);
    if ((e = exclusive_scan_u32(st, start, (u64)cap + 1, sums, m.d_status + TRACE_ST_GAPS)) != hipSuccess) return e;
    hipLaunchKernelGGL(mem_gap_fill_kernel, grid_for((u64)cap + 1, 256), dim3(256), 0, st, (const u32 *)start, cap, m.n / 16, m.unused_off, count, rep, m.npc,
                       m.d_pool_addr, m.d_status);
    hipLaunchKernelGGL(mem_check_kernel, grid_for(half + m.n_public, 256), dim3(256), 0, st, (const u32 *)m.d_pool_addr, half, m.d_public_addr, m.n_public,
                       m.d_public_value, m.pad_value, (const u32 *)rep, (const Fp *)m.npc, cap, m.d_status);
    hipLaunchKernelGGL(mem_starts_kernel, grid_for((u64)cap + 2, 256), dim3(256), 0, st, (const u32 *)count, cap, m.public_cells, m.n_public, m.d_status, start);
    if ((e = exclusive_scan_u32(st, start, (u64)cap + 2, sums, nullptr)) != hipSuccess) return e;
    hipLaunchKernelGGL(mem_fill_total_kernel, dim3(1), dim3(64), 0, st, (const u32 *)start, half, cap, m.d_status);
    hipLaunchKernelGGL(mem_fill_kernel, grid_for(half, 256), dim3(256), 0, st, (const u32 *)start, half, cap, (const u32 *)rep, (const Fp *)m.npc, m.d_public_value,
                       m.pad_value, m.memory, m.d_status);
    return hipGetLastError();
}

}  // namespace ss
