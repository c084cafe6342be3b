#include "trace_recursive.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>

#include "../../include/sandstorm_hip.h"
#include <omp.h>
#include "trace_common.hpp"
#include "device_trace.hpp"

namespace ssh {

using namespace tracedetail;

namespace {

constexpr uint64_t CYCLE_HEIGHT = 16, PUBLIC_MEMORY_STEP = 16, RANGE_CHECK_STEP = 4;
constexpr uint64_t PEDERSEN_BUILTIN_RATIO = 128, RANGE_CHECK_BUILTIN_RATIO = 8, RANGE_CHECK_BUILTIN_PARTS = 8, BITWISE_RATIO = 8;
enum { COL_FLAGS, COL_DILUTED_UNORDERED, COL_DILUTED_ORDERED, COL_NPC, COL_MEMORY, COL_RANGE_CHECK, COL_AUXILIARY, NUM_COLS };
// virtual-column offsets (layouts/src/recursive/air.rs:1324-1695)
enum { NPC_PC = 0, NPC_INSTRUCTION = 1, NPC_PUB_MEM_ADDR = 2, NPC_MEM_OP0_ADDR = 4, NPC_MEM_DST_ADDR = 8, NPC_MEM_OP1_ADDR = 12, NPC_UNUSED_ADDR = 14,
       NPC_PEDERSEN_INPUT0_ADDR = 10, NPC_PEDERSEN_INPUT1_ADDR = 1034, NPC_PEDERSEN_OUTPUT_ADDR = 522, NPC_RANGE_CHECK128_ADDR = 74,
       NPC_BITWISE_POOL_ADDR = 26, NPC_BITWISE_X_OR_Y_ADDR = 42 };
enum { RC_OFF_DST = 0, RC_ORDERED = 2, RC_OFF_OP1 = 4, RC_OFF_OP0 = 8, RC_UNUSED = 12 };
enum { AUX_AP = 1, AUX_TMP0 = 3, AUX_OP0_MUL_OP1 = 5, AUX_FP = 9, AUX_TMP1 = 11, AUX_RES = 13 };
// OpenMP only pays above this many cycles (thread start-up is ~50 ms of a 150 ms job at 2^14 cycles).
// SSH_TRACE_PARALLEL_MIN overrides it (tests force the parallel path on the small example with =1).
uint64_t parallel_min_cycles() {
    if (const char *e = getenv("SSH_TRACE_PARALLEL_MIN")) return strtoull(e, nullptr, 10);
    return 1ull << 16;
}

}  // namespace

std::vector<RegisterState> read_register_states(const uint8_t *data, size_t len) {
    if (len % 24) fail("trace file is not a sequence of (ap, fp, pc) u64 triples");
    std::vector<RegisterState> out(len / 24);
    for (size_t i = 0; i < out.size(); ++i) { memcpy(&out[i].ap, data + 24 * i, 8); memcpy(&out[i].fp, data + 24 * i + 8, 8); memcpy(&out[i].pc, data + 24 * i + 16, 8); }
    return out;
}

RegisterStates::RegisterStates(const uint8_t *bytes, size_t len) : data(bytes), count(len / 24) {
    if (len % 24) fail("trace file is not a sequence of (ap, fp, pc) u64 triples");
}

void read_memory(const uint8_t *data, size_t len, std::vector<U256> &memory, std::vector<uint8_t> &present) {
    if (len % 40) fail("memory file is not a sequence of (u64 address, 32-byte word) records");
    uint64_t max_addr = 0;
    for (size_t o = 0; o < len; o += 40) { uint64_t a; memcpy(&a, data + o, 8); max_addr = std::max(max_addr, a); }
    memory.assign(len ? max_addr + 1 : 0, U256{});
    present.assign(memory.size(), 0);
    for (size_t o = 0; o < len; o += 40) {
        uint64_t a;
        memcpy(&a, data + o, 8);
        memcpy(memory[a].data(), data + o + 8, 32);
        present[a] = 1;
    }
}

std::vector<std::vector<Felt>> recursive_base_trace(const RegisterStates &states, const std::vector<U256> &memory,
                                                    const std::vector<uint8_t> &present, const AirPublicInput &pi, const PrivateInput &priv) {
    std::vector<std::vector<Felt>> cols(NUM_COLS);
    Felt *out[NUM_COLS];
    for (int c = 0; c < NUM_COLS; ++c) { cols[c].resize(states.size() * CYCLE_HEIGHT); out[c] = cols[c].data(); }
    recursive_base_trace_into(out, states, memory, present, pi, priv);
    return cols;
}

namespace {

using U256x2 = std::pair<U256, U256>;

// body(k) for k in [0, count), on all host threads if `par`; the first exception a body throws is rethrown on the caller's thread
template <class Body> void parallel_for(uint64_t count, bool par, const Body &body) {
    std::exception_ptr err;
#pragma omp parallel for schedule(static) if (par)
    for (uint64_t k = 0; k < count; ++k) {
        try { body(k); } catch (...) {
#pragma omp critical(ssh_trace_error_rec)
            if (!err) err = std::current_exception();
        }
    }
    if (err) std::rethrow_exception(err);
}

const ss_trace_layout &cpu_layout() {                     // the CPU's cells in a cycle's 16 rows (recursive air.rs:1324-1695: Npc, RangeCheck, Auxiliary)
    static const ss_trace_layout l = [] {
        ss_trace_layout v;
        for (int j = 0; j < 8; ++j) v.npc_pair[j] = SS_TRACE_NPC_PAD;
        v.npc_pair[NPC_PC / 2] = SS_TRACE_NPC_PC; v.npc_pair[NPC_MEM_OP0_ADDR / 2] = SS_TRACE_NPC_OP0; v.npc_pair[NPC_MEM_DST_ADDR / 2] = SS_TRACE_NPC_DST;
        v.npc_pair[NPC_MEM_OP1_ADDR / 2] = SS_TRACE_NPC_OP1; v.npc_pair[NPC_PUB_MEM_ADDR / 2] = SS_TRACE_NPC_PUBLIC;
        for (int o = 0; o < 16; ++o) { v.rc_cell[o] = SS_TRACE_RC_FILL; v.aux_cell[o] = SS_TRACE_AUX_ZERO; }
        v.rc_cell[RC_OFF_DST] = SS_TRACE_RC_OFF_DST; v.rc_cell[RC_OFF_OP1] = SS_TRACE_RC_OFF_OP1; v.rc_cell[RC_OFF_OP0] = SS_TRACE_RC_OFF_OP0;
        v.aux_cell[AUX_AP] = SS_TRACE_AUX_AP; v.aux_cell[AUX_FP] = SS_TRACE_AUX_FP; v.aux_cell[AUX_TMP0] = SS_TRACE_AUX_TMP0; v.aux_cell[AUX_TMP1] = SS_TRACE_AUX_TMP1;
        v.aux_cell[AUX_OP0_MUL_OP1] = SS_TRACE_AUX_OP0_MUL_OP1; v.aux_cell[AUX_RES] = SS_TRACE_AUX_RES;
        return v;
    }();
    return l;
}

struct Inputs {
    const RegisterStates &states;
    const Mem &mem;
    const AirPublicInput &pi;
    const PrivateInput &priv;
    uint64_t num_cycles, n;
    Felt pad_value;
    bool par;
};

// ---- the host backend: the sections' cells straight into the caller's columns (pinned host memory the upload reads straight from: the
// GpuAllocator seam of layouts/src/recursive/trace.rs:115-120); every cell is written
struct HostBackend {
    const Inputs &in;
    Felt *const *out;
    const std::function<void(int)> *column_done;
    AddrArray npc_addr;                                  // the address half of the pool, as integers (sorting, gap search); assigned by its cycle
    const bool timing = getenv("SSH_TRACE_TIMING") != nullptr;   // diagnostic: per-section wall time on stderr
    std::chrono::steady_clock::time_point t_prev = std::chrono::steady_clock::now();
    HostBackend(const Inputs &in_, Felt *const *out_, const std::function<void(int)> *cd) : in(in_), out(out_), column_done(cd), npc_addr(in_.n / 2) {}
    void mark(const char *what) {
        if (!timing) return;
        auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[trace timing] %-22s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    }
    void done(std::initializer_list<int> cs) { if (column_done && *column_done) for (int c : cs) (*column_done)(c); }
    void set_pair(uint64_t row, uint64_t address, const Felt &value) { out[COL_NPC][row] = felt_from_u64(address); out[COL_NPC][row + 1] = value; npc_addr[row / 2] = address; }

    // a builtin's blocks: block i holds template of_block[i]; zero_col (or -1): a column whose rows of the block are zero where the
    // instance has no cell (the section owns the column)
    template <class Place> void builtin(const char *what, const std::vector<uint32_t> &of_block, uint32_t, uint64_t step, uint64_t begin, uint64_t per_block, int zero_col,
                                        const Place &place) {
        const Felt zero = felt_from_u64(0);
        parallel_for(of_block.size(), in.par, [&](uint64_t i) {
            if (zero_col >= 0) for (uint64_t o = 0; o < step; ++o) out[zero_col][i * step + o] = zero;
            HostSink sink{out, &npc_addr, COL_NPC, i * step, begin + per_block * i};
            place(sink, of_block[i]);
        });
        mark(what);
    }
    // the diluted pool (trace.rs:560-588): padding values into free cells of the unordered column, the ordered column whole
    void diluted(const DilutedPlan &dp, const std::vector<uint64_t> &pad_rows, const std::vector<uint64_t> &pad_values) {
        Felt *const un_col = out[COL_DILUTED_UNORDERED], *const od_col = out[COL_DILUTED_ORDERED];
        for (size_t k = 0; k < pad_rows.size(); ++k) un_col[pad_rows[k]] = felt_from_u64(pad_values[k]);
        const std::vector<uint32_t> &first_row = dp.first;
        parallel_for(1u << DILUTED_N_BITS, in.par, [&](uint64_t v) {
            const Felt f = felt_from_u64(dilute((uint32_t)v));
            if (first_row[v + 1] - first_row[v] < 4096) { for (uint64_t r2 = first_row[v]; r2 < first_row[v + 1]; ++r2) od_col[r2] = f; }
        });
        for (uint32_t v = 0; v < (1u << DILUTED_N_BITS); ++v)             // the long runs (value 0 of the dummy instances) by all threads
            if (first_row[v + 1] - first_row[v] >= 4096) {
                const Felt f = felt_from_u64(dilute(v));
                const uint64_t r0 = first_row[v];
                parallel_for(first_row[v + 1] - r0, in.par, [&](uint64_t j) { od_col[r0 + j] = f; });
            }
        mark("bitwise + diluted");
        done({COL_DILUTED_UNORDERED, COL_DILUTED_ORDERED});
    }
    // CPU cells (trace.rs:172-232).  The generator is bound by the host's memory traffic, so the four columns this section fills are
    // written ONCE: a cycle's 16 rows of each are made in a block on the stack - padding first, then what the cycle puts there - and
    // stored row after row; a builtin's cells of these columns get the padding here and their values in the builtin's section
    void cpu_section(const RcPoolPlan &pool, const std::vector<uint32_t> &rc_count, uint64_t rc_fill) {
        const uint64_t num_cycles = in.num_cycles;
        const RegisterStates &states = in.states;
        const Mem &mem = in.mem;
        Felt *const flags = out[COL_FLAGS], *const npc = out[COL_NPC], *const rc_col = out[COL_RANGE_CHECK], *const aux = out[COL_AUXILIARY];
        const Felt zero = felt_from_u64(0), pad_addr = felt_from_u64(1), pad_value = in.pad_value, rc_max_f = felt_from_u64(rc_fill);
        CountArray ordered_vals;
        ordered_runs(rc_count, pool.lo, pool.hi, ordered_vals);
        const uint64_t per = CYCLE_HEIGHT / RANGE_CHECK_STEP;
        mark("range check");
        constexpr uint64_t JNZ_BLOCK = 512;                  // cycles whose conditional jumps share one inversion (JnzInverses)
        parallel_for((num_cycles + JNZ_BLOCK - 1) / JNZ_BLOCK, in.par, [&](uint64_t blk_i) {
            const uint64_t first_cycle = blk_i * JNZ_BLOCK, end_cycle = std::min(num_cycles, first_cycle + JNZ_BLOCK);
            JnzInverses jnz(states, mem, first_cycle, end_cycle);
            for (uint64_t cycle = first_cycle; cycle < end_cycle; ++cycle) {
                const uint64_t r = cycle * CYCLE_HEIGHT, pc = states[cycle].pc, ap = states[cycle].ap, fp = states[cycle].fp;
                const U256 &iw = mem.at(pc);
                const Word w{iw[0]};
                if ((iw[1] | iw[2] | iw[3]) || w.flag(F_ZERO)) fail("memory cell " + std::to_string(pc) + " is not an instruction");
                const uint64_t dst_addr = w.off_dst() + (w.flag(F_DST_REG) ? fp : ap) - HALF_OFFSET;
                const uint64_t op0_addr = w.off_op0() + (w.flag(F_OP0_REG) ? fp : ap) - HALF_OFFSET;
                const int src = w.op1_src();
                if (src != 0 && src != 1 && src != 2 && src != 4) fail("invalid op1 source");
                const uint64_t base = src == 0 ? mem.small(op0_addr) : src == 1 ? pc : src == 2 ? fp : ap;
                const uint64_t op1_addr = w.off_op1() + base - HALF_OFFSET;
                const Felt dst = felt_from_canonical(mem.at(dst_addr)), op0 = felt_from_canonical(mem.at(op0_addr)), op1 = felt_from_canonical(mem.at(op1_addr));
                Felt res;
                if (w.pc_update() == 4) res = felt_is_zero(dst) ? zero : jnz.take(dst);           // get_res: dst^-1 on a jnz
                else if (w.res_logic() == 0) res = op1;
                else if (w.res_logic() == 1) res = felt_add(op0, op1);
                else if (w.res_logic() == 2) res = felt_mul(op0, op1);
                else fail("invalid res logic");
                const Felt tmp0 = w.flag(F_PC_JNZ) ? dst : zero;
                Felt blk[CYCLE_HEIGHT];
                for (int f = 0; f < 16; ++f) flags[r + f] = felt_from_u64(w.flag_prefix(f));
                // memory pool: (address, value) pairs, the padding pair where the CPU has none
                for (uint64_t o = 0; o < CYCLE_HEIGHT; o += 2) { blk[o] = pad_addr; blk[o + 1] = pad_value; npc_addr[(r + o) / 2] = 1; }
                auto pair = [&](uint64_t o, uint64_t address, const Felt &value) { blk[o] = felt_from_u64(address); blk[o + 1] = value; npc_addr[(r + o) / 2] = address; };
                pair(NPC_PC, pc, felt_from_canonical(iw));
                pair(NPC_MEM_OP0_ADDR, op0_addr, op0);
                pair(NPC_MEM_DST_ADDR, dst_addr, dst);
                pair(NPC_MEM_OP1_ADDR, op1_addr, op1);
                pair(NPC_PUB_MEM_ADDR, 0, zero);
                for (uint64_t o = 0; o < CYCLE_HEIGHT; ++o) npc[r + o] = blk[o];
                // range-check column: the declared maximum, the instruction's offsets, the odd cycles' next padding value, the ordered values
                for (uint64_t o = 0; o < CYCLE_HEIGHT; ++o) blk[o] = rc_max_f;
                blk[RC_OFF_DST] = felt_from_u64(w.off_dst()); blk[RC_OFF_OP1] = felt_from_u64(w.off_op1()); blk[RC_OFF_OP0] = felt_from_u64(w.off_op0());
                if (cycle % 2 == 1) blk[RC_UNUSED] = felt_from_u64(pool.pad(pool.pad0 + cycle / 2));
                for (uint64_t o = 0; o < CYCLE_HEIGHT; o += RANGE_CHECK_STEP) {
                    const uint64_t at = cycle * per + o / RANGE_CHECK_STEP;
                    blk[o + RC_ORDERED] = felt_from_u64(at < ordered_vals.size() ? ordered_vals[at] : pool.hi);
                }
                for (uint64_t o = 0; o < CYCLE_HEIGHT; ++o) rc_col[r + o] = blk[o];
                // auxiliary column: zero where no section writes
                for (uint64_t o = 0; o < CYCLE_HEIGHT; ++o) blk[o] = zero;
                blk[AUX_TMP0] = tmp0; blk[AUX_TMP1] = felt_mul(tmp0, res);
                blk[AUX_AP] = felt_from_u64(ap); blk[AUX_FP] = felt_from_u64(fp);
                blk[AUX_OP0_MUL_OP1] = felt_mul(op0, op1); blk[AUX_RES] = res;
                for (uint64_t o = 0; o < CYCLE_HEIGHT; ++o) aux[r + o] = blk[o];
            }
        });
        mark("cpu cells");
        done({COL_FLAGS});
    }
    // the range-check builtin's slots (trace.rs:236-284): the given instances, then dummies made of the pool's padding values
    void rc_builtin(const ss_trace_rc_plan &plan, const RcPoolPlan &pool, const std::vector<uint64_t> &given3) {
        done({COL_AUXILIARY});
        Felt *const rc_col = out[COL_RANGE_CHECK];
        parallel_for(plan.n_slots, in.par, [&](uint64_t s) {
            uint64_t lo = 0, hi = 0, index = s;
            if (s < plan.n_given) { index = given3[3 * s]; lo = given3[3 * s + 1]; hi = given3[3 * s + 2]; }
            else for (unsigned k = 0; k < RANGE_CHECK_BUILTIN_PARTS; ++k) { hi = (hi << 16) | (lo >> 48); lo = (lo << 16) | pool.pad(8 * (s - plan.n_given) + k); }
            const uint64_t base = s * plan.slot_rows;
            for (unsigned k = 0; k < RANGE_CHECK_BUILTIN_PARTS; ++k) {
                const unsigned sh = 16 * (RANGE_CHECK_BUILTIN_PARTS - 1 - k);
                rc_col[base + plan.part_stride * k + plan.part_off] = felt_from_u64((sh >= 64 ? hi >> (sh - 64) : lo >> sh) & 0xffff);
            }
            set_pair(base + plan.pair_off, plan.addr_begin + index, felt_from_canonical(U256{lo, hi, 0, 0}));
        });
        mark("rc builtin");
        done({COL_RANGE_CHECK});
    }
    // gap fillers (trace.rs:594-625) and the sorted memory (get_ordered_memory_accesses, utils.rs:112-152)
    void memory() {
        const std::vector<uint64_t> gaps = memory_gaps(npc_addr, in.pi.public_memory, in.num_cycles);
        if (gaps.size() > in.num_cycles) fail("more memory gaps than cycles to hold them");
        for (size_t k = 0; k < gaps.size(); ++k) set_pair(k * CYCLE_HEIGHT + NPC_UNUSED_ADDR, gaps[k], felt_from_u64(0));
        mark("gap fill");
        done({COL_NPC});
        ordered_memory_into(out[COL_MEMORY], in.n, npc_addr, out[COL_NPC], in.n / PUBLIC_MEMORY_STEP, in.pi.public_memory, in.pad_value);
        mark("sorted memory");
        done({COL_MEMORY});
    }
};

// ---- the device backend: the same sections as uploads of plans / templates and kernel launches (device_trace.hpp, csrc/trace.hip)
struct DeviceBackend {
    const Inputs &in;
    DeviceTrace &dt;
    template <class Place> void builtin(const char *, const std::vector<uint32_t> &of_block, uint32_t n_templates, uint64_t step, uint64_t begin, uint64_t per_block, int zero_col,
                                        const Place &place) {
        if (zero_col >= 0) dt.zero_column(zero_col);
        dt.builtin(n_templates, of_block, step, begin, per_block, place);
    }
    void diluted(const DilutedPlan &dp, const std::vector<uint64_t> &pad_rows, const std::vector<uint64_t> &pad_values) {
        dt.patch(COL_DILUTED_UNORDERED, pad_rows, pad_values);
        dt.ordered_runs(COL_DILUTED_ORDERED, 1, 0, in.n, dp.first, 0, true);
    }
    void cpu_section(const RcPoolPlan &pool, const std::vector<uint32_t> &rc_count, uint64_t rc_fill) {
        dt.cpu_cells(cpu_layout(), COL_FLAGS, COL_NPC, COL_RANGE_CHECK, COL_AUXILIARY, in.pad_value, rc_fill);
        ss_trace_rc_plan plan{};
        plan.ordered_step = RANGE_CHECK_STEP; plan.ordered_off = RC_ORDERED; plan.unused_off = RC_UNUSED;
        dt.rc_pool(plan, pool, rc_count, COL_RANGE_CHECK);
    }
    void rc_builtin(const ss_trace_rc_plan &plan, const RcPoolPlan &, const std::vector<uint64_t> &given3) { dt.rc_builtin(plan, given3, COL_RANGE_CHECK); }
    void memory() { dt.ordered_memory(COL_MEMORY, in.pi.public_memory, in.n / PUBLIC_MEMORY_STEP, in.pad_value, NPC_UNUSED_ADDR); }
};

// ExecutionTrace::new (layouts/src/recursive/trace.rs:89-688) section by section, for either backend: where a builtin's cells go is said
// ONCE (the `place` lambdas); the host stores them per block, the device per distinct instance
template <class Backend> void generate(Backend &be, const Inputs &in) {
    const uint64_t num_cycles = in.num_cycles, n = in.n;
    const AirPublicInput &pi = in.pi;
    const PrivateInput &priv = in.priv;
    const Segment &ped_seg = pi.segments[3], &rc_seg = pi.segments[4], &bw_seg = pi.segments[6];
    // ---- bitwise builtin and the diluted check (trace.rs:420-588), FIRST: the two diluted-check columns need no CPU cell, and a host-made
    // trace's uploads (ssh_prove_files: one copy stream, 9.5 ms a column) cannot begin before a column is final.  The instances' five
    // memory cells wait for the CPU section, which writes the memory pool's rows whole (below)
    const uint64_t bw_step = BITWISE_RATIO * CYCLE_HEIGHT;
    const auto bw_given = instances_by_index(priv.bitwise, n / bw_step, "bitwise");
    Instances<U256x2, BitwiseTrace> bitwise;
    bitwise.assign(n / bw_step, [&](uint64_t i) { auto it = bw_given.find((uint32_t)i); return it != bw_given.end() ? U256x2{it->second->x, it->second->y} : U256x2{}; });
    bitwise.trace_all([](const U256x2 &k) { return bitwise_instance_trace(k.first, k.second); });
    {
        const uint64_t shifted_cells[4] = {1, 65, 33, 97};
        be.builtin("bitwise cells", bitwise.of_block, (uint32_t)bitwise.keys.size(), bw_step, bw_seg.begin_addr, 5, COL_DILUTED_UNORDERED, [&](auto &s, uint32_t ti) {
            const BitwiseTrace &t = *bitwise.traces[ti];
            for (int k = 0; k < 4; ++k) s.cell(COL_DILUTED_UNORDERED, shifted_cells[k], t.shifted[k]);
            for (int p = 0; p < 4; ++p) for (int c = 0; c < 4; ++c) for (int sg = 0; sg < 4; ++sg) s.cell(COL_DILUTED_UNORDERED, 32 * p + 8 * c + 2 * sg, t.parts[p][c][sg]);
        });
        // the diluted pool: every instance's 68 diluted cells counted (by template: nearly every block holds the dummy instance)
        std::vector<uint32_t> dil_count(1u << DILUTED_N_BITS, 0), blocks_of(bitwise.keys.size(), 0);
        for (uint32_t t : bitwise.of_block) ++blocks_of[t];
        for (size_t t = 0; t < bitwise.keys.size(); ++t) for (uint32_t v : bitwise.traces[t]->undiluted) dil_count[v] += blocks_of[t];
        DilutedPlan dp;
        dp.from_counts(dil_count, n);
        dp.first[0] = 0;                // the rows before the first value are zeros, and so is the first value's image: one run from row 0
        std::vector<uint64_t> pad_rows, pad_values;
        size_t pi_ = 0;
        for (uint64_t blk = 0; blk < n / bw_step && pi_ < dp.padding.size(); ++blk)
            for (uint64_t off = 1; off < bw_step && pi_ < dp.padding.size(); off += 2) {
                if (off == 1 || off == 33 || off == 65 || off == 97) continue;
                pad_rows.push_back(blk * bw_step + off);
                pad_values.push_back(dilute(dp.padding[pi_++]));
            }
        if (pi_ < dp.padding.size()) fail("diluted-check values do not fit the trace");
        be.diluted(dp, pad_rows, pad_values);
    }
    // ---- the range-check pool: the offsets of every instruction and the builtin's parts counted first (trace.rs:131-160; utils.rs:357-380)
    std::vector<uint32_t> rc_count(1 << 16, 0);
    {
        std::vector<std::vector<uint32_t>> rc_count_of((size_t)omp_get_max_threads());
        parallel_for(num_cycles, in.par, [&](uint64_t cyc) {
            std::vector<uint32_t> &my_count = rc_count_of[(size_t)omp_get_thread_num()];
            if (my_count.empty()) my_count.assign(1 << 16, 0);
            const Word w{in.mem.at(in.states[cyc].pc)[0]};
            for (uint64_t v : {w.off_dst(), w.off_op0(), w.off_op1()}) ++my_count[v];
        });
        for (auto &part : rc_count_of) for (size_t v = 0; v < part.size(); ++v) rc_count[v] += part[v];
    }
    ss_trace_rc_plan rc_plan{};
    rc_plan.n_slots = num_cycles / RANGE_CHECK_BUILTIN_RATIO; rc_plan.n_given = priv.range_check.size(); rc_plan.slot_rows = RANGE_CHECK_BUILTIN_RATIO * CYCLE_HEIGHT;
    rc_plan.addr_begin = rc_seg.begin_addr; rc_plan.part_stride = CYCLE_HEIGHT; rc_plan.part_off = RC_UNUSED; rc_plan.pair_off = NPC_RANGE_CHECK128_ADDR;
    rc_plan.ordered_step = RANGE_CHECK_STEP; rc_plan.ordered_off = RC_ORDERED; rc_plan.unused_off = RC_UNUSED;
    if (rc_plan.n_given > rc_plan.n_slots) fail("more range-check instances than the trace has slots for");
    std::vector<uint64_t> rc_given;                       // index, value low, value high
    for (auto &inst : priv.range_check) {
        if (inst.value[2] | inst.value[3]) fail("range-check value does not fit 128 bits");
        rc_given.insert(rc_given.end(), {(uint64_t)inst.index, inst.value[0], inst.value[1]});
        for (unsigned k = 0; k < RANGE_CHECK_BUILTIN_PARTS; ++k) ++rc_count[(uint32_t)(shr(inst.value, 16 * (RANGE_CHECK_BUILTIN_PARTS - 1 - k))[0] & 0xffff)];
    }
    RcPoolPlan pool;
    pool.from_counts(rc_count, rc_plan.n_slots - rc_plan.n_given);          // the dummy instances take the first padding values (trace.rs:236-284)
    // the odd cycles take the next padding values in order, every cycle its CYCLE_HEIGHT / RANGE_CHECK_STEP ordered values
    if (pool.pad0 + num_cycles / 2 < pool.padding.size() || num_cycles * (CYCLE_HEIGHT / RANGE_CHECK_STEP) < pool.total) fail("range-check values do not fit the trace");
    rc_plan.rc_lo = pool.lo; rc_plan.rc_hi = pool.hi; rc_plan.n_padding = pool.padding.size(); rc_plan.pad0 = pool.pad0;
    be.cpu_section(pool, rc_count, pi.rc_max);
    // the bitwise instances' memory cells: x, y, x & y, x ^ y in the pool cells of the instance's four quarters, x | y beside
    be.builtin("bitwise memory cells", bitwise.of_block, (uint32_t)bitwise.keys.size(), bw_step, bw_seg.begin_addr, 5, -1, [&](auto &s, uint32_t ti) {
        const BitwiseTrace &t = *bitwise.traces[ti];
        for (int k = 0; k < 4; ++k) s.pair(NPC_BITWISE_POOL_ADDR + k * (bw_step / 4), k, t.memory[k]);
        s.pair(NPC_BITWISE_X_OR_Y_ADDR, 4, t.memory[4]);
    });
    // ---- Pedersen builtin (trace.rs:300-400; builtins/src/pedersen/mod.rs:81-163)
    {
        const uint64_t step = PEDERSEN_BUILTIN_RATIO * CYCLE_HEIGHT;
        const auto given = instances_by_index(priv.pedersen, n / step, "pedersen");
        Instances<U256x2, PedersenTrace> inst;
        inst.assign(n / step, [&](uint64_t i) { auto it = given.find((uint32_t)i); return it != given.end() ? U256x2{it->second->a, it->second->b} : U256x2{}; });
        inst.trace_all([](const U256x2 &k) { return pedersen_instance_trace_cached(k.first, k.second); });
        be.builtin("pedersen", inst.of_block, (uint32_t)inst.keys.size(), step, ped_seg.begin_addr, 3, -1, [&](auto &s, uint32_t t) {
            const PedersenTrace &c = *inst.traces[t];
            for (uint64_t j = 0; j < 512; ++j) {
                const uint64_t r = 4 * j;
                s.cell(COL_RANGE_CHECK, r + 1, c.steps[j].point.x); s.cell(COL_RANGE_CHECK, r + 3, c.steps[j].point.y);
                s.cell(COL_AUXILIARY, r, c.steps[j].suffix); s.cell(COL_AUXILIARY, r + 2, c.steps[j].slope);
            }
            const U256 *in2[2] = {&inst.keys[t].first, &inst.keys[t].second};
            for (int half = 0; half < 2; ++half) {
                const bool b251 = bit(*in2[half], 251), b196 = bit(*in2[half], 196), b192 = bit(*in2[half], 192);
                s.cell(COL_AUXILIARY, 1024 * half + 1022, felt_from_u64(b251 && b196));
                s.cell(COL_AUXILIARY, 1024 * half + 7, felt_from_u64(b251 && b196 && b192));
            }
            s.pair(NPC_PEDERSEN_INPUT0_ADDR, 0, felt_from_canonical(inst.keys[t].first));
            s.pair(NPC_PEDERSEN_INPUT1_ADDR, 1, felt_from_canonical(inst.keys[t].second));
            s.pair(NPC_PEDERSEN_OUTPUT_ADDR, 2, c.out);
        });
    }
    // ---- range-check builtin cells
    be.rc_builtin(rc_plan, pool, rc_given);
    // ---- gap fillers (trace.rs:594-625), sorted memory (get_ordered_memory_accesses, utils.rs:112-152)
    be.memory();
}

Inputs check_inputs(const RegisterStates &states, const Mem &mem, const AirPublicInput &pi, const PrivateInput &priv) {
    const uint64_t num_cycles = states.size();
    if (!num_cycles || (num_cycles & (num_cycles - 1))) fail("the number of cycles must be a power of two");
    const MemoryEntry *padding = nullptr;
    for (auto &e : pi.public_memory) if (e.address == 1) { padding = &e; break; }
    if (!padding) fail("public memory has no entry at address 1");
    if (!pi.segments[3].present || !pi.segments[4].present || !pi.segments[6].present) fail("the layout needs the pedersen, range_check and bitwise segments");
    return Inputs{states, mem, pi, priv, num_cycles, num_cycles * CYCLE_HEIGHT, felt_from_canonical(padding->value), num_cycles >= parallel_min_cycles()};
}

}  // namespace

// the same into the caller's columns; every cell is written.  No column is zeroed first: the section that owns a column writes its zeros
// with its values (the auxiliary column in the CPU's pass, the two diluted-check columns in the bitwise section): one pass per column
void recursive_base_trace_into(Felt *const out[7], const RegisterStates &states, const std::vector<U256> &memory,
                               const std::vector<uint8_t> &present, const AirPublicInput &pi, const PrivateInput &priv,
                               const std::function<void(int)> *column_done) {
    const HostThreadsScope host_threads_scope;              // OpenMP threads by the cgroup's CPU quota (trace_common.hpp)
    const Mem mem{memory, present};
    const Inputs in = check_inputs(states, mem, pi, priv);
    HostBackend be(in, out, column_done);
    be.mark("init");
    generate(be, in);
}

// the same columns made in HBM (d_cols: 7 device columns of 16 * cycles felts): trace.bin / memory.bin go up as they are, the cells are
// made by csrc/trace.hip; only what the host needs for the plans is read here (the instructions' offsets for the range-check pool)
void recursive_base_trace_device(ss_ctx *ctx, uint64_t *const d_cols[7], const uint8_t *trace_bin, uint64_t trace_len, const uint8_t *memory_bin, uint64_t memory_len,
                                 const std::vector<U256> &memory, const std::vector<uint8_t> &present, const AirPublicInput &pi, const PrivateInput &priv) {
    const HostThreadsScope host_threads_scope;
    const RegisterStates states(trace_bin, trace_len);
    const Mem mem{memory, present};
    const Inputs in = check_inputs(states, mem, pi, priv);
    DeviceTrace dt(ctx, in.num_cycles, d_cols, NUM_COLS);
    dt.lap("inputs checked");
    dt.load_inputs(trace_bin, trace_len, memory_bin, memory_len);
    DeviceBackend be{in, dt};
    generate(be, in);
    dt.finish();
}

}  // namespace ssh
