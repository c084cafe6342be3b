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

// the same into the caller's columns (pinned host memory the upload reads straight from: the GpuAllocator seam of
// layouts/src/recursive/trace.rs:115-120); every cell is written
void recursive_base_trace_into(Felt *const out[7], const RegisterStates &states, const std::vector<U256> &memory,
                               const std::vector<uint8_t> &present, const AirPublicInput &pi, const PrivateInput &priv,
                               const std::function<void(int)> *column_done) {
    auto done = [&](std::initializer_list<int> cs) { if (column_done && *column_done) for (int c : cs) (*column_done)(c); };
    const HostThreadsScope host_threads_scope;              // OpenMP threads by the cgroup's CPU quota (trace_common.hpp)
    const bool timing = getenv("SSH_TRACE_TIMING") != nullptr;   // diagnostic: per-section wall time on stderr
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!timing) return;
        auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[trace timing] %-22s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    const uint64_t num_cycles = states.size();
    if (!num_cycles || (num_cycles & (num_cycles - 1))) fail("the number of cycles must be a power of two");
    const uint64_t n = num_cycles * CYCLE_HEIGHT;
    const Mem mem{memory, present};
    const Felt zero = felt_from_u64(0);
    const bool par = num_cycles >= parallel_min_cycles();
    struct Col { Felt *p; Felt &operator[](uint64_t i) const { return p[i]; } };
    Col cols[NUM_COLS];
    for (int c = 0; c < NUM_COLS; ++c) cols[c] = Col{out[c]};
    // every cell no section writes is the field's zero; no column is zeroed first - the section that owns a column writes its zeros
    // with its values (the auxiliary column in the CPU's pass, the two diluted-check columns in the bitwise section): one pass per column
    const Col flags = cols[COL_FLAGS], un_col = cols[COL_DILUTED_UNORDERED], od_col = cols[COL_DILUTED_ORDERED], npc = cols[COL_NPC],
              rc_col = cols[COL_RANGE_CHECK], aux = cols[COL_AUXILIARY];
    AddrArray npc_addr(n / 2);                          // the address half of the pool, as integers (sorting, gap search); assigned by its cycle below

    const MemoryEntry *padding = nullptr;
    for (auto &e : pi.public_memory) if (e.address == 1) { padding = &e; break; }
    if (!padding) fail("public memory has no entry at address 1");
    const Felt pad_value = felt_from_canonical(padding->value), pad_addr = felt_from_u64(1);
    auto set_pair = [&](uint64_t row, uint64_t address, const Felt &value) { npc[row] = felt_from_u64(address); npc[row + 1] = value; npc_addr[row / 2] = address; };

    mark("init");
    const Segment &ped_seg = pi.segments[3], &rc_seg = pi.segments[4], &bw_seg = pi.segments[6];
    if (!ped_seg.present || !rc_seg.present || !bw_seg.present) fail("the layout needs the pedersen, range_check and bitwise segments");
    // ---- bitwise builtin and the diluted check (trace.rs:420-588), FIRST: the two diluted-check columns need no CPU cell, and the call
    // this generator runs inside (ssh_prove_files) is bound by the uploads of the finished columns - one copy stream, 9.5 ms a column -,
    // which cannot begin before a column is final.  The instances' five memory cells wait for the CPU section, which writes the
    // memory pool's rows whole (below)
    {
        const uint64_t step = BITWISE_RATIO * CYCLE_HEIGHT;
        std::map<uint32_t, const BitwiseInstance *> given;
        for (auto &inst : priv.bitwise) given[inst.index] = &inst;
        std::vector<uint32_t> dil_count(1u << DILUTED_N_BITS, 0);
        // one histogram per thread (nearly every instance is the dummy one: all threads would hammer the counter of value 0)
        std::vector<std::vector<uint32_t>> dil_count_of((size_t)omp_get_max_threads());
        const uint64_t shifted_cells[4] = {1, 65, 33, 97};
        std::string bw_error;
#pragma omp parallel for schedule(static) if (par)
        for (int64_t bi = 0; bi < (int64_t)(n / step); ++bi) try {
            std::vector<uint32_t> &my_count = dil_count_of[(size_t)omp_get_thread_num()];
            if (my_count.empty()) my_count.assign(1u << DILUTED_N_BITS, 0);
            const uint64_t i = (uint64_t)bi;
            const uint64_t base = i * step;
            U256 x{}, y{};
            auto it = given.find((uint32_t)i);
            if (it != given.end()) { x = it->second->x; y = it->second->y; }
            else {
                // the dummy instance (x = y = 0; nearly every instance of a run is one): each of its 4 + 64 diluted cells is the value 0 -
                // nothing to partition, dilute or check (that arithmetic, not the stores, was most of this section's time)
                for (uint64_t o = 0; o < step; ++o) un_col[base + o] = zero;
                my_count[0] += 4 + 64;
                continue;
            }
            U256 vand, vxor;
            for (int k = 0; k < 4; ++k) { vand[k] = x[k] & y[k]; vxor[k] = x[k] ^ y[k]; }
            const U256 *vals[4] = {&x, &y, &vand, &vxor};
            uint64_t parts[4][4][4];
            for (int p = 0; p < 4; ++p) for (int c = 0; c < 4; ++c) partition64((*vals[p])[c], parts[p][c]);
            Felt blk[BITWISE_RATIO * CYCLE_HEIGHT];                  // the instance's rows of the unordered column: zeros, then its cells
            for (uint64_t o = 0; o < step; ++o) blk[o] = zero;
            for (int k = 0; k < 4; ++k) {
                const uint64_t v = parts[2][3][k] + parts[3][3][k];
                const unsigned sh = k == 3 ? 8 : 4;
                if ((v << sh) >> sh != v) fail("bitwise instance: top segment does not fit");
                blk[shifted_cells[k]] = felt_from_u64(v << sh);
                ++my_count[undilute(v << sh)];
            }
            for (int p = 0; p < 4; ++p)
                for (int c = 0; c < 4; ++c)
                    for (int s = 0; s < 4; ++s) {
                        blk[32 * p + 8 * c + 2 * s] = felt_from_u64(parts[p][c][s]);
                        ++my_count[undilute(parts[p][c][s])];
                    }
            for (uint64_t o = 0; o < step; ++o) un_col[base + o] = blk[o];
        } catch (const std::exception &e) {
#pragma omp critical
            if (bw_error.empty()) bw_error = e.what();
        }
        if (!bw_error.empty()) throw std::runtime_error(bw_error);
        for (auto &part : dil_count_of) for (size_t v = 0; v < part.size(); ++v) dil_count[v] += part[v];
        std::vector<uint32_t> padding;
        uint64_t total = 0;
        for (uint32_t v = 0; v < (1u << DILUTED_N_BITS); ++v) { if (!dil_count[v]) padding.push_back(v); total += std::max(dil_count[v], 1u); }
        if (total > n) fail("diluted-check values do not fit the trace");
        size_t pi_ = 0;
        for (uint64_t blk = 0; blk < n / step && pi_ < padding.size(); ++blk)
            for (uint64_t off = 1; off < step && pi_ < padding.size(); off += 2) {
                if (off == 1 || off == 33 || off == 65 || off == 97) continue;
                un_col[blk * step + off] = felt_from_u64(dilute(padding[pi_++]));
            }
        if (pi_ < padding.size()) fail("diluted-check values do not fit the trace");
        std::vector<uint64_t> first_row((1u << DILUTED_N_BITS) + 1, n - total);
        for (uint32_t v = 0; v < (1u << DILUTED_N_BITS); ++v) first_row[v + 1] = first_row[v] + std::max(dil_count[v], 1u);
        first_row[0] = 0;               // the rows before the first value are zeros, and so is the first value's image: one run from row 0
#pragma omp parallel for schedule(dynamic, 64) if (par)
        for (int64_t v = 0; v < (int64_t)(1u << DILUTED_N_BITS); ++v) {
            const Felt f = felt_from_u64(dilute((uint32_t)v));
            if (first_row[v + 1] - first_row[v] < 4096) { for (uint64_t r2 = first_row[v]; r2 < first_row[v + 1]; ++r2) od_col[r2] = f; }
        }
        for (uint32_t v = 0; v < (1u << DILUTED_N_BITS); ++v)             // the long runs (value 0 of the dummy instances) by all threads
            if (first_row[v + 1] - first_row[v] >= 4096) {
                const Felt f = felt_from_u64(dilute(v));
#pragma omp parallel for schedule(static) if (par)
                for (int64_t r2 = (int64_t)first_row[v]; r2 < (int64_t)first_row[v + 1]; ++r2) od_col[r2] = f;
            }
    }
    mark("bitwise + diluted");
    done({COL_DILUTED_UNORDERED, COL_DILUTED_ORDERED});
    // ---- the range-check pool: the offsets of every instruction counted first, no column touched (trace.rs:131-160; utils.rs:357-380)
    std::vector<uint32_t> rc_count(1 << 16, 0);
    std::string first_error;                            // exceptions must not leave an OpenMP region
    // one histogram of the offsets per thread (an idling run has the same three offsets in every cycle: all threads would hammer
    // three counters)
    std::vector<std::vector<uint32_t>> rc_count_of((size_t)omp_get_max_threads());
#pragma omp parallel for schedule(static) if (par)
    for (int64_t cyc = 0; cyc < (int64_t)num_cycles; ++cyc) try {
        std::vector<uint32_t> &my_count = rc_count_of[(size_t)omp_get_thread_num()];
        if (my_count.empty()) my_count.assign(1 << 16, 0);
        const Word w{mem.at(states[(uint64_t)cyc].pc)[0]};
        for (uint64_t v : {w.off_dst(), w.off_op0(), w.off_op1()}) ++my_count[v];
    } catch (const std::exception &e) {
#pragma omp critical
        if (first_error.empty()) first_error = e.what();
    }
    if (!first_error.empty()) throw std::runtime_error(first_error);
    for (auto &part : rc_count_of) for (size_t v = 0; v < part.size(); ++v) rc_count[v] += part[v];
    // range-check builtin instances, ordered values and padding (trace.rs:236-284)
    struct Rc128 { uint32_t index; U256 value; };
    std::vector<Rc128> rc128;
    auto part_of = [](const U256 &v, unsigned k) { return (uint32_t)(shr(v, 16 * (RANGE_CHECK_BUILTIN_PARTS - 1 - k))[0] & 0xffff); };
    for (auto &inst : priv.range_check) {
        if (inst.value[2] | inst.value[3]) fail("range-check value does not fit 128 bits");
        rc128.push_back(Rc128{inst.index, inst.value});
        for (unsigned k = 0; k < RANGE_CHECK_BUILTIN_PARTS; ++k) ++rc_count[part_of(inst.value, k)];
    }
    uint32_t rc_lo = 0xffff, rc_hi = 0;
    for (uint32_t v = 0; v < (1u << 16); ++v) if (rc_count[v]) { rc_lo = std::min(rc_lo, v); rc_hi = std::max(rc_hi, v); }
    std::vector<uint32_t> padding_vals;
    for (uint32_t v = rc_lo; v <= rc_hi; ++v) if (!rc_count[v]) padding_vals.push_back(v);
    CountArray ordered_vals;
    ordered_runs(rc_count, rc_lo, rc_hi, ordered_vals);
    size_t pad_i = 0;
    auto next_padding = [&]() { return pad_i < padding_vals.size() ? padding_vals[pad_i++] : rc_hi; };
    for (uint64_t index = rc128.size(); index < num_cycles / RANGE_CHECK_BUILTIN_RATIO; ++index) {
        U256 value{};
        for (unsigned k = 0; k < RANGE_CHECK_BUILTIN_PARTS; ++k) {      // value = (value << 16) + part
            value[3] = (value[3] << 16) | (value[2] >> 48); value[2] = (value[2] << 16) | (value[1] >> 48);
            value[1] = (value[1] << 16) | (value[0] >> 48); value[0] = (value[0] << 16) | next_padding();
        }
        rc128.push_back(Rc128{(uint32_t)index, value});
    }
    // the odd cycles take the next padding values in order, every cycle its CYCLE_HEIGHT / RANGE_CHECK_STEP ordered values: both
    // sequences are indexed by the cycle, so the cycles go in parallel
    const size_t pad0 = pad_i;
    const uint64_t per = CYCLE_HEIGHT / RANGE_CHECK_STEP;
    if (pad0 + num_cycles / 2 < padding_vals.size() || num_cycles * per < ordered_vals.size()) fail("range-check values do not fit the trace");
    const Felt rc_max_f = felt_from_u64(pi.rc_max);

    mark("range check");
    // ---- CPU cells (trace.rs:172-232).  The generator is bound by the host's memory traffic, so the four columns this section fills are
    // written ONCE: a cycle's 16 rows of each are made in a block on the stack - padding first, then what the cycle puts there, in the
    // order the separate passes of the first version wrote them - and stored row after row; a builtin's cells of these columns get the
    // padding here and their values in the builtin's section
    constexpr uint64_t JNZ_BLOCK = 512;                  // cycles whose conditional jumps share one inversion (JnzInverses)
#pragma omp parallel for schedule(static) if (par)
    for (int64_t blk_i = 0; blk_i < (int64_t)((num_cycles + JNZ_BLOCK - 1) / JNZ_BLOCK); ++blk_i) try {
        const uint64_t first_cycle = (uint64_t)blk_i * JNZ_BLOCK, end_cycle = std::min(num_cycles, first_cycle + JNZ_BLOCK);
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
            if (cycle % 2 == 1) {
                const size_t at = pad0 + cycle / 2;
                blk[RC_UNUSED] = felt_from_u64(at < padding_vals.size() ? padding_vals[at] : rc_hi);
            }
            for (uint64_t o = 0; o < CYCLE_HEIGHT; o += RANGE_CHECK_STEP) {
                const uint64_t at = cycle * per + o / RANGE_CHECK_STEP;
                blk[o + RC_ORDERED] = felt_from_u64(at < ordered_vals.size() ? ordered_vals[at] : rc_hi);
            }
            for (uint64_t o = 0; o < CYCLE_HEIGHT; ++o) rc_col[r + o] = blk[o];
            // auxiliary column: zero where no section writes
            for (uint64_t o = 0; o < CYCLE_HEIGHT; ++o) blk[o] = zero;
            blk[AUX_TMP0] = tmp0; blk[AUX_TMP1] = felt_mul(tmp0, res);
            blk[AUX_AP] = felt_from_u64(ap); blk[AUX_FP] = felt_from_u64(fp);
            blk[AUX_OP0_MUL_OP1] = felt_mul(op0, op1); blk[AUX_RES] = res;
            for (uint64_t o = 0; o < CYCLE_HEIGHT; ++o) aux[r + o] = blk[o];
        }
    } catch (const std::exception &e) {
#pragma omp critical
        if (first_error.empty()) first_error = e.what();
    }
    if (!first_error.empty()) throw std::runtime_error(first_error);

    mark("cpu cells");
    done({COL_FLAGS});
    {   // the bitwise instances' memory cells: x, y, x & y, x ^ y in the pool cells of the instance's four quarters, x | y beside
        const uint64_t step = BITWISE_RATIO * CYCLE_HEIGHT;
        std::map<uint32_t, const BitwiseInstance *> given;
        for (auto &inst : priv.bitwise) given[inst.index] = &inst;
#pragma omp parallel for schedule(static) if (par)
        for (int64_t bi = 0; bi < (int64_t)(n / step); ++bi) {
            const uint64_t i = (uint64_t)bi, base = i * step, addr = bw_seg.begin_addr + 5 * i;
            U256 x{}, y{};
            auto it = given.find((uint32_t)i);
            if (it != given.end()) { x = it->second->x; y = it->second->y; }
            U256 vand, vxor, vor;
            for (int k = 0; k < 4; ++k) { vand[k] = x[k] & y[k]; vxor[k] = x[k] ^ y[k]; vor[k] = x[k] | y[k]; }
            const U256 *vals[4] = {&x, &y, &vand, &vxor};
            for (int k = 0; k < 4; ++k) set_pair(base + NPC_BITWISE_POOL_ADDR + k * (step / 4), addr + k, felt_from_canonical(*vals[k]));
            set_pair(base + NPC_BITWISE_X_OR_Y_ADDR, addr + 4, felt_from_canonical(vor));
        }
    }
    mark("bitwise memory cells");

    // ---- Pedersen builtin (trace.rs:300-400; builtins/src/pedersen/mod.rs:81-163)
    {
        const uint64_t step = PEDERSEN_BUILTIN_RATIO * CYCLE_HEIGHT;
        std::map<uint32_t, const PedersenInstance *> given;
        for (auto &inst : priv.pedersen) given[inst.index] = &inst;
        struct Cached { std::vector<Step> steps; Felt out; };
        std::map<std::pair<U256, U256>, Cached> cache;
        const Pt p0 = pedersen_point(0);
        // the DISTINCT instances are found first (sequential: the map is shared; nearly every instance is the dummy one), their traces
        // are made by all threads (a real instance is 512 curve steps: a run may hold thousands of them), then the cells in parallel
        std::vector<const Cached *> of_block(n / step);
        std::vector<std::pair<const std::pair<U256, U256> *, Cached *>> distinct;
        for (uint64_t i = 0; i < n / step; ++i) {
            auto it = given.find((uint32_t)i);
            const auto ins = cache.emplace(it != given.end() ? std::make_pair(it->second->a, it->second->b) : std::make_pair(U256{}, U256{}), Cached{});
            if (ins.second) distinct.emplace_back(&ins.first->first, &ins.first->second);
            of_block[i] = &ins.first->second;
        }
        parallel_items(distinct.size(), [&](uint64_t k) {
            const U256 &a = distinct[k].first->first, &b = distinct[k].first->second;
            Cached &c = *distinct[k].second;
            c.steps.reserve(512);
            const Pt mid = element_steps(a, p0, 0, c.steps);
            element_steps(b, mid, 1, c.steps);
            c.out = c.steps.back().point.x;
            Felt want;
            const Felt fa = felt_from_canonical(a), fb = felt_from_canonical(b);
            if (ss_pedersen_hash_host(fa.data(), fb.data(), want.data()) != SS_OK || !felt_eq(want, c.out))
                fail("Pedersen partial sums do not end at the hash");                     // the reference's own assert
        });
#pragma omp parallel for schedule(static) if (par)
        for (int64_t bi = 0; bi < (int64_t)(n / step); ++bi) {
            const uint64_t i = (uint64_t)bi;
            U256 a{}, b{};
            auto it = given.find((uint32_t)i);
            if (it != given.end()) { a = it->second->a; b = it->second->b; }
            const Cached &c = *of_block[i];
            const uint64_t base = i * step, addr = ped_seg.begin_addr + 3 * i;
            for (uint64_t j = 0; j < 512; ++j) {
                const uint64_t r = base + 4 * j;
                rc_col[r + 1] = c.steps[j].point.x; rc_col[r + 3] = c.steps[j].point.y;
                aux[r] = c.steps[j].suffix; aux[r + 2] = c.steps[j].slope;
            }
            const U256 *in[2] = {&a, &b};
            for (int half = 0; half < 2; ++half) {
                const bool b251 = bit(*in[half], 251), b196 = bit(*in[half], 196), b192 = bit(*in[half], 192);
                aux[base + 1024 * half + 1022] = felt_from_u64(b251 && b196);
                aux[base + 1024 * half + 7] = felt_from_u64(b251 && b196 && b192);
            }
            set_pair(base + NPC_PEDERSEN_INPUT0_ADDR, addr, felt_from_canonical(a));
            set_pair(base + NPC_PEDERSEN_INPUT1_ADDR, addr + 1, felt_from_canonical(b));
            set_pair(base + NPC_PEDERSEN_OUTPUT_ADDR, addr + 2, c.out);
        }
    }
    mark("pedersen");
    done({COL_AUXILIARY});
    // ---- range-check builtin cells
    {
        const uint64_t step = RANGE_CHECK_BUILTIN_RATIO * CYCLE_HEIGHT;
#pragma omp parallel for schedule(static) if (par)
        for (int64_t blk = 0; blk < (int64_t)rc128.size(); ++blk) {
            const uint64_t block = (uint64_t)blk, base = block * step;
            for (unsigned k = 0; k < RANGE_CHECK_BUILTIN_PARTS; ++k) rc_col[base + CYCLE_HEIGHT * k + RC_UNUSED] = felt_from_u64(part_of(rc128[block].value, k));
            set_pair(base + NPC_RANGE_CHECK128_ADDR, rc_seg.begin_addr + rc128[block].index, felt_from_canonical(rc128[block].value));
        }
    }
    mark("rc builtin");
    done({COL_RANGE_CHECK});
    // ---- gap fillers (trace.rs:594-625)
    {
        const std::vector<uint64_t> gaps = memory_gaps(npc_addr, pi.public_memory, num_cycles);
        if (gaps.size() > num_cycles) fail("more memory gaps than cycles to hold them");
        for (size_t k = 0; k < gaps.size(); ++k) set_pair(k * CYCLE_HEIGHT + NPC_UNUSED_ADDR, gaps[k], zero);
    }
    mark("gap fill");
    done({COL_NPC});
    // ---- sorted memory (get_ordered_memory_accesses, utils.rs:112-152)
    ordered_memory_into(out[COL_MEMORY], n, npc_addr, out[COL_NPC], n / PUBLIC_MEMORY_STEP, pi.public_memory, pad_value);
    mark("sorted memory");
    done({COL_MEMORY});
}

}  // namespace ssh
