// device_trace.hpp — the host's side of the base trace made ON the device (csrc/trace.hip behind the ss_trace_* entry points of
// include/sandstorm_hip.h).  ExecutionTrace::new (layouts/src/starknet/trace.rs:99-987,
// layouts/src/recursive/trace.rs:89-688) writes 3.6 / 4.8 GB of cells that are functions of ~25 MB of `cairo-run` output; the
// layouts' generators (trace_recursive.cpp, trace_starknet.cpp) drive either backend from ONE description of where a builtin's
// cells go:
//   * a section's per-instance cells are written through a Sink - HostSink stores into the host columns (the host generator),
//     TemplateSink records them as ONE template per DISTINCT instance, which the device copies into every block that holds it;
//   * DeviceTrace uploads the raw files, plans and templates and launches the kernels in the order the host sections run
//     (a later section overwrites an earlier one's cells, as on the host); the input's errors come back as status bits and are
//     thrown with the host generator's messages.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/sandstorm_hip.h"
#include "trace_common.hpp"

namespace ssh {
namespace tracedetail {

// ---- where a section's cells go
// base: the instance's first row; addr0: its first memory address (begin + cells_per_instance * i)
struct HostSink {
    Felt *const *cols;
    AddrArray *npc_addr;
    int npc_col;
    uint64_t base, addr0;
    void cell(int col, uint64_t off, const Felt &v) const { cols[col][base + off] = v; }
    void pair(uint64_t off, uint64_t j, const Felt &v) const {                       // (address, value) at rows off, off + 1 of the memory pool
        cols[npc_col][base + off] = felt_from_u64(addr0 + j);
        cols[npc_col][base + off + 1] = v;
        (*npc_addr)[(base + off) / 2] = addr0 + j;
    }
};
// the same calls recorded: the cells of ONE instance as (column, row offset) -> value, the last write of a cell wins (as in memory)
struct TemplateSink {
    struct Rec { uint64_t key; ss_trace_cell cell; Felt value; uint32_t seq; };
    std::vector<Rec> recs;
    int npc_col;
    void cell(int col, uint64_t off, const Felt &v) { recs.push_back(Rec{((uint64_t)col << 32) | off, ss_trace_cell{(uint32_t)col, (uint32_t)off, SS_TRACE_CELL_VALUE, 0}, v, (uint32_t)recs.size()}); }
    void pair(uint64_t off, uint64_t j, const Felt &v) {
        recs.push_back(Rec{((uint64_t)npc_col << 32) | off, ss_trace_cell{(uint32_t)npc_col, (uint32_t)off, SS_TRACE_CELL_ADDRESS, (uint32_t)j}, Felt{}, (uint32_t)recs.size()});
        cell(npc_col, off + 1, v);
    }
    // sorted by (column, offset) - consecutive entries are consecutive rows of a column: coalesced stores - with the LAST write of every cell
    void finish(std::vector<ss_trace_cell> &cells, std::vector<Felt> &values) {
        std::stable_sort(recs.begin(), recs.end(), [](const Rec &a, const Rec &b) { return a.key < b.key; });
        cells.clear();
        for (size_t k = 0; k < recs.size(); ++k) {
            if (k + 1 < recs.size() && recs[k + 1].key == recs[k].key) continue;
            cells.push_back(recs[k].cell);
            values.push_back(recs[k].value);
        }
    }
};

// A builtin's instances by template: the DISTINCT instances of a run (nearly always one: the dummy instance) and which one every
// block holds.  Key: the instance's inputs; Trace: what its cells are made from.
template <class Key, class Trace> struct Instances {
    std::vector<Key> keys;
    std::vector<std::shared_ptr<const Trace>> traces;
    std::vector<uint32_t> of_block;
    // key_of(i) for every block, sequentially (the map is shared; nearly every block is the dummy instance)
    template <class KeyOf> void assign(uint64_t nblocks, const KeyOf &key_of) {
        std::map<Key, uint32_t> index;
        of_block.resize(nblocks);
        uint32_t last = 0;
        for (uint64_t i = 0; i < nblocks; ++i) {
            const Key k = key_of(i);
            if (i && k == keys[last]) { of_block[i] = last; continue; }       // (runs of one instance - the dummy's - cost a comparison, not a map lookup)
            auto ins = index.emplace(k, (uint32_t)keys.size());
            if (ins.second) keys.push_back(k);
            of_block[i] = last = ins.first->second;
        }
        traces.resize(keys.size());
    }
    // the distinct instances' traces by all threads (a real Pedersen instance is 512 curve steps; a run may hold tens of thousands)
    template <class Make> void trace_all(const Make &make) {
        parallel_items(keys.size(), [&](uint64_t k) { traces[k] = make(keys[k]); });
    }
    bool uniform() const { return keys.size() == 1; }
};

// instance indices of a private input: below the builtin's slot count and unique (air-private-input.json is the caller's: an index
// out of range or given twice would write outside the columns / leave a block without its trace)
template <class Inst> std::map<uint32_t, const Inst *> instances_by_index(const std::vector<Inst> &v, uint64_t slots, const char *what) {
    std::map<uint32_t, const Inst *> given;
    for (auto &inst : v) {
        if (inst.index >= slots) fail(std::string(what) + " instance " + std::to_string(inst.index) + " beyond the trace's " + std::to_string(slots) + " slots");
        if (!given.emplace(inst.index, &inst).second) fail(std::string(what) + " instance " + std::to_string(inst.index) + " given twice");
    }
    return given;
}

// ---- what both layouts' builtin sections share
// a Pedersen instance's 512 curve steps and its hash (builtins/src/pedersen/mod.rs:81-163), with the reference's own assert
struct PedersenTrace { std::vector<Step> steps; Felt out; };
inline std::shared_ptr<const PedersenTrace> pedersen_instance_trace(const U256 &a, const U256 &b);
// the dummy instance (a = b = 0: what nearly every block of a run holds) is traced once per process
inline std::shared_ptr<const PedersenTrace> pedersen_instance_trace_cached(const U256 &a, const U256 &b) {
    if ((a[0] | a[1] | a[2] | a[3] | b[0] | b[1] | b[2] | b[3]) == 0) {
        static const std::shared_ptr<const PedersenTrace> dummy = pedersen_instance_trace(U256{}, U256{});
        return dummy;
    }
    return pedersen_instance_trace(a, b);
}
inline std::shared_ptr<const PedersenTrace> pedersen_instance_trace(const U256 &a, const U256 &b) {
    auto c = std::make_shared<PedersenTrace>();
    c->steps.reserve(512);
    const Pt mid = element_steps(a, pedersen_point(0), 0, c->steps);
    element_steps(b, mid, 1, c->steps);
    c->out = c->steps.back().point.x;
    Felt want;
    const Felt fa = felt_from_canonical(a), fb = felt_from_canonical(b);
    if (ss_pedersen_hash_host(fa.data(), fb.data(), want.data()) != SS_OK || !felt_eq(want, c->out)) fail("Pedersen partial sums do not end at the hash");
    return c;
}
// a bitwise instance's cells (builtins/src/bitwise/mod.rs; trace.rs:525-667): the four values' 64 diluted parts, the four shifted
// top segments of x & y + x ^ y, the five memory values; `undiluted`: the 68 diluted cells as the pool counts them
struct BitwiseTrace { Felt parts[4][4][4], shifted[4], memory[5]; uint32_t undiluted[68]; };
inline std::shared_ptr<const BitwiseTrace> bitwise_instance_trace(const U256 &x, const U256 &y) {
    auto t = std::make_shared<BitwiseTrace>();
    U256 vand, vxor, vor;
    for (int k = 0; k < 4; ++k) { vand[k] = x[k] & y[k]; vxor[k] = x[k] ^ y[k]; vor[k] = x[k] | y[k]; }
    const U256 *vals[5] = {&x, &y, &vand, &vxor, &vor};
    uint64_t parts[4][4][4];
    for (int p = 0; p < 4; ++p) for (int c = 0; c < 4; ++c) partition64((*vals[p])[c], parts[p][c]);
    for (unsigned k = 0; k < 4; ++k) {
        const uint64_t v = parts[2][3][k] + parts[3][3][k];
        const unsigned sh = k == 3 ? 8 : 4;
        if (((v << sh) >> sh) != v) fail("bitwise instance: top segment does not fit");
        t->shifted[k] = felt_from_u64(v << sh);
        t->undiluted[k] = undilute(v << sh);
    }
    for (int p = 0; p < 4; ++p)
        for (int c = 0; c < 4; ++c)
            for (int sg = 0; sg < 4; ++sg) {
                t->parts[p][c][sg] = felt_from_u64(parts[p][c][sg]);
                t->undiluted[4 + 16 * p + 4 * c + sg] = undilute(parts[p][c][sg]);
            }
    for (int k = 0; k < 5; ++k) t->memory[k] = felt_from_canonical(*vals[k]);
    return t;
}
// the diluted pool (utils.rs DilutedCheckPool; trace.rs:668-705): from the histogram of the 16-bit values behind the diluted cells
struct DilutedPlan {
    std::vector<uint32_t> padding;           // the 16-bit values nothing uses, ascending: they go to free unordered cells
    uint64_t total = 0;                      // ordered slots the values take: max(count, 1) each
    std::vector<uint32_t> first;             // 65537: first[v] = the first ordered slot of value v, first[0] = slots - total
    void from_counts(const std::vector<uint32_t> &count, uint64_t slots) {
        padding.clear();
        total = 0;
        for (uint32_t v = 0; v < (1u << DILUTED_N_BITS); ++v) { if (!count[v]) padding.push_back(v); total += std::max(count[v], 1u); }
        if (total > slots) fail("diluted-check values do not fit the trace");
        first.assign((1u << DILUTED_N_BITS) + 1, (uint32_t)(slots - total));
        for (uint32_t v = 0; v < (1u << DILUTED_N_BITS); ++v) first[v + 1] = first[v] + std::max(count[v], 1u);
    }
};

// ---- the 16-bit range-check pool's plan (utils.rs:357-380): from the histogram of the pool's values
struct RcPoolPlan {
    uint32_t lo = 0xffff, hi = 0;
    std::vector<uint32_t> padding;           // the values of [lo, hi] nothing uses, ascending
    uint64_t total = 0;                      // ordered values: every value of [lo, hi] max(count, 1) times
    size_t pad0 = 0;                         // padding values the dummy instances of the builtin take
    uint32_t pad(size_t j) const { return j < padding.size() ? padding[j] : hi; }
    void from_counts(const std::vector<uint32_t> &count, uint64_t dummy_slots) {
        lo = 0xffff; hi = 0;
        for (uint32_t v = 0; v < (1u << 16); ++v) if (count[v]) { lo = std::min(lo, v); hi = std::max(hi, v); }
        padding.clear();
        total = 0;
        for (uint32_t v = lo; v <= hi; ++v) { if (!count[v]) padding.push_back(v); total += std::max(count[v], 1u); }
        pad0 = std::min<uint64_t>(8 * dummy_slots, padding.size());
    }
    // first[j] = ordered values before value lo + j, j <= hi - lo + 1
    std::vector<uint32_t> first(const std::vector<uint32_t> &count) const {
        std::vector<uint32_t> f((size_t)(hi - lo) + 2, 0);
        for (uint32_t v = lo; v <= hi; ++v) f[v - lo + 1] = f[v - lo] + std::max(count[v], 1u);
        return f;
    }
};

// ---- the device backend
class DeviceTrace {
  public:
    DeviceTrace(ss_ctx *ctx, uint64_t num_cycles, uint64_t *const *d_cols, uint32_t ncols) : ctx_(ctx), num_cycles_(num_cycles), n_(16 * num_cycles), ncols_(ncols) {
        for (uint32_t c = 0; c < ncols; ++c) cols_.push_back(d_cols[c]);
        d_status_ = (uint32_t *)alloc(SS_TRACE_STATUS_WORDS * 4);
        check(ss_dev_zero(ctx_, d_status_, SS_TRACE_STATUS_WORDS * 4));
        d_pool_addr_ = (uint32_t *)alloc(n_ / 2 * 4);
    }
    // SSH_TRACE_TIMING: the HOST's time per step of a generation on stderr (uploads, plans, templates, launches; the kernels run behind)
    void lap(const char *what) {
        if (!timing_) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[device trace] %-24s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last_).count());
        t_last_ = now;
    }
    ~DeviceTrace() {
        if (!finished_) (void)ss_ctx_sync(ctx_);                  // (an exception on the way: nothing may still read what is freed here)
        for (void *p : owned_) (void)ss_dev_free(ctx_, p);
    }
    DeviceTrace(const DeviceTrace &) = delete;
    DeviceTrace &operator=(const DeviceTrace &) = delete;
    uint64_t rows() const { return n_; }

    // trace.bin and memory.bin as they are: the only uploads of the size of the input
    void load_inputs(const uint8_t *trace_bin, uint64_t trace_len, const uint8_t *memory_bin, uint64_t memory_len) {
        if (trace_len != 24 * num_cycles_) fail("trace file does not hold the run's cycles");
        if (memory_len % 40) fail("memory file is not a sequence of (u64 address, 32-byte word) records");
        d_states_ = (uint64_t *)upload(trace_bin, trace_len);
        const uint64_t n_records = memory_len / 40;
        // continuous memory has an access per address and n / 2 accesses in all: no valid run touches a cell beyond n / 2
        cells_ = n_ / 2 + 2;
        uint64_t *d_records = n_records ? (uint64_t *)upload(memory_bin, memory_len) : nullptr;
        d_image_ = (uint64_t *)alloc(cells_ * 32);
        check(ss_trace_memory_image(ctx_, d_records, n_records, d_image_, cells_));
        lap("files uploaded");
    }
    void zero_column(int col) { check(ss_dev_zero(ctx_, cols_[col], n_ * 32)); }
    void cpu_cells(const ss_trace_layout &layout, int flags_col, int npc_col, int rc_col, int aux_col, const Felt &pad_value, uint64_t rc_fill) {
        npc_col_ = npc_col;
        check(ss_trace_cpu_cells(ctx_, &layout, d_states_, num_cycles_, d_image_, cells_, pad_value.data(), rc_fill, cols_[flags_col], cols_[npc_col], cols_[rc_col],
                                 cols_[aux_col], d_pool_addr_, d_status_));
        lap("cpu cells launched");
    }
    // a builtin's blocks from its templates: place(sink, template) records template t's cells (every template the same cells)
    template <class Place> void builtin(uint32_t n_templates, const std::vector<uint32_t> &of_block, uint64_t block_rows, uint64_t addr_begin, uint64_t addr_per_block,
                                        const Place &place) {
        std::vector<ss_trace_cell> cells;
        auto values = std::make_shared<std::vector<Felt>>();
        for (uint32_t t = 0; t < n_templates; ++t) {
            TemplateSink sink;
            sink.npc_col = npc_col_;
            place(sink, t);
            std::vector<ss_trace_cell> c;
            sink.finish(c, *values);
            if (t == 0) cells = std::move(c);
            else if (c.size() != cells.size() || memcmp(c.data(), cells.data(), c.size() * sizeof(ss_trace_cell)) != 0) fail("a builtin's instances do not share their cells");
        }
        if (cells.empty()) return;
        const uint32_t n_cells = (uint32_t)cells.size();
        const uint64_t n_blocks = of_block.size();
        const uint32_t *d_of_block = n_templates > 1 ? upload_vec(std::vector<uint32_t>(of_block)) : nullptr;
        const ss_trace_cell *d_cells = upload_vec(std::move(cells));
        const uint64_t *d_values = (const uint64_t *)upload_vec(std::move(*values));
        check(ss_trace_builtin(ctx_, cols_.data(), ncols_, d_cells, n_cells, d_values, n_templates, d_of_block, n_blocks, block_rows, addr_begin, addr_per_block,
                               d_pool_addr_));
        lap("builtin templates");
    }
    // the range-check pool: plan + histogram -> the pool's cells of every cycle; then (later, in the host sections' order) the builtin
    void rc_pool(ss_trace_rc_plan &plan, const RcPoolPlan &pool, const std::vector<uint32_t> &count, int rc_col) {
        plan.rc_lo = pool.lo; plan.rc_hi = pool.hi; plan.n_padding = pool.padding.size(); plan.pad0 = pool.pad0;
        std::vector<uint16_t> padding(pool.padding.begin(), pool.padding.end());
        d_rc_first_ = upload_vec(pool.first(count));
        d_rc_padding_ = padding.empty() ? nullptr : upload_vec(std::move(padding));
        check(ss_trace_rc_pool(ctx_, &plan, d_rc_first_, d_rc_padding_, num_cycles_, cols_[rc_col]));
    }
    void rc_builtin(const ss_trace_rc_plan &plan, std::vector<uint64_t> given3, int rc_col) {
        const uint64_t *d_given = given3.empty() ? nullptr : upload_vec(std::move(given3));
        check(ss_trace_rc_builtin(ctx_, &plan, d_given, d_rc_padding_, cols_[rc_col], cols_[npc_col_], d_pool_addr_));
    }
    void ordered_runs(int col, uint64_t stride, uint64_t off, uint64_t slots, std::vector<uint32_t> first, uint32_t lo, bool diluted) {
        const uint32_t n_values = (uint32_t)first.size() - 1;
        const uint32_t *d_first = upload_vec(std::move(first));
        check(ss_trace_ordered_runs(ctx_, cols_[col], stride, off, slots, d_first, n_values, lo, diluted ? 1 : 0));
    }
    void patch(int col, std::vector<uint64_t> rows, std::vector<uint64_t> values) {
        if (rows.empty()) return;
        const uint64_t count = rows.size();
        const uint64_t *d_rows = upload_vec(std::move(rows)), *d_values = upload_vec(std::move(values));
        check(ss_trace_patch(ctx_, cols_[col], n_, d_rows, d_values, count));
    }
    void ordered_memory(int mem_col, const std::vector<MemoryEntry> &public_memory, uint64_t public_cells, const Felt &pad_value, uint32_t unused_off) {
        std::vector<uint32_t> addr(public_memory.size());
        std::vector<Felt> value(public_memory.size());
        for (size_t k = 0; k < public_memory.size(); ++k) {
            addr[k] = (uint32_t)public_memory[k].address;
            value[k] = felt_from_canonical(public_memory[k].value);
        }
        if (public_memory.size() > public_cells) fail("public memory does not fit");
        const uint32_t n_public = (uint32_t)addr.size();
        const uint32_t *d_addr = addr.empty() ? nullptr : upload_vec(std::move(addr));
        const uint64_t *d_value = value.empty() ? nullptr : (const uint64_t *)upload_vec(std::move(value));
        check(ss_trace_ordered_memory(ctx_, n_, cols_[npc_col_], cols_[mem_col], d_pool_addr_, d_addr, d_value, n_public, public_cells, pad_value.data(), unused_off,
                                      d_status_));
        lap("ordered memory launched");
    }
    // waits for the kernels and turns the status bits into the generator's refusals
    void finish() {
        uint32_t st[SS_TRACE_STATUS_WORDS];
        check(ss_trace_status(ctx_, d_status_, st));
        finished_ = true;
        lap("kernels done (status)");
        const uint32_t err = st[0];
        if (!err) return;
        const std::string where = std::to_string((uint32_t)~st[1]);
        if (err & SS_TRACE_ERR_MISSING_CELL) fail("the run reads a memory cell that memory.bin does not hold (cycle " + where + ")");
        if (err & SS_TRACE_ERR_NOT_INSTRUCTION) fail("a memory cell the run executes is not an instruction (cycle " + where + ")");
        if (err & SS_TRACE_ERR_BAD_OP1_SOURCE) fail("invalid op1 source (cycle " + where + ")");
        if (err & SS_TRACE_ERR_BAD_RES_LOGIC) fail("invalid res logic (cycle " + where + ")");
        if (err & SS_TRACE_ERR_NOT_AN_ADDRESS) fail("a memory cell is used as an address but is not one (cycle " + where + ")");
        if (err & SS_TRACE_ERR_TOO_MANY_GAPS) fail("more memory gaps than cycles to hold them");
        if (err & SS_TRACE_ERR_PUBLIC_ZERO) fail("the public-memory cells of the pool must be the only accesses of address 0");
        if (err & SS_TRACE_ERR_PUBLIC_CELLS) fail("the public-memory cells of the pool must be the only accesses of address 0");
        if (err & SS_TRACE_ERR_NO_ONES) fail("memory must start at address 1");
        if (err & (SS_TRACE_ERR_ADDRESS_RANGE | SS_TRACE_ERR_NOT_SINGLE_VALUED | SS_TRACE_ERR_NOT_CONTINUOUS))
            fail("memory is not continuous and single-valued at address " + where);
        fail("the ordered memory does not fill its column");
    }

  private:
    void check(ss_status st) const { if (st != SS_OK) fail(std::string("device: ") + ss_last_error()); }
    void *alloc(size_t bytes) {
        void *p = nullptr;
        check(ss_dev_alloc(ctx_, bytes, &p));
        owned_.push_back(p);
        return p;
    }
    // device copy of host bytes: on the context's copy stream, the kernels enqueued after this call behind it (the host does not wait
    // for the kernels enqueued so far).  The bytes must stay where they are until finish(): upload_vec keeps a vector that long
    template <class T> const T *upload_vec(std::vector<T> v) {
        auto keep = std::make_shared<std::vector<T>>(std::move(v));
        keep_.push_back(keep);
        return (const T *)upload(keep->data(), keep->size() * sizeof(T));
    }
    void *upload(const void *src, size_t bytes) {
        void *d = alloc(bytes);
        uint64_t ticket = 0;
        check(ss_upload_async(ctx_, d, src, bytes, &ticket));
        check(ss_wait_upload(ctx_, ticket));
        return d;
    }
    ss_ctx *ctx_;
    uint64_t num_cycles_, n_;
    uint32_t ncols_;
    std::vector<uint64_t *> cols_;
    std::vector<void *> owned_;
    std::vector<std::shared_ptr<void>> keep_;
    uint32_t *d_status_ = nullptr, *d_pool_addr_ = nullptr;
    uint64_t *d_states_ = nullptr, *d_image_ = nullptr;
    uint64_t cells_ = 0;
    int npc_col_ = 0;
    bool finished_ = false;
    const bool timing_ = getenv("SSH_TRACE_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t_last_ = std::chrono::steady_clock::now();
    const uint32_t *d_rc_first_ = nullptr;
    const uint16_t *d_rc_padding_ = nullptr;
};

}  // namespace tracedetail
}  // namespace ssh
