#include "extension.hpp"

#include <cstring>
#include <stdexcept>

namespace ssh {

namespace {
void ok(ss_status s) {
    if (s != SS_OK) throw std::runtime_error(ss_last_error());
}
// layouts/src/{recursive,starknet}/mod.rs and the air.rs enums
constexpr uint64_t MEMORY_STEP = 2, RANGE_CHECK_STEP = 4;
constexpr uint64_t RC_OFF_DST = 0, RC_ORDERED = 2;                      // enum RangeCheck
enum { MEM_Z = 0, MEM_A = 1, RC_Z = 2, DC_Z = 3, AGG_Z = 4, AGG_A = 5 };  // challenge indices

ss_perm_operand operand(const uint64_t *col, uint64_t stride, uint64_t a, int64_t v) {
    ss_perm_operand o;
    o.d_data = col; o.stride = stride; o.addr_offset = a; o.value_offset = v;
    return o;
}
bool is_one(const Felt &last) {
    const Felt one = felt_from_u64(1);
    return last[0] == one[0] && last[1] == one[1] && last[2] == one[2] && last[3] == one[3];
}

// What a layout's extension trace is made of, said once for the whole columns (one device) and for their row blocks (one column
// over several devices): the running products (operand = array_chunks::<step>() of a column; output cell `off` of every `step`
// rows of column `col`) and the one affine recurrence.
struct Product {
    ss_perm_operand num, den;
    uint64_t step, off;
    uint32_t col;
    int z, alpha;                       // challenge indices (alpha < 0: single-value terms)
    const char *closes;                 // non-null: the reference asserts that the product ends at one (trace.rs:734, 757-760)
};
struct Aggregate {
    const uint64_t *x;                  // the ordered diluted values: cell x_off of every `step` rows
    uint64_t step, x_off, off;
    uint32_t col;
};
struct Plan {
    uint32_t ncols;
    std::vector<Product> products;
    Aggregate agg;
};
Plan plan_for(const std::string &layout, const TraceColumns &c) {
    const bool recursive = layout == "recursive";
    if (!recursive && layout != "starknet") throw std::runtime_error("unknown layout " + layout);
    if (!c.npc || !c.memory || !c.range_check || (recursive && (!c.diluted_unordered || !c.diluted_ordered)))
        throw std::runtime_error("build_extension_columns: missing trace column");
    const ss_perm_operand mem_n = operand(c.npc, MEMORY_STEP, 0, 1), mem_d = operand(c.memory, MEMORY_STEP, 0, 1);
    const ss_perm_operand rc_n = operand(c.range_check, RANGE_CHECK_STEP, RC_OFF_DST, -1), rc_d = operand(c.range_check, RANGE_CHECK_STEP, RC_ORDERED, -1);
    Plan p;
    if (recursive) {
        // columns [diluted_check_aggregate, diluted_check_permutation, mem_and_rc_permutation];
        // Permutation::col_and_shift: Memory (9, 0), RangeCheck (9, 1), DilutedCheck (8, 0); DILUTED_CHECK_STEP = 1
        p.ncols = 3;
        p.products.push_back({mem_n, mem_d, MEMORY_STEP, 0, 2, MEM_Z, MEM_A, nullptr});
        p.products.push_back({rc_n, rc_d, RANGE_CHECK_STEP, 1, 2, RC_Z, -1, "range-check"});
        p.products.push_back({operand(c.diluted_unordered, 1, 0, -1), operand(c.diluted_ordered, 1, 0, -1), 1, 0, 1, DC_Z, -1, "diluted-check"});
        p.agg = {c.diluted_ordered, 1, 0, 0, 0};
    } else {
        // enum Permutation {Memory = 0, RangeCheck = 1, DilutedCheck = 7}; DilutedCheck {Unordered = 1, Ordered = 5,
        // Aggregate = 3}; DILUTED_CHECK_STEP = 8: everything shares the one permutation column
        constexpr uint64_t STEP = 8;
        p.ncols = 1;
        p.products.push_back({mem_n, mem_d, MEMORY_STEP, 0, 0, MEM_Z, MEM_A, nullptr});
        p.products.push_back({rc_n, rc_d, RANGE_CHECK_STEP, 1, 0, RC_Z, -1, "range-check"});
        p.products.push_back({operand(c.range_check, STEP, 1, -1), operand(c.range_check, STEP, 5, -1), STEP, 7, 0, DC_Z, -1, "diluted-check"});
        p.agg = {c.range_check, STEP, 5, 3, 0};
    }
    return p;
}
Felt felt_of(const uint8_t *p) { Felt f; memcpy(f.data(), p, 32); return f; }
}  // namespace

Matrix build_extension_columns(ss_ctx *ctx, const std::string &layout, const TraceColumns &c, const std::vector<Felt> &ch, bool check) {
    if (ch.size() < 6) throw std::runtime_error("build_extension_columns needs the 6 permutation challenges");
    const Plan plan = plan_for(layout, c);
    const uint64_t n = c.trace_len;
    Matrix out = Matrix::alloc(ctx, plan.ncols, n);
    for (uint64_t *col : out.cols) ok(ss_dev_zero(ctx, col, 32 * n));                  // resize(trace_len, Fp::ZERO)
    for (const Product &p : plan.products) {
        Felt last;
        ok(ss_permutation_product(ctx, &p.num, &p.den, n / p.step, ch[p.z].data(), p.alpha < 0 ? nullptr : ch[p.alpha].data(), out.cols[p.col], p.step, p.off,
                                  (check && p.closes) ? last.data() : nullptr));
        if (check && p.closes && !is_one(last)) throw std::runtime_error(std::string(p.closes) + " permutation product does not close to one");
    }
    const Aggregate &a = plan.agg;
    ok(ss_diluted_aggregate(ctx, a.x, a.step, a.x_off, n / a.step, ch[AGG_Z].data(), ch[AGG_A].data(), out.cols[a.col], a.step, a.off));
    return out;
}

// The reference's loops are sequential in the row index ("TODO: multithread", trace.rs:700); as scans over a monoid they split
// at any row: rank r scans its rows, the R block totals are exchanged ONCE (7 felts per rank), and every rank folds the blocks
// before it into its own - a factor for a running product, a start value for the aggregate's affine maps.  The aggregate's term
// at a block's first row reads the ordered value of the row before it (another rank's): both boundary values travel with the
// totals and the one map between two blocks is composed here, on the host.
Matrix build_extension_blocks(ss_ctx *ctx, const std::string &layout, const TraceColumns &c, const std::vector<Felt> &ch, const BlockGather &g,
                              bool check) {
    if (ch.size() < 6) throw std::runtime_error("build_extension_blocks needs the 6 permutation challenges");
    if (!g.world || g.rank >= g.world || !g.all_gather) throw std::runtime_error("build_extension_blocks: rank / world / all_gather");
    const Plan plan = plan_for(layout, c);
    const uint32_t R = g.world, r = g.rank;
    const uint64_t n = c.trace_len, nb = n / R;
    if (nb * R != n || !nb) throw std::runtime_error("build_extension_blocks: the trace does not split into that many row blocks");
    for (const Product &p : plan.products) if (nb % p.step) throw std::runtime_error("build_extension_blocks: a row block cuts an item of a running product");
    const Aggregate &a = plan.agg;
    if (nb % a.step) throw std::runtime_error("build_extension_blocks: a row block cuts an item of the aggregate");
    Matrix out = Matrix::alloc(ctx, plan.ncols, nb);
    for (uint64_t *col : out.cols) ok(ss_dev_zero(ctx, col, 32 * nb));
    // 1. every scan on this rank's rows
    for (const Product &p : plan.products)
        ok(ss_permutation_product(ctx, &p.num, &p.den, nb / p.step, ch[p.z].data(), p.alpha < 0 ? nullptr : ch[p.alpha].data(), out.cols[p.col], p.step, p.off, nullptr));
    const uint64_t acount = nb / a.step;
    DeviceBuffer maps(ctx, 64 * acount);
    ok(ss_diluted_aggregate_block(ctx, a.x, a.step, a.x_off, acount, r == 0, ch[AGG_Z].data(), ch[AGG_A].data(), maps.u64(), nullptr));
    // 2. the block's totals and the aggregate's two boundary values: np felts, then (M, C), then (x_first, x_last)
    const size_t np = plan.products.size(), per = 32 * (np + 4);
    std::vector<uint8_t> mine(per);
    for (size_t k = 0; k < np; ++k) {
        const Product &p = plan.products[k];
        ok(ss_download(ctx, mine.data() + 32 * k, out.cols[p.col] + 4 * ((nb / p.step - 1) * p.step + p.off), 32));
    }
    ok(ss_download(ctx, mine.data() + 32 * np, maps.u64() + 8 * (acount - 1), 64));
    ok(ss_download(ctx, mine.data() + 32 * (np + 2), a.x + 4 * a.x_off, 32));
    ok(ss_download(ctx, mine.data() + 32 * (np + 3), a.x + 4 * ((acount - 1) * a.step + a.x_off), 32));
    const std::vector<uint8_t> all = g.all_gather(mine);
    if (all.size() != per * R) throw std::runtime_error("build_extension_blocks: all_gather returned another size");
    // 3. the blocks before this one folded in
    for (size_t k = 0; k < np; ++k) {
        const Product &p = plan.products[k];
        Felt before = felt_from_u64(1), whole = felt_from_u64(1);
        for (uint32_t s = 0; s < R; ++s) {
            if (s == r) before = whole;
            whole = felt_mul(whole, felt_of(all.data() + per * s + 32 * k));
        }
        if (r) ok(ss_scale_strided(ctx, out.cols[p.col], p.step, p.off, nb / p.step, before.data()));
        if (check && p.closes && !is_one(whole)) throw std::runtime_error(std::string(p.closes) + " permutation product does not close to one");
    }
    Felt value{}, start{};                       // the aggregate after block s - 1; block 0's maps are constant (acc_0 = 1): any start does
    for (uint32_t s = 0; s <= r; ++s) {
        const uint8_t *blk = all.data() + per * s + 32 * np;
        Felt st = value;
        if (s) {                                 // the term between the blocks: acc (1 + z u) + alpha u^2, u = x_first(s) - x_last(s - 1)
            const Felt u = felt_sub(felt_of(blk + 64), felt_of(blk - per + 96));
            st = felt_add(felt_mul(value, felt_add(felt_from_u64(1), felt_mul(ch[AGG_Z], u))), felt_mul(ch[AGG_A], felt_mul(u, u)));
        }
        if (s == r) { start = st; break; }
        value = felt_add(felt_mul(felt_of(blk), st), felt_of(blk + 32));
    }
    ok(ss_affine_apply(ctx, maps.u64(), acount, start.data(), out.cols[a.col], a.step, a.off));
    ok(ss_ctx_sync(ctx));                        // `maps` goes out of scope
    return out;
}

}  // namespace ssh
