#include "extension.hpp"

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
bool is_one(const uint64_t last[4]) {
    const Felt one = felt_from_u64(1);
    return last[0] == one[0] && last[1] == one[1] && last[2] == one[2] && last[3] == one[3];
}
}  // namespace

Matrix build_extension_columns(ss_ctx *ctx, const std::string &layout, const TraceColumns &c, const std::vector<Felt> &ch, bool check) {
    if (ch.size() < 6) throw std::runtime_error("build_extension_columns needs the 6 permutation challenges");
    const bool recursive = layout == "recursive";
    if (!recursive && layout != "starknet") throw std::runtime_error("unknown layout " + layout);
    if (!c.npc || !c.memory || !c.range_check || (recursive && (!c.diluted_unordered || !c.diluted_ordered)))
        throw std::runtime_error("build_extension_columns: missing trace column");
    const uint64_t n = c.trace_len;
    Matrix out = Matrix::alloc(ctx, recursive ? 3 : 1, n);
    for (uint64_t *col : out.cols) ok(ss_dev_zero(ctx, col, 32 * n));                  // resize(trace_len, Fp::ZERO)
    uint64_t last_rc[4], last_dc[4];
    const ss_perm_operand mem_n = operand(c.npc, MEMORY_STEP, 0, 1), mem_d = operand(c.memory, MEMORY_STEP, 0, 1);
    const ss_perm_operand rc_n = operand(c.range_check, RANGE_CHECK_STEP, RC_OFF_DST, -1), rc_d = operand(c.range_check, RANGE_CHECK_STEP, RC_ORDERED, -1);
    if (recursive) {
        // Permutation::col_and_shift: Memory (9, 0), RangeCheck (9, 1), DilutedCheck (8, 0); DILUTED_CHECK_STEP = 1
        uint64_t *agg = out.cols[0], *dperm = out.cols[1], *mem_rc = out.cols[2];
        ok(ss_permutation_product(ctx, &mem_n, &mem_d, n / MEMORY_STEP, ch[MEM_Z].data(), ch[MEM_A].data(), mem_rc, MEMORY_STEP, 0, nullptr));
        ok(ss_permutation_product(ctx, &rc_n, &rc_d, n / RANGE_CHECK_STEP, ch[RC_Z].data(), nullptr, mem_rc, RANGE_CHECK_STEP, 1, last_rc));
        const ss_perm_operand dc_n = operand(c.diluted_unordered, 1, 0, -1), dc_d = operand(c.diluted_ordered, 1, 0, -1);
        ok(ss_permutation_product(ctx, &dc_n, &dc_d, n, ch[DC_Z].data(), nullptr, dperm, 1, 0, last_dc));
        ok(ss_diluted_aggregate(ctx, c.diluted_ordered, 1, 0, n, ch[AGG_Z].data(), ch[AGG_A].data(), agg, 1, 0));
    } else {
        // enum Permutation {Memory = 0, RangeCheck = 1, DilutedCheck = 7}; DilutedCheck {Unordered = 1, Ordered = 5,
        // Aggregate = 3}; DILUTED_CHECK_STEP = 8: everything shares the one permutation column
        constexpr uint64_t STEP = 8;
        uint64_t *perm = out.cols[0];
        ok(ss_permutation_product(ctx, &mem_n, &mem_d, n / MEMORY_STEP, ch[MEM_Z].data(), ch[MEM_A].data(), perm, MEMORY_STEP, 0, nullptr));
        ok(ss_permutation_product(ctx, &rc_n, &rc_d, n / RANGE_CHECK_STEP, ch[RC_Z].data(), nullptr, perm, RANGE_CHECK_STEP, 1, last_rc));
        const ss_perm_operand dc_n = operand(c.range_check, STEP, 1, -1), dc_d = operand(c.range_check, STEP, 5, -1);
        ok(ss_permutation_product(ctx, &dc_n, &dc_d, n / STEP, ch[DC_Z].data(), nullptr, perm, STEP, 7, last_dc));
        ok(ss_diluted_aggregate(ctx, c.range_check, STEP, 5, n / STEP, ch[AGG_Z].data(), ch[AGG_A].data(), perm, STEP, 3));
    }
    if (check) {
        if (!is_one(last_rc)) throw std::runtime_error("range-check permutation product does not close to one");
        if (!is_one(last_dc)) throw std::runtime_error("diluted-check permutation product does not close to one");
    }
    return out;
}

}  // namespace ssh
