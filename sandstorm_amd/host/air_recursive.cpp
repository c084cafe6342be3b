// air_recursive.cpp — the `recursive` layout's AIR on the C++ host: the 93 constraints of
// layouts/src/recursive/air.rs:61-1180 in the reference's order, its composition constraint
// (air.rs:1183-1199) and the hints (air.rs:1216-1260, layouts/src/utils.rs:14-108); the table and
// lowering machinery is air_layout.{hpp,cpp}.  Mirror of sandstorm_amd/layouts/recursive.py,
// where every piece is documented and validated against the reference's example run; a GPU test
// proves that both hosts emit the same proof for it.
#include "air_layout.hpp"

namespace ssh {

namespace {

using namespace layout;

constexpr uint64_t CYCLE_HEIGHT = 16;
enum { COL_FLAGS, COL_DILUTED_UNORDERED, COL_DILUTED_ORDERED, COL_NPC, COL_MEMORY, COL_RANGE_CHECK, COL_AUXILIARY, COL_DILUTED_AGGREGATE,
       COL_DILUTED_PERMUTATION, COL_MEM_RC_PERMUTATION };
enum { NPC_PC = 0, NPC_INSTRUCTION = 1, NPC_PUB_MEM_ADDR = 2, NPC_PUB_MEM_VAL = 3, NPC_MEM_OP0_ADDR = 4, NPC_MEM_OP0 = 5, NPC_MEM_DST_ADDR = 8,
       NPC_MEM_DST = 9, NPC_MEM_OP1_ADDR = 12, NPC_MEM_OP1 = 13, NPC_PEDERSEN_INPUT0_ADDR = 10, NPC_PEDERSEN_INPUT1_ADDR = 1034,
       NPC_PEDERSEN_OUTPUT_ADDR = 522, NPC_RANGE_CHECK128_ADDR = 74, NPC_BITWISE_POOL_ADDR = 26, NPC_BITWISE_X_OR_Y_ADDR = 42 };
enum { RC_OFF_DST = 0, RC_ORDERED = 2, RC_OFF_OP1 = 4, RC_OFF_OP0 = 8, RC16_COMPONENT = 12 };
enum { AUX_AP = 1, AUX_TMP0 = 3, AUX_OP0_MUL_OP1 = 5, AUX_FP = 9, AUX_TMP1 = 11, AUX_RES = 13 };
enum { F_DST_REG, F_OP0_REG, F_OP1_IMM, F_OP1_FP, F_OP1_AP, F_RES_ADD, F_RES_MUL, F_PC_JUMP_ABS, F_PC_JUMP_REL, F_PC_JNZ, F_AP_ADD, F_AP_ADD1,
       F_OPCODE_CALL, F_OPCODE_RET, F_OPCODE_ASSERT_EQ };
enum { MEM_Z, MEM_A, RC_Z, DC_Z, AGG_Z, AGG_A };

struct Hints {
    Felt initial_ap, initial_pc, final_ap, final_pc, range_check_min, range_check_max, initial_rc_addr, initial_bitwise_addr,
        initial_pedersen_addr, memory_quotient, diluted_cumulative_value;
};

class RecursiveAir : public LayoutAir {
public:
    RecursiveAir(ss_ctx *ctx, const AirPublicInput &pi, uint32_t log_n, uint32_t lb, uint64_t lde_offset) : LayoutAir(ctx, pi, log_n, lb, lde_offset) {
        if (pi.layout != "recursive") throw std::runtime_error("the public input is not of the recursive layout");
        name = "recursive"; num_base_columns = 7; num_extension_columns = 3; num_challenges = 6;
        finish_construction();
    }
    using LayoutAir::describe_tables;

private:
    // ---- hints (air.rs:1216-1260)
    Hints hints(const std::vector<Felt> &ch) const {
        Hints h;
        auto seg = [&](int k, bool stop) { return felt_from_u64(stop ? pi_.segments[k].stop_ptr : pi_.segments[k].begin_addr); };
        h.initial_ap = seg(1, false); h.final_ap = seg(1, true); h.initial_pc = seg(0, false); h.final_pc = seg(0, true);
        h.range_check_min = felt_from_u64(pi_.rc_min); h.range_check_max = felt_from_u64(pi_.rc_max);
        h.initial_pedersen_addr = seg(3, false); h.initial_rc_addr = seg(4, false); h.initial_bitwise_addr = seg(6, false);
        h.memory_quotient = public_memory_quotient(pi_, ch[MEM_Z], ch[MEM_A], n_, 16);
        h.diluted_cumulative_value = diluted_cumulative_value(ch[AGG_Z], ch[AGG_A]);
        return h;
    }

    // ---- periodic columns: the Pedersen points, one per 4-row step of a 2048-row hash
    size_t num_periodic_columns() const override { return 2; }
    std::vector<Felt> column_values(size_t c) const override { return pedersen_column((int)c); }
    uint64_t column_period(size_t) const override { return 2048; }

    // ---- the composition: sum_i alpha^i numerator_i * multiplier(domain_i), grouped by domain (first-use order)
    int composition(Graph &g, const std::vector<Felt> &ch, const Felt &alpha) override {
        const Hints h = hints(ch);
        auto T = [&](uint32_t col, uint64_t off) { return E{&g, g.trace(col, (uint32_t)off)}; };
        auto C = [&](uint64_t v) { return E{&g, g.constant_u64(v)}; };
        auto CF = [&](const Felt &f) { return E{&g, g.constant(f)}; };                 // structural constants only
        // per-proof values: interned by symbol, so that the program's shape is the same for every statement (air_program.hpp)
#define HINT(name) E{&g, g.runtime_constant(Graph::sym("hint." #name), h.name)}
#define CHAL(index) E{&g, g.runtime_constant(Graph::sym("challenge", index), ch[index])}
        auto pow2 = [&](unsigned k) { return CF(felt_pow(felt_from_u64(2), k)); };
        auto flag = [&](int f) { return T(COL_FLAGS, f) - (T(COL_FLAGS, f + 1) + T(COL_FLAGS, f + 1)); };
        auto npc = [&](uint64_t cell, uint64_t cycle = 0) { return T(COL_NPC, CYCLE_HEIGHT * cycle + cell); };
        auto rc = [&](uint64_t cell) { return T(COL_RANGE_CHECK, cell); };
        auto aux = [&](uint64_t cell, uint64_t cycle = 0) { return T(COL_AUXILIARY, CYCLE_HEIGHT * cycle + cell); };
        const E one = C(1), two = C(2), four = C(4), offset_size = C(1ull << 16), half_offset_size = C(1ull << 15);

        Composer composer(*this, g, alpha);
        auto add = [&](const std::string &dom_name, const Domain &d, const E &numerator) { composer.add(dom_name, d, numerator); };
        const Domain ALL_CYCLES = every(16), ALL_CYCLES_EXCEPT_LAST = every_except_last(16);
        const Domain FLAG_ROWS{{{n_ / 16, 15 * n_ / 16}}, {{n_, 0}}}, FLAG_ZERO_ROWS{{}, {{n_ / 16, 15 * n_ / 16}}};
        const Domain FIRST_ROW{{}, {{1, 0}}}, LAST_CYCLE = row_from_end(16);
#define ADD(dom, expr) add(#dom, dom, expr)

        // ---- cpu (air.rs:82-443)
        const E flag_op1_base_op0_0 = one - (flag(F_OP1_IMM) + flag(F_OP1_AP) + flag(F_OP1_FP));
        const E flag_res_op1_0 = one - (flag(F_RES_ADD) + flag(F_RES_MUL) + flag(F_PC_JNZ));
        const E flag_pc_update_regular_0 = one - (flag(F_PC_JUMP_ABS) + flag(F_PC_JUMP_REL) + flag(F_PC_JNZ));
        const E fp_update_regular_0 = one - (flag(F_OPCODE_CALL) + flag(F_OPCODE_RET));
        const E npc_reg_0 = npc(NPC_PC) + flag(F_OP1_IMM) + one;
        const E whole_flag_prefix = T(COL_FLAGS, 0);
        ADD(FLAG_ROWS, flag(F_DST_REG) * flag(F_DST_REG) - flag(F_DST_REG));
        ADD(FLAG_ZERO_ROWS, whole_flag_prefix);
        ADD(ALL_CYCLES, npc(NPC_INSTRUCTION) - (((whole_flag_prefix * offset_size + rc(RC_OFF_OP1)) * offset_size + rc(RC_OFF_OP0)) * offset_size + rc(RC_OFF_DST)));
        for (const E &e : {flag_op1_base_op0_0, flag_res_op1_0, flag_pc_update_regular_0, fp_update_regular_0}) ADD(ALL_CYCLES, e * e - e);
        ADD(ALL_CYCLES, npc(NPC_MEM_DST_ADDR) + half_offset_size - (flag(F_DST_REG) * aux(AUX_FP) + (one - flag(F_DST_REG)) * aux(AUX_AP) + rc(RC_OFF_DST)));
        ADD(ALL_CYCLES, npc(NPC_MEM_OP0_ADDR) + half_offset_size - (flag(F_OP0_REG) * aux(AUX_FP) + (one - flag(F_OP0_REG)) * aux(AUX_AP) + rc(RC_OFF_OP0)));
        ADD(ALL_CYCLES, npc(NPC_MEM_OP1_ADDR) + half_offset_size
                            - (flag(F_OP1_IMM) * npc(NPC_PC) + flag(F_OP1_AP) * aux(AUX_AP) + flag(F_OP1_FP) * aux(AUX_FP)
                               + flag_op1_base_op0_0 * npc(NPC_MEM_OP0) + rc(RC_OFF_OP1)));
        ADD(ALL_CYCLES, aux(AUX_OP0_MUL_OP1) - npc(NPC_MEM_OP0) * npc(NPC_MEM_OP1));
        ADD(ALL_CYCLES, (one - flag(F_PC_JNZ)) * aux(AUX_RES)
                            - (flag(F_RES_ADD) * (npc(NPC_MEM_OP0) + npc(NPC_MEM_OP1)) + flag(F_RES_MUL) * aux(AUX_OP0_MUL_OP1) + flag_res_op1_0 * npc(NPC_MEM_OP1)));
        ADD(ALL_CYCLES_EXCEPT_LAST, aux(AUX_TMP0) - flag(F_PC_JNZ) * npc(NPC_MEM_DST));
        ADD(ALL_CYCLES_EXCEPT_LAST, aux(AUX_TMP1) - aux(AUX_TMP0) * aux(AUX_RES));
        ADD(ALL_CYCLES_EXCEPT_LAST, (one - flag(F_PC_JNZ)) * npc(NPC_PC, 1) + aux(AUX_TMP0) * (npc(NPC_PC, 1) - (npc(NPC_PC) + npc(NPC_MEM_OP1)))
                                        - (flag_pc_update_regular_0 * npc_reg_0 + flag(F_PC_JUMP_ABS) * aux(AUX_RES) + flag(F_PC_JUMP_REL) * (npc(NPC_PC) + aux(AUX_RES))));
        ADD(ALL_CYCLES_EXCEPT_LAST, (aux(AUX_TMP1) - flag(F_PC_JNZ)) * (npc(NPC_PC, 1) - npc_reg_0));
        ADD(ALL_CYCLES_EXCEPT_LAST, aux(AUX_AP, 1) - (aux(AUX_AP) + flag(F_AP_ADD) * aux(AUX_RES) + flag(F_AP_ADD1) + flag(F_OPCODE_CALL) * two));
        ADD(ALL_CYCLES_EXCEPT_LAST, aux(AUX_FP, 1) - (fp_update_regular_0 * aux(AUX_FP) + flag(F_OPCODE_RET) * npc(NPC_MEM_DST) + flag(F_OPCODE_CALL) * (aux(AUX_AP) + two)));
        ADD(ALL_CYCLES, flag(F_OPCODE_CALL) * (npc(NPC_MEM_DST) - aux(AUX_FP)));
        ADD(ALL_CYCLES, flag(F_OPCODE_CALL) * (npc(NPC_MEM_OP0) - (npc(NPC_PC) + flag(F_OP1_IMM) + one)));
        ADD(ALL_CYCLES, flag(F_OPCODE_CALL) * (rc(RC_OFF_DST) - half_offset_size));
        ADD(ALL_CYCLES, flag(F_OPCODE_CALL) * (rc(RC_OFF_OP0) - (half_offset_size + one)));
        ADD(ALL_CYCLES, flag(F_OPCODE_CALL) * (flag(F_OPCODE_CALL) + flag(F_OPCODE_CALL) + one + one - (flag(F_DST_REG) + flag(F_OP0_REG) + four)));
        ADD(ALL_CYCLES, flag(F_OPCODE_RET) * (rc(RC_OFF_DST) + two - half_offset_size));
        ADD(ALL_CYCLES, flag(F_OPCODE_RET) * (rc(RC_OFF_OP1) + one - half_offset_size));
        ADD(ALL_CYCLES, flag(F_OPCODE_RET) * (flag(F_PC_JUMP_ABS) + flag(F_DST_REG) + flag(F_OP1_FP) + flag_res_op1_0 - four));
        ADD(ALL_CYCLES, flag(F_OPCODE_ASSERT_EQ) * (npc(NPC_MEM_DST) - aux(AUX_RES)));
        ADD(FIRST_ROW, aux(AUX_AP) - HINT(initial_ap));
        ADD(FIRST_ROW, aux(AUX_FP) - HINT(initial_ap));
        ADD(FIRST_ROW, npc(NPC_PC) - HINT(initial_pc));
        ADD(LAST_CYCLE, aux(AUX_AP) - HINT(final_ap));
        ADD(LAST_CYCLE, aux(AUX_FP) - HINT(initial_ap));
        ADD(LAST_CYCLE, npc(NPC_PC) - HINT(final_pc));

        // ---- memory (air.rs:444-497)
        const Domain EVERY_2ND_EXCEPT_LAST = every_except_last(2), SECOND_LAST_ROW = row_from_end(2);
        auto mem = [&](uint64_t cell, uint64_t k = 0) { return T(COL_MEMORY, 2 * k + cell); };
        auto perm_memory = [&](uint64_t k = 0) { return T(COL_MEM_RC_PERMUTATION, 2 * k); };
        auto perm_rc = [&](uint64_t k = 0) { return T(COL_MEM_RC_PERMUTATION, 4 * k + 1); };
        {
            const E z = CHAL(MEM_Z), a = CHAL(MEM_A);
            const E address_diff = mem(0, 1) - mem(0);
            ADD(FIRST_ROW, (z - (mem(0) + a * mem(1))) * perm_memory() + npc(NPC_PC) + a * npc(NPC_INSTRUCTION) - z);
            ADD(EVERY_2ND_EXCEPT_LAST, (z - (mem(0, 1) + a * mem(1, 1))) * perm_memory(1) - (z - (T(COL_NPC, 2) + a * T(COL_NPC, 3))) * perm_memory());
            ADD(SECOND_LAST_ROW, perm_memory() - HINT(memory_quotient));
            ADD(EVERY_2ND_EXCEPT_LAST, address_diff * address_diff - address_diff);
            ADD(EVERY_2ND_EXCEPT_LAST, (address_diff - one) * (mem(1) - mem(1, 1)));
            ADD(FIRST_ROW, mem(0) - one);
            ADD(ALL_CYCLES, npc(NPC_PUB_MEM_ADDR));
            ADD(ALL_CYCLES, npc(NPC_PUB_MEM_VAL));
        }
        // ---- 16-bit range check (air.rs:499-538)
        const Domain EVERY_4TH_EXCEPT_LAST = every_except_last(4), FOURTH_LAST_ROW = row_from_end(4);
        auto rc_ordered = [&](uint64_t k = 0) { return T(COL_RANGE_CHECK, 4 * k + RC_ORDERED); };
        {
            const E z = CHAL(RC_Z);
            const E diff = rc_ordered(1) - rc_ordered();
            ADD(FIRST_ROW, (z - rc_ordered()) * perm_rc() + rc(RC_OFF_DST) - z);
            ADD(EVERY_4TH_EXCEPT_LAST, (z - rc_ordered(1)) * perm_rc(1) - (z - T(COL_RANGE_CHECK, 4)) * perm_rc());
            ADD(FOURTH_LAST_ROW, perm_rc() - one);
            ADD(EVERY_4TH_EXCEPT_LAST, diff * diff - diff);
            ADD(FIRST_ROW, rc_ordered() - HINT(range_check_min));
            ADD(FOURTH_LAST_ROW, rc_ordered() - HINT(range_check_max));
        }
        // ---- diluted check (air.rs:540-603)
        const Domain EVERY_ROW_EXCEPT_LAST = every_except_last(1), LAST_ROW = row_from_end(1);
        {
            const E z = CHAL(DC_Z), za = CHAL(AGG_Z), aa = CHAL(AGG_A);
            auto un = [&](uint64_t o = 0) { return T(COL_DILUTED_UNORDERED, o); };
            auto od = [&](uint64_t o = 0) { return T(COL_DILUTED_ORDERED, o); };
            auto perm = [&](uint64_t o = 0) { return T(COL_DILUTED_PERMUTATION, o); };
            auto agg = [&](uint64_t o = 0) { return T(COL_DILUTED_AGGREGATE, o); };
            const E diff = od(1) - od();
            ADD(FIRST_ROW, (z - od()) * perm() + un() - z);
            ADD(EVERY_ROW_EXCEPT_LAST, (z - od(1)) * perm(1) - (z - un(1)) * perm());
            ADD(LAST_ROW, perm() - one);
            ADD(FIRST_ROW, agg() - one);
            ADD(FIRST_ROW, od() - C(0));
            ADD(EVERY_ROW_EXCEPT_LAST, agg(1) - (agg() * (one + za * diff) + aa * diff * diff));
            ADD(LAST_ROW, agg() - HINT(diluted_cumulative_value));
        }
        // ---- Pedersen builtin (air.rs:605-895)
        {
            auto suffix = [&](uint64_t k = 0) { return T(COL_AUXILIARY, 4 * k); };
            auto slope = [&](uint64_t k = 0) { return T(COL_AUXILIARY, 4 * k + 2); };
            auto sum_x = [&](uint64_t k = 0) { return T(COL_RANGE_CHECK, 4 * k + 1); };
            auto sum_y = [&](uint64_t k = 0) { return T(COL_RANGE_CHECK, 4 * k + 3); };
            const E bit_251_196_192 = T(COL_AUXILIARY, 7), bit_251_196 = T(COL_AUXILIARY, 1022);
            const E point_x{&g, g.table(0)}, point_y{&g, g.table(1)};
            auto bit = [&](uint64_t k) { return suffix(k) - (suffix(k + 1) + suffix(k + 1)); };
            const E b0 = bit(0), b0_negate = one - b0;
            const Domain EVERY_1024 = every(1024), EVERY_2048 = every(2048), EVERY_2048_EXCEPT_LAST = every_except_last(2048);
            const Domain PEDERSEN_TRANSITION{{{n_ / 1024, 255 * n_ / 256}}, {{n_ / 4, 0}}};
            const Domain STEP_252{{}, {{n_ / 1024, 63 * n_ / 64}}}, STEP_255{{}, {{n_ / 1024, 255 * n_ / 256}}};
            ADD(EVERY_1024, bit_251_196_192 * bit(0));
            ADD(EVERY_1024, bit_251_196_192 * (suffix(1) - suffix(192) * pow2(191)));
            ADD(EVERY_1024, bit_251_196_192 - bit_251_196 * bit(192));
            ADD(EVERY_1024, bit_251_196 * (suffix(193) - suffix(196) * pow2(3)));
            ADD(EVERY_1024, bit_251_196 - bit(251) * bit(196));
            ADD(EVERY_1024, bit(251) * (suffix(197) - suffix(251) * pow2(54)));
            ADD(PEDERSEN_TRANSITION, b0 * (b0 - one));
            ADD(STEP_252, suffix());
            ADD(STEP_255, suffix());
            ADD(PEDERSEN_TRANSITION, b0 * (sum_y() - point_y) - slope() * (sum_x() - point_x));
            ADD(PEDERSEN_TRANSITION, slope() * slope() - b0 * (sum_x() + point_x + sum_x(1)));
            ADD(PEDERSEN_TRANSITION, b0 * (sum_y() + sum_y(1)) - slope() * (sum_x() - sum_x(1)));
            ADD(PEDERSEN_TRANSITION, b0_negate * (sum_x(1) - sum_x()));
            ADD(PEDERSEN_TRANSITION, b0_negate * (sum_y(1) - sum_y()));
            ADD(EVERY_2048, sum_x(256) - sum_x(255));
            ADD(EVERY_2048, sum_y(256) - sum_y(255));
            ADD(EVERY_2048, sum_x() - CF(pedersen_coord(0, 0)));
            ADD(EVERY_2048, sum_y() - CF(pedersen_coord(0, 1)));
            ADD(EVERY_2048, T(COL_NPC, NPC_PEDERSEN_INPUT0_ADDR + 1) - suffix());
            ADD(EVERY_2048_EXCEPT_LAST, T(COL_NPC, 2048 + NPC_PEDERSEN_INPUT0_ADDR) - (T(COL_NPC, NPC_PEDERSEN_OUTPUT_ADDR) + one));
            ADD(FIRST_ROW, T(COL_NPC, NPC_PEDERSEN_INPUT0_ADDR) - HINT(initial_pedersen_addr));
            ADD(EVERY_2048, T(COL_NPC, NPC_PEDERSEN_INPUT1_ADDR + 1) - suffix(256));
            ADD(EVERY_2048, T(COL_NPC, NPC_PEDERSEN_INPUT1_ADDR) - (T(COL_NPC, NPC_PEDERSEN_INPUT0_ADDR) + one));
            ADD(EVERY_2048, T(COL_NPC, NPC_PEDERSEN_OUTPUT_ADDR + 1) - sum_x(511));
            ADD(EVERY_2048, T(COL_NPC, NPC_PEDERSEN_OUTPUT_ADDR) - (T(COL_NPC, NPC_PEDERSEN_INPUT1_ADDR) + one));
        }
        // ---- range-check builtin (air.rs:899-918)
        const Domain EVERY_128 = every(128), EVERY_128_EXCEPT_LAST = every_except_last(128);
        {
            E value = T(COL_RANGE_CHECK, RC16_COMPONENT);
            for (uint64_t k = 1; k < 8; ++k) value = value * offset_size + T(COL_RANGE_CHECK, CYCLE_HEIGHT * k + RC16_COMPONENT);
            ADD(EVERY_128, value - T(COL_NPC, NPC_RANGE_CHECK128_ADDR + 1));
            ADD(EVERY_128_EXCEPT_LAST, T(COL_NPC, 128 + NPC_RANGE_CHECK128_ADDR) - (T(COL_NPC, NPC_RANGE_CHECK128_ADDR) + one));
            ADD(FIRST_ROW, T(COL_NPC, NPC_RANGE_CHECK128_ADDR) - HINT(initial_rc_addr));
        }
        // ---- bitwise builtin (air.rs:920-1081)
        {
            auto bw = [&](uint64_t o) { return T(COL_DILUTED_UNORDERED, o); };
            auto pool_addr = [&](uint64_t k) { return T(COL_NPC, 32 * k + NPC_BITWISE_POOL_ADDR); };
            auto pool_val = [&](uint64_t k) { return T(COL_NPC, 32 * k + NPC_BITWISE_POOL_ADDR + 1); };
            const Domain EVERY_32 = every(32);
            const Domain BITWISE_TRANSITION{{{n_ / 128, 3 * n_ / 4}}, {{n_ / 32, 0}}};
            Domain EVERY_16_BIT_SEGMENT;
            for (uint64_t k = 0; k < 16; ++k) EVERY_16_BIT_SEGMENT.den.push_back({n_ / 128, k * n_ / 64});
            E sum_var = bw(0);
            for (unsigned chunk = 0; chunk < 4; ++chunk)
                for (unsigned stream = 0; stream < 4; ++stream) {
                    if (chunk == 0 && stream == 0) continue;
                    sum_var = sum_var + bw(8 * chunk + 2 * stream) * pow2(64 * chunk + stream);
                }
            ADD(FIRST_ROW, pool_addr(0) - HINT(initial_bitwise_addr));
            ADD(BITWISE_TRANSITION, pool_addr(1) - (pool_addr(0) + one));
            ADD(EVERY_128, T(COL_NPC, NPC_BITWISE_X_OR_Y_ADDR) - (pool_addr(3) + one));
            ADD(EVERY_128_EXCEPT_LAST, pool_addr(4) - (T(COL_NPC, NPC_BITWISE_X_OR_Y_ADDR) + one));
            ADD(EVERY_32, sum_var - pool_val(0));
            ADD(EVERY_128, T(COL_NPC, NPC_BITWISE_X_OR_Y_ADDR + 1) - (pool_val(2) + pool_val(3)));
            ADD(EVERY_16_BIT_SEGMENT, bw(0) + bw(32) - (bw(96) + bw(64) + bw(64)));
            const uint64_t cells[4] = {1, 65, 33, 97};
            for (unsigned k = 0; k < 4; ++k)
                ADD(EVERY_128, (bw(64 + 24 + 2 * k) + bw(96 + 24 + 2 * k)) * pow2(k == 3 ? 8 : 4) - bw(cells[k]));
        }
#undef ADD
        return composer.total();
    }

};

}  // namespace

std::unique_ptr<Air> make_recursive_air(ss_ctx *ctx, const AirPublicInput &pi, uint32_t log_n, uint32_t log_blowup, uint64_t lde_offset) {
    return std::unique_ptr<Air>(new RecursiveAir(ctx, pi, log_n, log_blowup, lde_offset));
}

}  // namespace ssh
