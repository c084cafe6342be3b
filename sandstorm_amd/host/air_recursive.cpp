// air_recursive.cpp — the `recursive` layout's AIR on the C++ host: the 93 constraints of
// layouts/src/recursive/air.rs:61-1180 in the reference's order, its composition constraint
// (air.rs:1183-1199), the hints (air.rs:1216-1260, layouts/src/utils.rs:14-108) and the
// periodic / zerofier tables the constraint VM reads.  Mirror of sandstorm_amd/layouts/recursive.py,
// where every piece is documented and validated against the reference's example run; a GPU test
// proves that both hosts emit the same proof for it.
#include <algorithm>
#include <cstring>
#include <map>
#include <stdexcept>

#include "prover.hpp"
#include "public_input.hpp"

namespace ssh {

namespace {

void ok(ss_status s) {
    if (s != SS_OK) throw std::runtime_error(ss_last_error());
}

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

// ---- expression wrapper over the hash-consed Graph
struct E {
    Graph *g;
    int id;
};
E operator+(const E &a, const E &b) { return E{a.g, a.g->add(a.id, b.id)}; }
E operator-(const E &a, const E &b) { return E{a.g, a.g->sub(a.id, b.id)}; }
E operator*(const E &a, const E &b) { return E{a.g, a.g->mul(a.id, b.id)}; }

// ---- domains: multiplier prod(num) / prod(den), factor (p, e) = X^p - g^e
struct Factor {
    uint64_t p, e;
    bool operator<(const Factor &o) const { return p != o.p ? p < o.p : e < o.e; }
    bool operator==(const Factor &o) const { return p == o.p && e == o.e; }
};
struct Domain { std::vector<Factor> num, den; };

struct TableSpec {
    int kind;                               // 0 pedersen x, 1 pedersen y, 2 periodic multiplier, 3 full-length inverse
    std::vector<Factor> num, den;           // kind 2
    uint64_t e = 0;                         // kind 3: 1 / (X - g^e)
    bool operator<(const TableSpec &o) const {
        if (kind != o.kind) return kind < o.kind;
        if (e != o.e) return e < o.e;
        if (num != o.num) return num < o.num;
        return den < o.den;
    }
};
struct Hints {
    Felt initial_ap, initial_pc, final_ap, final_pc, range_check_min, range_check_max, initial_rc_addr, initial_bitwise_addr,
        initial_pedersen_addr, memory_quotient, diluted_cumulative_value;
};

// builtins/src/pedersen/constants.rs:5-30 (canonical little-endian limbs)
const uint64_t PEDERSEN_POINTS[5][2][4] = {
    {{0x551fde4050ca6804ull, 0x716b0b1022947733ull, 0x00ee1b87eb599f16ull, 0x049ee3eba8c16007ull}, {0xd0405d266e10268aull, 0x4e621062c0e056c1ull, 0xf346d49d06ea0ed3ull, 0x03ca0cfe4b3bc6ddull}},
    {{0x1080d17957ebe47bull, 0x8fa8120b6d56eb0cull, 0x969c748655fca9e5ull, 0x0234287dcbaffe7full}, {0x6ed0268ee89e5615ull, 0x940135dd7a6c94ccull, 0x1e889527d41f4e39ull, 0x03b056f100f96fb2ull}},
    {{0xb7a6932dba8aa378ull, 0x99099ec1de5e3018ull, 0x3f9dab2656558f33ull, 0x04fa56f376c83db3ull}, {0x5168f4e80ff5b54dull, 0x562761f92a7a23b4ull, 0x8113e0c0e47e4401ull, 0x03fa0984c931c9e3ull}},
    {{0x3aa372f0bd2d6997ull, 0x40c690c74709e90full, 0x764910f75b45f74bull, 0x04ba4cc166be8decull}, {0x48151f27b24b219cull, 0xcac5c59a5ce5ae7cull, 0x4b971e46c4ede85full, 0x0040301cf5c1751full}},
    {{0xd36ff12c49a58202ull, 0x2ca65048d53fb325ull, 0x6e44cca8f61a63bbull, 0x054302dcb0e6cc1cull}, {0x879dcc77e99c2426ull, 0xce98ad783c25561aull, 0xb348046268d8ae25ull, 0x01b77b3e37d13504ull}},
};
Felt pedersen_coord(int point, int which) {
    Felt c;
    memcpy(c.data(), PEDERSEN_POINTS[point][which], 32);
    return felt_from_canonical(c);
}
struct Pt { Felt x, y; };
Pt ec_double(const Pt &p) {
    const Felt xx = felt_mul(p.x, p.x);
    const Felt lam = felt_mul(felt_add(felt_add(felt_add(xx, xx), xx), felt_from_u64(1)), felt_inv(felt_add(p.y, p.y)));
    const Felt x3 = felt_sub(felt_mul(lam, lam), felt_add(p.x, p.x));
    return Pt{x3, felt_sub(felt_mul(lam, felt_sub(p.x, x3)), p.y)};
}
// the 512 values of a Pedersen periodic column (builtins/src/pedersen/periodic.rs:1211-1250)
std::vector<Felt> pedersen_column(int which) {
    std::vector<Felt> out;
    for (int e = 0; e < 2; ++e) {
        std::vector<Pt> half;
        Pt acc{pedersen_coord(1 + 2 * e, 0), pedersen_coord(1 + 2 * e, 1)};
        for (int i = 0; i < 248; ++i) { half.push_back(acc); acc = ec_double(acc); }
        acc = Pt{pedersen_coord(2 + 2 * e, 0), pedersen_coord(2 + 2 * e, 1)};
        for (int i = 0; i < 4; ++i) { half.push_back(acc); acc = ec_double(acc); }
        for (int i = 0; i < 4; ++i) half.push_back(half[251]);
        for (auto &p : half) out.push_back(which ? p.y : p.x);
    }
    return out;
}
// coefficients of the interpolant over <w_m> (natural order), m a power of two: plain O(m log m) inverse transform
std::vector<Felt> interpolate(std::vector<Felt> a) {
    const size_t m = a.size();
    uint32_t lg = 0;
    while ((1ull << lg) < m) ++lg;
    for (size_t i = 0; i < m; ++i) {
        size_t r = 0;
        for (uint32_t b = 0; b < lg; ++b) r |= ((i >> b) & 1) << (lg - 1 - b);
        if (r > i) std::swap(a[i], a[r]);
    }
    const Felt w_inv = felt_inv(root_of_unity(lg));
    for (size_t len = 2; len <= m; len <<= 1) {
        const Felt wl = felt_pow(w_inv, m / len);
        for (size_t s = 0; s < m; s += len) {
            Felt w = felt_from_u64(1);
            for (size_t k = 0; k < len / 2; ++k) {
                const Felt u = a[s + k], v = felt_mul(a[s + k + len / 2], w);
                a[s + k] = felt_add(u, v); a[s + k + len / 2] = felt_sub(u, v);
                w = felt_mul(w, wl);
            }
        }
    }
    const Felt m_inv = felt_inv(felt_from_u64(m));
    for (auto &v : a) v = felt_mul(v, m_inv);
    return a;
}

class RecursiveAir : public Air {
public:
    RecursiveAir(ss_ctx *ctx, const AirPublicInput &pi, uint32_t log_n, uint32_t lb, uint64_t lde_offset)
        : ctx_(ctx), pi_(pi), log_n_(log_n), lb_(lb), offset_(lde_offset), n_(1ull << log_n), g_(root_of_unity(log_n)) {
        if (pi.layout != "recursive") throw std::runtime_error("the public input is not of the recursive layout");
        name = "recursive"; num_base_columns = 7; num_extension_columns = 3; num_challenges = 6;
        // register every table (their set and order do not depend on the challenges) and collect the mask
        std::vector<Felt> ch(6, felt_from_u64(2));
        Graph g;
        const int root = composition(g, ch, felt_from_u64(17));
        std::set<std::pair<uint32_t, uint32_t>> cells;
        std::vector<char> seen(g.nodes().size(), 0);
        std::vector<int> stack{root};
        while (!stack.empty()) {
            const int id = stack.back(); stack.pop_back();
            if (seen[id]) continue;
            seen[id] = 1;
            const Node &nd = g.nodes()[id];
            if (nd.kind == NodeKind::Trace) cells.insert({nd.p0, nd.p1});
            if (nd.a >= 0) stack.push_back(nd.a);
            if (nd.b >= 0) stack.push_back(nd.b);
        }
        mask.assign(cells.begin(), cells.end());
        if (ctx_) build_tables();
    }

    AirProgramData build_program(uint64_t n, const std::vector<Felt> &ch, const Felt &alpha) override {
        if (n != n_) throw std::runtime_error("this recursive AIR was built for another trace length");
        Graph g;
        const int root = composition(g, ch, alpha);
        AirProgramData pd;
        pd.program = lower(g, root);
        pd.d_tables = tables_ ? tables_->u64() : nullptr;
        pd.table_desc = desc_;
        return pd;
    }

    Felt composition_at(uint64_t n, const std::vector<Felt> &ch, const Felt &alpha, const Felt &z, const std::vector<Felt> &ood) override {
        if (n != n_) throw std::runtime_error("this recursive AIR was built for another trace length");
        Graph g;
        const int root = composition(g, ch, alpha);
        std::map<std::pair<uint32_t, uint32_t>, Felt> cell;
        for (size_t j = 0; j < mask.size(); ++j) cell[mask[j]] = ood[j];
        return evaluate(g, root, z, [&](uint32_t c, uint32_t o) { return cell.at({c, o}); }, [&](uint32_t t) { return table_value_at(specs_.at(t), z); });
    }

    // flat description of the tables for host-side checks: per table kind, e, #num, (p, e)..., #den, (p, e)...
    std::vector<uint64_t> describe_tables() const {
        std::vector<uint64_t> out{specs_.size()};
        for (auto &s : specs_) {
            out.push_back((uint64_t)s.kind); out.push_back(s.e);
            out.push_back(s.num.size());
            for (auto &f : s.num) { out.push_back(f.p); out.push_back(f.e); }
            out.push_back(s.den.size());
            for (auto &f : s.den) { out.push_back(f.p); out.push_back(f.e); }
        }
        return out;
    }

private:
    // ---- hints (air.rs:1216-1260)
    Hints hints(const std::vector<Felt> &ch) const {
        Hints h;
        auto seg = [&](int k, bool stop) { return felt_from_u64(stop ? pi_.segments[k].stop_ptr : pi_.segments[k].begin_addr); };
        h.initial_ap = seg(1, false); h.final_ap = seg(1, true); h.initial_pc = seg(0, false); h.final_pc = seg(0, true);
        h.range_check_min = felt_from_u64(pi_.rc_min); h.range_check_max = felt_from_u64(pi_.rc_max);
        h.initial_pedersen_addr = seg(3, false); h.initial_rc_addr = seg(4, false); h.initial_bitwise_addr = seg(6, false);
        // compute_public_memory_quotient (layouts/src/utils.rs:14-46)
        const Felt &z = ch[MEM_Z], &a = ch[MEM_A];
        const uint64_t s = n_ / 16, count = pi_.public_memory.size();
        Felt den = felt_from_u64(1);
        const MemoryEntry *pad = nullptr;
        for (auto &e : pi_.public_memory) {
            den = felt_mul(den, felt_sub(z, felt_add(felt_mul(a, felt_from_canonical(e.value)), felt_from_u64(e.address))));
            if (!pad && e.address == 1) pad = &e;
        }
        if (!pad) throw std::runtime_error("public memory has no entry at address 1");
        den = felt_mul(den, felt_pow(felt_sub(z, felt_add(felt_mul(a, felt_from_canonical(pad->value)), felt_from_u64(1))), s - count));
        h.memory_quotient = felt_mul(felt_pow(z, s), felt_inv(den));
        // compute_diluted_cumulative_value (layouts/src/utils.rs:48-108), 16 bits, spacing 4
        const Felt &dz = ch[AGG_Z], &da = ch[AGG_A];
        const Felt one = felt_from_u64(1), mult = felt_from_u64(16);
        Felt diff_x = felt_from_u64(14), p = felt_add(dz, one), q = one, x = one;
        for (int i = 1; i < 16; ++i) {
            x = felt_add(x, diff_x);
            diff_x = felt_mul(diff_x, mult);
            const Felt xp = felt_mul(x, p), y = felt_add(p, felt_mul(dz, xp));
            q = felt_add(q, felt_add(felt_mul(q, y), felt_mul(x, xp)));
            p = felt_mul(p, y);
        }
        h.diluted_cumulative_value = felt_add(p, felt_mul(q, da));
        return h;
    }

    // ---- domains (mirror of layouts/recursive.py)
    Domain every(uint64_t k) const { return Domain{{}, {{n_ / k, 0}}}; }
    Domain every_except_last(uint64_t k) const { return Domain{{{1, n_ - k}}, {{n_ / k, 0}}}; }
    Domain row_from_end(uint64_t k) const { return Domain{{}, {{1, n_ - k}}}; }

    int table_index(const TableSpec &s) {
        auto it = table_ix_.find(s);
        if (it != table_ix_.end()) return it->second;
        const int ix = (int)specs_.size();
        specs_.push_back(s);
        table_ix_[s] = ix;
        return ix;
    }
    E multiplier(Graph &g, const Domain &d) {
        TableSpec per; per.kind = 2;
        for (auto &f : d.num) if (f.p > 1) per.num.push_back(f);
        for (auto &f : d.den) if (f.p > 1) per.den.push_back(f);
        bool have = false;
        E expr{&g, -1};
        if (!per.num.empty() || !per.den.empty()) { expr = E{&g, g.table((uint32_t)table_index(per))}; have = true; }
        for (auto &f : d.num) if (f.p == 1) {
            const E lin = E{&g, g.x()} - E{&g, g.constant(felt_pow(g_, f.e))};
            expr = have ? expr * lin : lin; have = true;
        }
        for (auto &f : d.den) if (f.p == 1) {
            TableSpec inv; inv.kind = 3; inv.e = f.e;
            const E t{&g, g.table((uint32_t)table_index(inv))};
            expr = have ? expr * t : t; have = true;
        }
        return expr;
    }

    // ---- the composition: sum_i alpha^i numerator_i * multiplier(domain_i), grouped by domain (first-use order)
    int composition(Graph &g, const std::vector<Felt> &ch, const Felt &alpha) {
        if (specs_.empty()) { TableSpec px; px.kind = 0; table_index(px); TableSpec py; py.kind = 1; table_index(py); }
        const Hints h = hints(ch);
        auto T = [&](uint32_t col, uint64_t off) { return E{&g, g.trace(col, (uint32_t)off)}; };
        auto C = [&](uint64_t v) { return E{&g, g.constant_u64(v)}; };
        auto CF = [&](const Felt &f) { return E{&g, g.constant(f)}; };
        auto pow2 = [&](unsigned k) { return CF(felt_pow(felt_from_u64(2), k)); };
        auto flag = [&](int f) { return T(COL_FLAGS, f) - (T(COL_FLAGS, f + 1) + T(COL_FLAGS, f + 1)); };
        auto npc = [&](uint64_t cell, uint64_t cycle = 0) { return T(COL_NPC, CYCLE_HEIGHT * cycle + cell); };
        auto rc = [&](uint64_t cell) { return T(COL_RANGE_CHECK, cell); };
        auto aux = [&](uint64_t cell, uint64_t cycle = 0) { return T(COL_AUXILIARY, CYCLE_HEIGHT * cycle + cell); };
        const E one = C(1), two = C(2), four = C(4), offset_size = C(1ull << 16), half_offset_size = C(1ull << 15);

        struct Group { Domain d; int sum; };
        std::vector<std::pair<std::string, Group>> groups;
        Felt apow = felt_from_u64(1);
        auto add = [&](const std::string &dom_name, const Domain &d, const E &numerator) {
            const E term = numerator * CF(apow);
            auto it = std::find_if(groups.begin(), groups.end(), [&](const std::pair<std::string, Group> &p) { return p.first == dom_name; });
            if (it == groups.end()) groups.push_back({dom_name, Group{d, term.id}});
            else it->second.sum = g.add(it->second.sum, term.id);
            apow = felt_mul(apow, alpha);
        };
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
        ADD(FIRST_ROW, aux(AUX_AP) - CF(h.initial_ap));
        ADD(FIRST_ROW, aux(AUX_FP) - CF(h.initial_ap));
        ADD(FIRST_ROW, npc(NPC_PC) - CF(h.initial_pc));
        ADD(LAST_CYCLE, aux(AUX_AP) - CF(h.final_ap));
        ADD(LAST_CYCLE, aux(AUX_FP) - CF(h.initial_ap));
        ADD(LAST_CYCLE, npc(NPC_PC) - CF(h.final_pc));

        // ---- memory (air.rs:444-497)
        const Domain EVERY_2ND_EXCEPT_LAST = every_except_last(2), SECOND_LAST_ROW = row_from_end(2);
        auto mem = [&](uint64_t cell, uint64_t k = 0) { return T(COL_MEMORY, 2 * k + cell); };
        auto perm_memory = [&](uint64_t k = 0) { return T(COL_MEM_RC_PERMUTATION, 2 * k); };
        auto perm_rc = [&](uint64_t k = 0) { return T(COL_MEM_RC_PERMUTATION, 4 * k + 1); };
        {
            const E z = CF(ch[MEM_Z]), a = CF(ch[MEM_A]);
            const E address_diff = mem(0, 1) - mem(0);
            ADD(FIRST_ROW, (z - (mem(0) + a * mem(1))) * perm_memory() + npc(NPC_PC) + a * npc(NPC_INSTRUCTION) - z);
            ADD(EVERY_2ND_EXCEPT_LAST, (z - (mem(0, 1) + a * mem(1, 1))) * perm_memory(1) - (z - (T(COL_NPC, 2) + a * T(COL_NPC, 3))) * perm_memory());
            ADD(SECOND_LAST_ROW, perm_memory() - CF(h.memory_quotient));
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
            const E z = CF(ch[RC_Z]);
            const E diff = rc_ordered(1) - rc_ordered();
            ADD(FIRST_ROW, (z - rc_ordered()) * perm_rc() + rc(RC_OFF_DST) - z);
            ADD(EVERY_4TH_EXCEPT_LAST, (z - rc_ordered(1)) * perm_rc(1) - (z - T(COL_RANGE_CHECK, 4)) * perm_rc());
            ADD(FOURTH_LAST_ROW, perm_rc() - one);
            ADD(EVERY_4TH_EXCEPT_LAST, diff * diff - diff);
            ADD(FIRST_ROW, rc_ordered() - CF(h.range_check_min));
            ADD(FOURTH_LAST_ROW, rc_ordered() - CF(h.range_check_max));
        }
        // ---- diluted check (air.rs:540-603)
        const Domain EVERY_ROW_EXCEPT_LAST = every_except_last(1), LAST_ROW = row_from_end(1);
        {
            const E z = CF(ch[DC_Z]), za = CF(ch[AGG_Z]), aa = CF(ch[AGG_A]);
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
            ADD(LAST_ROW, agg() - CF(h.diluted_cumulative_value));
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
            ADD(FIRST_ROW, T(COL_NPC, NPC_PEDERSEN_INPUT0_ADDR) - CF(h.initial_pedersen_addr));
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
            ADD(FIRST_ROW, T(COL_NPC, NPC_RANGE_CHECK128_ADDR) - CF(h.initial_rc_addr));
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
            ADD(FIRST_ROW, pool_addr(0) - CF(h.initial_bitwise_addr));
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
        int total = -1;
        for (auto &gr : groups) {
            const E term = E{&g, gr.second.sum} * multiplier(g, gr.second.d);
            total = total < 0 ? term.id : g.add(total, term.id);
        }
        return total;
    }

    // the function a table tabulates, at an arbitrary point
    Felt table_value_at(const TableSpec &s, const Felt &x) const {
        if (s.kind <= 1) {
            if (pedersen_coeffs_[s.kind].empty()) pedersen_coeffs_[s.kind] = interpolate(pedersen_column(s.kind));
            const Felt arg = felt_pow(x, n_ / 2048);
            Felt acc = felt_from_u64(0);
            for (size_t k = pedersen_coeffs_[s.kind].size(); k-- > 0;) acc = felt_add(felt_mul(acc, arg), pedersen_coeffs_[s.kind][k]);
            return acc;
        }
        if (s.kind == 3) return felt_inv(felt_sub(x, felt_pow(g_, s.e)));
        Felt num = felt_from_u64(1), den = felt_from_u64(1);
        for (auto &f : s.num) num = felt_mul(num, felt_sub(felt_pow(x, f.p), felt_pow(g_, f.e)));
        for (auto &f : s.den) den = felt_mul(den, felt_sub(felt_pow(x, f.p), felt_pow(g_, f.e)));
        return felt_mul(num, felt_inv(den));
    }

    // ---- tables on the device
    void build_tables() {
        const uint64_t N = n_ << lb_;
        std::vector<uint64_t> lengths;
        uint64_t off = 0;
        for (auto &s : specs_) {
            uint64_t len = 0;
            if (s.kind <= 1) len = 2048ull << lb_;
            else if (s.kind == 3) len = N;
            else { for (auto &f : s.num) len = std::max(len, N / f.p); for (auto &f : s.den) len = std::max(len, N / f.p); }
            uint32_t ll = 0;
            while ((1ull << ll) < len) ++ll;
            desc_.push_back((uint32_t)off); desc_.push_back(ll);
            lengths.push_back(len);
            off += len;
        }
        tables_.reset(new DeviceBuffer(ctx_, 32 * off));
        const Felt offset = felt_from_u64(offset_), w = root_of_unity(log_n_ + lb_);
        for (size_t t = 0; t < specs_.size(); ++t) {
            const TableSpec &s = specs_[t];
            uint64_t *dst = tables_->u64() + 4ull * desc_[2 * t];
            if (s.kind == 3) {
                const Felt c = felt_pow(g_, s.e);
                ok(ss_inverse_table(ctx_, log_n_ + lb_, offset.data(), c.data(), dst));
                continue;
            }
            std::vector<Felt> host(lengths[t]);
            if (s.kind <= 1) {
                const std::vector<Felt> coeffs = interpolate(pedersen_column(s.kind));
                const Felt step = felt_pow(w, n_ / 2048);
                Felt x = felt_pow(offset, n_ / 2048);
                for (auto &v : host) {
                    Felt acc = felt_from_u64(0);
                    for (size_t k = coeffs.size(); k-- > 0;) acc = felt_add(felt_mul(acc, x), coeffs[k]);
                    v = acc;
                    x = felt_mul(x, step);
                }
            } else {
                Felt x = offset;
                for (auto &v : host) {
                    Felt num = felt_from_u64(1), den = felt_from_u64(1);
                    for (auto &f : s.num) num = felt_mul(num, felt_sub(felt_pow(x, f.p), felt_pow(g_, f.e)));
                    for (auto &f : s.den) den = felt_mul(den, felt_sub(felt_pow(x, f.p), felt_pow(g_, f.e)));
                    v = felt_mul(num, felt_inv(den));
                    x = felt_mul(x, w);
                }
            }
            ok(ss_upload(ctx_, dst, host.data(), host.size() * 32));
        }
        ok(ss_ctx_sync(ctx_));
    }

    ss_ctx *ctx_;
    AirPublicInput pi_;
    uint32_t log_n_, lb_;
    uint64_t offset_, n_;
    Felt g_;
    std::vector<TableSpec> specs_;
    std::map<TableSpec, int> table_ix_;
    std::vector<uint32_t> desc_;
    std::unique_ptr<DeviceBuffer> tables_;
    mutable std::vector<Felt> pedersen_coeffs_[2];
};

}  // namespace

std::unique_ptr<Air> make_recursive_air(ss_ctx *ctx, const AirPublicInput &pi, uint32_t log_n, uint32_t log_blowup, uint64_t lde_offset) {
    return std::unique_ptr<Air>(new RecursiveAir(ctx, pi, log_n, log_blowup, lde_offset));
}
std::vector<uint64_t> recursive_air_tables(const Air &air) {
    const RecursiveAir *r = dynamic_cast<const RecursiveAir *>(&air);
    if (!r) throw std::runtime_error("not a recursive AIR");
    return r->describe_tables();
}

}  // namespace ssh
