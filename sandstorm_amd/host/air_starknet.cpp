// air_starknet.cpp — the `starknet` layout's AIR on the C++ host: the 195 constraints of
// layouts/src/starknet/air.rs:115-2386 in the reference's order, the hints (air.rs:2408-2479) and the nine periodic
// columns (Pedersen points, ECDSA generator doublings, Poseidon round keys).  Mirror of
// sandstorm_amd/layouts/starknet.py, where every piece is documented, derived and pinned to the reference; the two
// are compared on the CPU (tests/test_layout_starknet.py: same tables, same composition on a whole evaluation domain).
#include "air_layout.hpp"
#include "trace_starknet.hpp"

namespace ssh {

namespace {

using namespace layout;

constexpr uint64_t CYCLE_HEIGHT = 16;
enum { COL_FLAGS, COL_PEDERSEN_X, COL_PEDERSEN_Y, COL_PEDERSEN_SUFFIX, COL_PEDERSEN_SLOPE, COL_NPC, COL_MEMORY, COL_RANGE_CHECK, COL_AUXILIARY,
       COL_PERMUTATION };
enum { NPC_PC = 0, NPC_INSTRUCTION = 1, NPC_PUB_MEM_ADDR = 2, NPC_PUB_MEM_VAL = 3, NPC_MEM_OP0_ADDR = 4, NPC_MEM_OP0 = 5, NPC_MEM_DST_ADDR = 8,
       NPC_MEM_DST = 9, NPC_MEM_OP1_ADDR = 12, NPC_MEM_OP1 = 13, NPC_PEDERSEN_INPUT0_ADDR = 6, NPC_PEDERSEN_INPUT1_ADDR = 262,
       NPC_PEDERSEN_OUTPUT_ADDR = 134, NPC_RANGE_CHECK128_ADDR = 70, NPC_ECDSA_PUBKEY_ADDR = 390, NPC_ECDSA_MESSAGE_ADDR = 16774,
       NPC_BITWISE_POOL_ADDR = 198, NPC_BITWISE_X_OR_Y_ADDR = 902, NPC_EC_OP_P_X_ADDR = 8582, NPC_EC_OP_P_Y_ADDR = 4486, NPC_EC_OP_Q_X_ADDR = 12678,
       NPC_EC_OP_Q_Y_ADDR = 2438, NPC_EC_OP_M_ADDR = 10630, NPC_EC_OP_R_X_ADDR = 6534, NPC_EC_OP_R_Y_ADDR = 14726 };
const uint64_t NPC_POSEIDON_ADDRS[6] = {38, 102, 166, 230, 294, 358};
enum { RC_OFF_DST = 0, RC_ORDERED = 2, RC_OFF_OP1 = 4, RC_OFF_OP0 = 8, RC16_COMPONENT = 12 };
enum { AUX_AP = 0, AUX_TMP0 = 2, AUX_OP0_MUL_OP1 = 4, AUX_FP = 8, AUX_TMP1 = 10, AUX_RES = 12 };
enum { F_DST_REG, F_OP0_REG, F_OP1_IMM, F_OP1_FP, F_OP1_AP, F_RES_ADD, F_RES_MUL, F_PC_JUMP_ABS, F_PC_JUMP_REL, F_PC_JNZ, F_AP_ADD, F_AP_ADD1,
       F_OPCODE_CALL, F_OPCODE_RET, F_OPCODE_ASSERT_EQ };
enum { MEM_Z, MEM_A, RC_Z, DC_Z, AGG_Z, AGG_A };
// Ecdsa (air.rs:2691-2783), EcOp (air.rs:2636-2689): cells of column 8
enum { EC_PUBKEY_DOUBLING_X = 1, EC_PUBKEY_DOUBLING_Y = 33, EC_PUBKEY_DOUBLING_SLOPE = 35, EC_PUBKEY_PARTIAL_SUM_X = 17, EC_PUBKEY_PARTIAL_SUM_Y = 49,
       EC_PUBKEY_PARTIAL_SUM_X_DIFF_INV = 51, EC_PUBKEY_PARTIAL_SUM_SLOPE = 19, EC_R_SUFFIX = 9, EC_MESSAGE_SUFFIX = 59, EC_GENERATOR_PARTIAL_SUM_Y = 91,
       EC_GENERATOR_PARTIAL_SUM_X = 27, EC_GENERATOR_PARTIAL_SUM_X_DIFF_INV = 7, EC_GENERATOR_PARTIAL_SUM_SLOPE = 123, EC_R_POINT_SLOPE = 16331,
       EC_R_POINT_X_DIFF_INV = 32715, EC_MESSAGE_INV = 16363, EC_PUBKEY_X_SQUARED = 32747, EC_B_SLOPE = 32763, EC_B_X_DIFF_INV = 32647 };
enum { OP_Q_DOUBLING_X = 41, OP_Q_DOUBLING_Y = 25, OP_Q_DOUBLING_SLOPE = 57, OP_R_PARTIAL_SUM_X = 5, OP_R_PARTIAL_SUM_Y = 37, OP_R_PARTIAL_SUM_SLOPE = 11,
       OP_R_PARTIAL_SUM_X_DIFF_INV = 43, OP_M_SUFFIX = 21, OP_M_BIT251_AND_BIT196_AND_BIT192 = 16371, OP_M_BIT251_AND_BIT196 = 16339 };
// periodic columns = table indices
enum { T_PEDERSEN_X, T_PEDERSEN_Y, T_ECDSA_GENERATOR_X, T_ECDSA_GENERATOR_Y, T_POSEIDON_FULL_KEY0, T_POSEIDON_FULL_KEY1, T_POSEIDON_FULL_KEY2,
       T_POSEIDON_PARTIAL_KEY0, T_POSEIDON_PARTIAL_KEY1, NUM_PERIODIC };

struct Hints {
    Felt initial_ap, initial_pc, final_ap, final_pc, range_check_min, range_check_max, memory_quotient, diluted_cumulative_value, initial_pedersen_addr,
        initial_rc_addr, initial_ecdsa_addr, initial_bitwise_addr, initial_ec_op_addr, initial_poseidon_addr;
};

Felt canonical_limbs(uint64_t l0, uint64_t l1, uint64_t l2, uint64_t l3) {
    Felt c{l0, l1, l2, l3};
    return felt_from_canonical(c);
}
// builtins/src/utils.rs:151-158
const Felt &curve_beta() {
    static const Felt v = canonical_limbs(0xf4cdfcb99cee9e89ull, 0x609ad26c15c915c1ull, 0x150e596d72f7a8c5ull, 0x06f21413efbe40deull);
    return v;
}
Pt curve_generator() {
    return Pt{canonical_limbs(0x3d723d8bc943cfcaull, 0xdeacfd9b0d1819e0ull, 0x7beced415a40f0c7ull, 0x01ef15c18599971bull),
              canonical_limbs(0x2873000c36e8dc1full, 0xde53ecd11abe43a3ull, 0xb7be4801df46ec62ull, 0x005668060aa49730ull)};
}

// ---- SHA-256 (FIPS 180-4), for the Hades round constants only
struct Sha256 {
    static uint32_t rotr(uint32_t x, int k) { return (x >> k) | (x << (32 - k)); }
    static std::array<uint8_t, 32> hash(const std::string &msg) {
        static const uint32_t K[64] = {
            0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
            0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
            0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
            0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
            0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
            0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
        uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
        std::vector<uint8_t> m(msg.begin(), msg.end());
        const uint64_t bits = 8ull * m.size();
        m.push_back(0x80);
        while (m.size() % 64 != 56) m.push_back(0);
        for (int k = 7; k >= 0; --k) m.push_back((uint8_t)(bits >> (8 * k)));
        for (size_t off = 0; off < m.size(); off += 64) {
            uint32_t w[64];
            for (int i = 0; i < 16; ++i) w[i] = (uint32_t)m[off + 4 * i] << 24 | (uint32_t)m[off + 4 * i + 1] << 16 | (uint32_t)m[off + 4 * i + 2] << 8 | m[off + 4 * i + 3];
            for (int i = 16; i < 64; ++i) {
                const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
                w[i] = w[i - 16] + s0 + w[i - 7] + s1;
            }
            uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
            for (int i = 0; i < 64; ++i) {
                const uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
                const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
                hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
            }
            h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
        }
        std::array<uint8_t, 32> out;
        for (int i = 0; i < 8; ++i) for (int k = 0; k < 4; ++k) out[4 * i + k] = (uint8_t)(h[i] >> (24 - 8 * k));
        return out;
    }
};

// StarkWare's Hades round constants (builtins/src/poseidon/params.rs ROUND_KEYS): sha256("Hades" + index), big-endian, mod p
Felt hades_constant(unsigned index) {
    const std::array<uint8_t, 32> d = Sha256::hash("Hades" + std::to_string(index));
    Felt v{};
    for (int i = 0; i < 32; ++i) v[(31 - i) / 8] |= (uint64_t)d[i] << (8 * ((31 - i) % 8));
    const Felt p{1, 0, 0, 0x0800000000000011ull};
    auto geq = [&](const Felt &a, const Felt &b) { for (int k = 3; k >= 0; --k) if (a[k] != b[k]) return a[k] > b[k]; return true; };
    while (geq(v, p)) {
        unsigned __int128 borrow = 0;
        for (int k = 0; k < 4; ++k) {
            const unsigned __int128 t = (unsigned __int128)v[k] - p[k] - (uint64_t)borrow;
            v[k] = (uint64_t)t;
            borrow = (t >> 64) & 1;
        }
    }
    return felt_from_canonical(v);
}

struct PoseidonKeys {
    std::vector<std::array<Felt, 3>> round;         // the 91 plain round keys
    std::vector<Felt> partial;                      // 80 keys of the AIR's partial-round recurrence
    Felt margin_full_to_partial[3], margin_partial_to_full[3];
    std::vector<Felt> full[3];                      // periodic values, 8 per state element
};
std::array<Felt, 3> mds(const std::array<Felt, 3> &s) {
    return {felt_add(felt_add(felt_add(felt_add(s[0], s[0]), s[0]), s[1]), s[2]), felt_add(felt_sub(s[0], s[1]), s[2]),
            felt_sub(felt_add(s[0], s[1]), felt_add(s[2], s[2]))};
}
Felt cube(const Felt &v) { return felt_mul(felt_mul(v, v), v); }
Felt times(const Felt &v, uint64_t k) { return felt_mul(v, felt_from_u64(k)); }

// poseidon_air_keys of layouts/starknet.py: the AIR's keys are what is left of its linear recurrences on the permutation of
// (0, 0, 0) (poseidon/mod.rs:45-98, air.rs:2040-2160)
const PoseidonKeys &poseidon_keys() {
    static const PoseidonKeys keys = [] {
        PoseidonKeys k;
        for (unsigned i = 0; i < 91; ++i) k.round.push_back({hades_constant(3 * i), hades_constant(3 * i + 1), hades_constant(3 * i + 2)});
        std::array<Felt, 3> st{felt_from_u64(0), felt_from_u64(0), felt_from_u64(0)};
        std::vector<std::array<Felt, 3>> full;
        std::vector<Felt> s;
        unsigned r = 0;
        for (int phase = 0; phase < 3; ++phase)
            for (int i = 0; i < (phase == 1 ? 83 : 4); ++i, ++r) {
                for (int j = 0; j < 3; ++j) st[j] = felt_add(st[j], k.round[r][j]);
                if (phase == 1) { s.push_back(st[2]); st[2] = cube(st[2]); }
                else { full.push_back(st); for (auto &v : st) v = cube(v); }
                st = mds(st);
            }
        std::vector<Felt> c;
        for (auto &v : s) c.push_back(cube(v));
        const Felt f3[3] = {cube(full[3][0]), cube(full[3][1]), cube(full[3][2])};
        auto sub = felt_sub; auto add = felt_add;
        for (int i = 0; i < 80; ++i)
            k.partial.push_back(sub(s[i + 3], sub(add(add(add(times(c[i], 8), times(s[i + 1], 4)), times(c[i + 1], 6)), times(s[i + 2], 2)), times(c[i + 2], 2))));
        k.margin_full_to_partial[0] = k.round[4][2];
        k.margin_full_to_partial[1] = sub(s[1], sub(add(times(f3[2], 10), times(s[0], 4)), add(times(f3[1], 4), times(c[0], 2))));
        k.margin_full_to_partial[2] = sub(s[2], sub(add(add(add(times(f3[2], 8), times(s[0], 4)), times(c[0], 6)), times(s[1], 2)), times(c[1], 2)));
        k.margin_partial_to_full[0] = sub(full[4][0], add(add(add(add(times(c[80], 16), times(s[81], 8)), times(c[81], 16)), times(s[82], 6)), c[82]));
        k.margin_partial_to_full[1] = sub(full[4][1], add(add(times(c[81], 4), times(s[82], 2)), c[82]));
        k.margin_partial_to_full[2] = sub(full[4][2], sub(add(add(add(times(c[80], 8), times(s[81], 4)), times(c[81], 6)), times(s[82], 2)), times(c[82], 2)));
        const Felt zero = felt_from_u64(0);
        for (int j = 0; j < 3; ++j) k.full[j] = {k.round[1][j], k.round[2][j], k.round[3][j], zero, k.round[88][j], k.round[89][j], k.round[90][j], zero};
        return k;
    }();
    return keys;
}

}  // namespace

const std::vector<std::array<Felt, 3>> &poseidon_round_keys() { return poseidon_keys().round; }
void starknet_curve(Felt &generator_x, Felt &generator_y, Felt &beta) {
    const Pt g = curve_generator();
    generator_x = g.x; generator_y = g.y; beta = curve_beta();
}

namespace {

class StarknetAir : public LayoutAir {
public:
    StarknetAir(ss_ctx *ctx, const AirPublicInput &pi, uint32_t log_n, uint32_t lb, uint64_t lde_offset) : LayoutAir(ctx, pi, log_n, lb, lde_offset) {
        if (pi.layout != "starknet") throw std::runtime_error("the public input is not of the starknet layout");
        if (log_n < 15) throw std::runtime_error("the starknet layout needs a trace of at least 2^15 rows (one ECDSA instance)");
        for (int k : {3, 4, 5, 6, 7, 8}) if (!pi.segments[k].present) throw std::runtime_error("the starknet layout requires every builtin segment");
        name = "starknet"; num_base_columns = 9; num_extension_columns = 1; num_challenges = 6;
        finish_construction();
    }

private:
    // ---- hints (air.rs:2408-2479)
    Hints hints(const std::vector<Felt> &ch) const {
        Hints h;
        auto seg = [&](int k, bool stop) { return felt_from_u64(stop ? pi_.segments[k].stop_ptr : pi_.segments[k].begin_addr); };
        h.initial_ap = seg(1, false); h.final_ap = seg(1, true); h.initial_pc = seg(0, false); h.final_pc = seg(0, true);
        h.range_check_min = felt_from_u64(pi_.rc_min); h.range_check_max = felt_from_u64(pi_.rc_max);
        h.initial_pedersen_addr = seg(3, false); h.initial_rc_addr = seg(4, false); h.initial_ecdsa_addr = seg(5, false);
        h.initial_bitwise_addr = seg(6, false); h.initial_ec_op_addr = seg(7, false); h.initial_poseidon_addr = seg(8, false);
        h.memory_quotient = public_memory_quotient(pi_, ch[MEM_Z], ch[MEM_A], n_, 8);
        h.diluted_cumulative_value = diluted_cumulative_value(ch[AGG_Z], ch[AGG_A]);
        return h;
    }

    // ---- periodic columns (air.rs:47-104)
    size_t num_periodic_columns() const override { return NUM_PERIODIC; }
    uint64_t column_period(size_t c) const override { return c <= T_PEDERSEN_Y ? 512 : c <= T_ECDSA_GENERATOR_Y ? 32768 : 512; }
    std::vector<Felt> column_values(size_t c) const override {
        if (c <= T_PEDERSEN_Y) return pedersen_column((int)c);
        if (c <= T_ECDSA_GENERATOR_Y) {                      // 2^i G for i <= 250, the last one repeated (ecdsa/mod.rs:103, 157-190)
            std::vector<Felt> out;
            Pt acc = curve_generator();
            for (int i = 0; i < 256; ++i) { out.push_back(c == T_ECDSA_GENERATOR_X ? acc.x : acc.y); if (i < 250) acc = ec_double(acc); }
            return out;
        }
        const PoseidonKeys &k = poseidon_keys();
        if (c <= T_POSEIDON_FULL_KEY2) return k.full[c - T_POSEIDON_FULL_KEY0];
        const Felt zero = felt_from_u64(0);
        if (c == T_POSEIDON_PARTIAL_KEY0) { std::vector<Felt> v(k.partial.begin(), k.partial.begin() + 61); v.resize(64, zero); return v; }
        std::vector<Felt> v(k.partial.begin() + 61, k.partial.begin() + 80);
        v.resize(32, zero);
        return v;
    }

    // ---- the composition, constraint by constraint as layouts/starknet.py::constraints
    int composition(Graph &g, const std::vector<Felt> &ch, const Felt &alpha) override {
        const Hints h = hints(ch);
        auto T = [&](uint32_t col, uint64_t off) { return E{&g, g.trace(col, (uint32_t)off)}; };
        auto C = [&](uint64_t v) { return E{&g, g.constant_u64(v)}; };
        auto CF = [&](const Felt &f) { return E{&g, g.constant(f)}; };                 // structural constants only
        // per-proof values: interned by symbol, so that the program's shape is the same for every statement (air_program.hpp)
#define HINT(name) E{&g, g.runtime_constant(Graph::sym("hint." #name), h.name)}
#define CHAL(index) E{&g, g.runtime_constant(Graph::sym("challenge", index), ch[index])}
        auto NEG = [&](uint64_t v) { return CF(felt_neg(felt_from_u64(v))); };
        auto pow2 = [&](unsigned k) { return CF(felt_pow(felt_from_u64(2), k)); };
        auto flag = [&](int f) { return T(COL_FLAGS, f) - (T(COL_FLAGS, f + 1) + T(COL_FLAGS, f + 1)); };
        auto npc = [&](uint64_t cell, uint64_t cycle = 0) { return T(COL_NPC, CYCLE_HEIGHT * cycle + cell); };
        auto npc_at = [&](uint64_t off) { return T(COL_NPC, off); };
        auto rc = [&](uint64_t cell) { return T(COL_RANGE_CHECK, cell); };
        auto aux = [&](uint64_t cell, uint64_t cycle = 0) { return T(COL_AUXILIARY, CYCLE_HEIGHT * cycle + cell); };
        auto c7 = [&](uint64_t off) { return T(COL_RANGE_CHECK, off); };
        auto c8 = [&](uint64_t off) { return T(COL_AUXILIARY, off); };
        const E one = C(1), two = C(2), four = C(4), offset_size = C(1ull << 16), half_offset_size = C(1ull << 15);
        using Cell = std::function<E(uint64_t)>;

        Composer composer(*this, g, alpha);
        auto add = [&](const std::string &dom_name, const Domain &d, const E &numerator) { composer.add(dom_name, d, numerator); };
        const Domain ALL_CYCLES = every(16), ALL_CYCLES_EXCEPT_LAST = every_except_last(16);
        const Domain FLAG_ROWS{{{n_ / 16, 15 * n_ / 16}}, {{n_, 0}}}, FLAG_ZERO_ROWS{{}, {{n_ / 16, 15 * n_ / 16}}};
        const Domain FIRST_ROW{{}, {{1, 0}}}, LAST_CYCLE = row_from_end(16);
#define ADD(dom, expr) add(#dom, dom, expr)

        // ---- cpu (air.rs:128-557)
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

        // ---- memory (air.rs:560-600)
        const Domain EVERY_2ND_EXCEPT_LAST = every_except_last(2), SECOND_LAST_ROW = row_from_end(2), EVERY_8 = every(8);
        {
            auto mem = [&](uint64_t cell, uint64_t k = 0) { return T(COL_MEMORY, 2 * k + cell); };
            auto perm = [&](uint64_t k = 0) { return T(COL_PERMUTATION, 2 * k); };
            const E z = CHAL(MEM_Z), a = CHAL(MEM_A);
            const E diff = mem(0, 1) - mem(0);
            ADD(FIRST_ROW, (z - (mem(0) + a * mem(1))) * perm() + npc(NPC_PC) + a * npc(NPC_INSTRUCTION) - z);
            ADD(EVERY_2ND_EXCEPT_LAST, (z - (mem(0, 1) + a * mem(1, 1))) * perm(1) - (z - (npc_at(2) + a * npc_at(3))) * perm());
            ADD(SECOND_LAST_ROW, perm() - HINT(memory_quotient));
            ADD(EVERY_2ND_EXCEPT_LAST, diff * diff - diff);
            ADD(EVERY_2ND_EXCEPT_LAST, (diff - one) * (mem(1) - mem(1, 1)));
            ADD(FIRST_ROW, mem(0) - one);
            ADD(EVERY_8, npc_at(NPC_PUB_MEM_ADDR));
            ADD(EVERY_8, npc_at(NPC_PUB_MEM_VAL));
        }
        // ---- 16-bit range check (air.rs:602-630)
        const Domain EVERY_4TH_EXCEPT_LAST = every_except_last(4), FOURTH_LAST_ROW = row_from_end(4);
        {
            auto ordered = [&](uint64_t k = 0) { return c7(4 * k + RC_ORDERED); };
            auto perm = [&](uint64_t k = 0) { return T(COL_PERMUTATION, 4 * k + 1); };
            const E z = CHAL(RC_Z);
            const E diff = ordered(1) - ordered();
            ADD(FIRST_ROW, (z - ordered()) * perm() + rc(RC_OFF_DST) - z);
            ADD(EVERY_4TH_EXCEPT_LAST, (z - ordered(1)) * perm(1) - (z - c7(4)) * perm());
            ADD(FOURTH_LAST_ROW, perm() - one);
            ADD(EVERY_4TH_EXCEPT_LAST, diff * diff - diff);
            ADD(FIRST_ROW, ordered() - HINT(range_check_min));
            ADD(FOURTH_LAST_ROW, ordered() - HINT(range_check_max));
        }
        // ---- diluted check (air.rs:632-690): every 8 rows of columns 7 and 9
        const Domain EVERY_8_EXCEPT_LAST = every_except_last(8), EIGHTH_LAST_ROW = row_from_end(8);
        {
            const E z = CHAL(DC_Z), za = CHAL(AGG_Z), aa = CHAL(AGG_A);
            auto un = [&](uint64_t k = 0) { return c7(8 * k + 1); };
            auto od = [&](uint64_t k = 0) { return c7(8 * k + 5); };
            auto perm = [&](uint64_t k = 0) { return T(COL_PERMUTATION, 8 * k + 7); };
            auto agg = [&](uint64_t k = 0) { return T(COL_PERMUTATION, 8 * k + 3); };
            const E diff = od(1) - od();
            ADD(FIRST_ROW, (z - od()) * perm() + un() - z);
            ADD(EVERY_8_EXCEPT_LAST, (z - od(1)) * perm(1) - (z - un(1)) * perm());
            ADD(EIGHTH_LAST_ROW, perm() - one);
            ADD(FIRST_ROW, agg() - one);
            ADD(FIRST_ROW, od() - C(0));
            ADD(EVERY_8_EXCEPT_LAST, agg(1) - (agg() * (one + za * diff) + aa * diff * diff));
            ADD(EIGHTH_LAST_ROW, agg() - HINT(diluted_cumulative_value));
        }

        // ---- shared shapes (layouts/starknet.py::_bit_unpacking, _subset_sum, _doubling)
        auto bit_unpacking = [&](const std::string &dn, const Domain &d, const Cell &suffix, const E &b_all, const E &b_251_196) {
            auto bit = [&](uint64_t k) { return suffix(k) - (suffix(k + 1) + suffix(k + 1)); };
            add(dn, d, b_all * bit(0));
            add(dn, d, b_all * (suffix(1) - suffix(192) * pow2(191)));
            add(dn, d, b_all - b_251_196 * bit(192));
            add(dn, d, b_251_196 * (suffix(193) - suffix(196) * pow2(3)));
            add(dn, d, b_251_196 - bit(251) * bit(196));
            add(dn, d, bit(251) * (suffix(197) - suffix(251) * pow2(54)));
        };
        auto subset_sum = [&](const std::string &dn, const Domain &d, const E &bit, const Cell &sx, const Cell &sy, const Cell &slope, const E &px, const E &py,
                              const Cell *x_diff_inv) {
            add(dn, d, bit * (sy(0) - py) - slope(0) * (sx(0) - px));
            add(dn, d, slope(0) * slope(0) - bit * (sx(0) + px + sx(1)));
            add(dn, d, bit * (sy(0) + sy(1)) - slope(0) * (sx(0) - sx(1)));
            if (x_diff_inv) add(dn, d, (*x_diff_inv)(0) * (sx(0) - px) - one);
            add(dn, d, (one - bit) * (sx(1) - sx(0)));
            add(dn, d, (one - bit) * (sy(1) - sy(0)));
        };
        auto doubling = [&](const std::string &dn, const Domain &d, const Cell &x, const Cell &y, const Cell &slope) {
            const E x2 = x(0) * x(0);
            add(dn, d, x2 + x2 + x2 + one - (y(0) + y(0)) * slope(0));
            add(dn, d, slope(0) * slope(0) - (x(0) + x(0) + x(1)));
            add(dn, d, y(0) + y(1) - slope(0) * (x(0) - x(1)));
        };
        auto cell8 = [&](uint64_t step, uint64_t cell) { return Cell([=](uint64_t k) { return c8(step * k + cell); }); };

        // ---- Pedersen builtin (air.rs:692-1025): one step per row of columns 1-4
        const Domain EVERY_256 = every(256), EVERY_512 = every(512), EVERY_512_EXCEPT_LAST = every_except_last(512);
        const Domain EVERY_256_EXCEPT_LAST = every_except_last(256);
        {
            const Cell suffix = [&](uint64_t k) { return T(COL_PEDERSEN_SUFFIX, k); }, slope = [&](uint64_t k) { return T(COL_PEDERSEN_SLOPE, k); };
            const Cell sum_x = [&](uint64_t k) { return T(COL_PEDERSEN_X, k); }, sum_y = [&](uint64_t k) { return T(COL_PEDERSEN_Y, k); };
            const E b0 = suffix(0) - (suffix(1) + suffix(1));
            const Domain PEDERSEN_TRANSITION{{F(256, 255, 256)}, {F(1)}}, PEDERSEN_STEP_252{{}, {F(256, 63, 64)}}, PEDERSEN_STEP_255{{}, {F(256, 255, 256)}};
            const Domain PEDERSEN_HASH_START{{F(512, 1, 2)}, {F(256)}};
            bit_unpacking("EVERY_256", EVERY_256, suffix, c8(71), T(COL_PEDERSEN_SLOPE, 255));
            ADD(PEDERSEN_TRANSITION, b0 * (b0 - one));
            ADD(PEDERSEN_STEP_252, suffix(0));
            ADD(PEDERSEN_STEP_255, suffix(0));
            subset_sum("PEDERSEN_TRANSITION", PEDERSEN_TRANSITION, b0, sum_x, sum_y, slope, column(g, T_PEDERSEN_X), column(g, T_PEDERSEN_Y), nullptr);
            ADD(PEDERSEN_HASH_START, sum_x(256) - sum_x(255));
            ADD(PEDERSEN_HASH_START, sum_y(256) - sum_y(255));
            ADD(EVERY_512, sum_x(0) - CF(pedersen_coord(0, 0)));
            ADD(EVERY_512, sum_y(0) - CF(pedersen_coord(0, 1)));
            ADD(EVERY_512, npc_at(NPC_PEDERSEN_INPUT0_ADDR + 1) - suffix(0));
            ADD(EVERY_512_EXCEPT_LAST, npc_at(512 + NPC_PEDERSEN_INPUT0_ADDR) - (npc_at(NPC_PEDERSEN_OUTPUT_ADDR) + one));
            ADD(FIRST_ROW, npc_at(NPC_PEDERSEN_INPUT0_ADDR) - HINT(initial_pedersen_addr));
            ADD(EVERY_512, npc_at(NPC_PEDERSEN_INPUT1_ADDR + 1) - suffix(256));
            ADD(EVERY_512, npc_at(NPC_PEDERSEN_INPUT1_ADDR) - (npc_at(NPC_PEDERSEN_INPUT0_ADDR) + one));
            ADD(EVERY_512, npc_at(NPC_PEDERSEN_OUTPUT_ADDR + 1) - sum_x(511));
            ADD(EVERY_512, npc_at(NPC_PEDERSEN_OUTPUT_ADDR) - (npc_at(NPC_PEDERSEN_INPUT1_ADDR) + one));
        }
        // ---- range-check builtin (air.rs:1027-1040)
        {
            E value = c7(RC16_COMPONENT);
            for (uint64_t k = 1; k < 8; ++k) value = value * offset_size + c7(32 * k + RC16_COMPONENT);
            ADD(EVERY_256, value - npc_at(NPC_RANGE_CHECK128_ADDR + 1));
            ADD(EVERY_256_EXCEPT_LAST, npc_at(256 + NPC_RANGE_CHECK128_ADDR) - (npc_at(NPC_RANGE_CHECK128_ADDR) + one));
            ADD(FIRST_ROW, npc_at(NPC_RANGE_CHECK128_ADDR) - HINT(initial_rc_addr));
        }
        // ---- ECDSA builtin (air.rs:1042-1503)
        const Domain ALL_ECDSA = every(32768), ALL_ECDSA_EXCEPT_LAST = every_except_last(32768), ALL_EC_OP = every(16384),
                     ALL_EC_OP_EXCEPT_LAST = every_except_last(16384);
        const Domain EC_OP_TRANSITION{{F(16384, 255, 256)}, {F(64)}}, ECDSA_TRANSITION{{F(32768, 255, 256)}, {F(128)}};
        const Domain ECDSA_STEP_251{{}, {F(32768, 251, 256)}}, ECDSA_STEP_255{{}, {F(32768, 255, 256)}};
        const Domain EC_OP_STEP_251{{}, {F(16384, 251, 256)}}, EC_OP_STEP_252{{}, {F(16384, 63, 64)}}, EC_OP_STEP_255{{}, {F(16384, 255, 256)}};
        const E shift_x = CF(pedersen_coord(0, 0)), shift_y = CF(pedersen_coord(0, 1));
        {
            const Cell dx = cell8(64, EC_PUBKEY_DOUBLING_X), dy = cell8(64, EC_PUBKEY_DOUBLING_Y), dslope = cell8(64, EC_PUBKEY_DOUBLING_SLOPE);
            const Cell kx = cell8(64, EC_PUBKEY_PARTIAL_SUM_X), ky = cell8(64, EC_PUBKEY_PARTIAL_SUM_Y), kslope = cell8(64, EC_PUBKEY_PARTIAL_SUM_SLOPE),
                       kinv = cell8(64, EC_PUBKEY_PARTIAL_SUM_X_DIFF_INV), rsuffix = cell8(64, EC_R_SUFFIX);
            const Cell gx = cell8(128, EC_GENERATOR_PARTIAL_SUM_X), gy = cell8(128, EC_GENERATOR_PARTIAL_SUM_Y), gslope = cell8(128, EC_GENERATOR_PARTIAL_SUM_SLOPE),
                       ginv = cell8(128, EC_GENERATOR_PARTIAL_SUM_X_DIFF_INV), msuffix = cell8(128, EC_MESSAGE_SUFFIX);
            const E gen_b0 = msuffix(0) - (msuffix(1) + msuffix(1)), key_b0 = rsuffix(0) - (rsuffix(1) + rsuffix(1));
            doubling("EC_OP_TRANSITION", EC_OP_TRANSITION, dx, dy, dslope);
            ADD(ECDSA_TRANSITION, gen_b0 * (gen_b0 - one));
            ADD(ECDSA_STEP_251, msuffix(0));
            ADD(ECDSA_STEP_255, msuffix(0));
            subset_sum("ECDSA_TRANSITION", ECDSA_TRANSITION, gen_b0, gx, gy, gslope, column(g, T_ECDSA_GENERATOR_X), column(g, T_ECDSA_GENERATOR_Y), &ginv);
            ADD(EC_OP_TRANSITION, key_b0 * (key_b0 - one));
            ADD(EC_OP_STEP_251, rsuffix(0));
            ADD(EC_OP_STEP_255, rsuffix(0));
            subset_sum("EC_OP_TRANSITION", EC_OP_TRANSITION, key_b0, kx, ky, kslope, dx(0), dy(0), &kinv);
            const E b_slope = c8(EC_B_SLOPE), b_inv = c8(EC_B_X_DIFF_INV), r_slope = c8(EC_R_POINT_SLOPE), r_inv_x = c8(EC_R_POINT_X_DIFF_INV);
            ADD(ALL_ECDSA, gx(0) - shift_x);
            ADD(ALL_ECDSA, gy(0) + shift_y);
            ADD(ALL_EC_OP, kx(0) - shift_x);
            ADD(ALL_EC_OP, ky(0) - shift_y);
            ADD(ALL_ECDSA, gy(255) - (ky(255) + b_slope * (gx(255) - kx(255))));
            ADD(ALL_ECDSA, b_slope * b_slope - (gx(255) + kx(255) + dx(256)));
            ADD(ALL_ECDSA, gy(255) + dy(256) - b_slope * (gx(255) - dx(256)));
            ADD(ALL_ECDSA, b_inv * (gx(255) - kx(255)) - one);
            ADD(ALL_ECDSA, ky(511) + shift_y - r_slope * (kx(511) - shift_x));
            ADD(ALL_ECDSA, r_slope * r_slope - (kx(511) + shift_x + rsuffix(0)));
            ADD(ALL_ECDSA, r_inv_x * (kx(511) - shift_x) - one);
            ADD(ALL_ECDSA, msuffix(0) * c8(EC_MESSAGE_INV) - one);
            ADD(ALL_EC_OP, rsuffix(0) * dslope(255) - one);
            ADD(ALL_ECDSA, c8(EC_PUBKEY_X_SQUARED) - dx(0) * dx(0));
            ADD(ALL_ECDSA, dy(0) * dy(0) - (dx(0) * c8(EC_PUBKEY_X_SQUARED) + dx(0) * one + CF(curve_beta())));
            ADD(FIRST_ROW, npc_at(NPC_ECDSA_PUBKEY_ADDR) - HINT(initial_ecdsa_addr));
            ADD(ALL_ECDSA, npc_at(NPC_ECDSA_MESSAGE_ADDR) - (npc_at(NPC_ECDSA_PUBKEY_ADDR) + one));
            ADD(ALL_ECDSA_EXCEPT_LAST, npc_at(32768 + NPC_ECDSA_PUBKEY_ADDR) - (npc_at(NPC_ECDSA_MESSAGE_ADDR) + one));
            ADD(ALL_ECDSA, npc_at(NPC_ECDSA_MESSAGE_ADDR + 1) - msuffix(0));
            ADD(ALL_ECDSA, npc_at(NPC_ECDSA_PUBKEY_ADDR + 1) - dx(0));
        }
        // ---- bitwise builtin (air.rs:1505-1660)
        {
            auto cell = [](unsigned chunk, unsigned stream) { return 16ull * (4 * chunk + stream) + 1; };
            auto pool_addr = [&](uint64_t k) { return npc_at(256 * k + NPC_BITWISE_POOL_ADDR); };
            auto pool_val = [&](uint64_t k) { return npc_at(256 * k + NPC_BITWISE_POOL_ADDR + 1); };
            const Domain ALL_BITWISE = every(1024), ALL_BITWISE_EXCEPT_LAST = every_except_last(1024);
            const Domain BITWISE_TRANSITION{{F(1024, 3, 4)}, {F(256)}};
            Domain EVERY_16_BIT_SEGMENT;
            for (uint64_t k = 1; k < 16; ++k) EVERY_16_BIT_SEGMENT.den.push_back(F(1024, k, 64));
            EVERY_16_BIT_SEGMENT.den.push_back(F(1024));
            E sum_var = c7(cell(0, 0));
            for (unsigned chunk = 0; chunk < 4; ++chunk)
                for (unsigned stream = 0; stream < 4; ++stream) {
                    if (chunk == 0 && stream == 0) continue;
                    sum_var = sum_var + c7(cell(chunk, stream)) * pow2(64 * chunk + stream);
                }
            ADD(FIRST_ROW, pool_addr(0) - HINT(initial_bitwise_addr));
            ADD(BITWISE_TRANSITION, pool_addr(1) - (pool_addr(0) + one));
            ADD(ALL_BITWISE, npc_at(NPC_BITWISE_X_OR_Y_ADDR) - (pool_addr(3) + one));
            ADD(ALL_BITWISE_EXCEPT_LAST, pool_addr(4) - (npc_at(NPC_BITWISE_X_OR_Y_ADDR) + one));
            ADD(EVERY_256, sum_var - pool_val(0));
            ADD(ALL_BITWISE, npc_at(NPC_BITWISE_X_OR_Y_ADDR + 1) - (pool_val(2) + pool_val(3)));
            ADD(EVERY_16_BIT_SEGMENT, c7(1) + c7(257) - (c7(769) + c7(513) + c7(513)));
            const uint64_t shifted[4] = {9, 521, 265, 777};
            for (unsigned k = 0; k < 4; ++k) ADD(ALL_BITWISE, (c7(512 + cell(3, k)) + c7(768 + cell(3, k))) * pow2(k == 3 ? 8 : 4) - c7(shifted[k]));
        }
        // ---- EC-op builtin (air.rs:1662-1866)
        {
            const Cell qx = cell8(64, OP_Q_DOUBLING_X), qy = cell8(64, OP_Q_DOUBLING_Y), qslope = cell8(64, OP_Q_DOUBLING_SLOPE);
            const Cell rx = cell8(64, OP_R_PARTIAL_SUM_X), ry = cell8(64, OP_R_PARTIAL_SUM_Y), rslope = cell8(64, OP_R_PARTIAL_SUM_SLOPE),
                       rinv = cell8(64, OP_R_PARTIAL_SUM_X_DIFF_INV), msuffix = cell8(64, OP_M_SUFFIX);
            const E b0 = msuffix(0) - (msuffix(1) + msuffix(1));
            ADD(FIRST_ROW, npc_at(NPC_EC_OP_P_X_ADDR) - HINT(initial_ec_op_addr));
            ADD(ALL_EC_OP_EXCEPT_LAST, npc_at(16384 + NPC_EC_OP_P_X_ADDR) - (npc_at(NPC_EC_OP_P_X_ADDR) + C(7)));
            ADD(ALL_EC_OP, npc_at(NPC_EC_OP_P_Y_ADDR) - (npc_at(NPC_EC_OP_P_X_ADDR) + one));
            ADD(ALL_EC_OP, npc_at(NPC_EC_OP_Q_X_ADDR) - (npc_at(NPC_EC_OP_P_Y_ADDR) + one));
            ADD(ALL_EC_OP, npc_at(NPC_EC_OP_Q_Y_ADDR) - (npc_at(NPC_EC_OP_Q_X_ADDR) + one));
            ADD(ALL_EC_OP, npc_at(NPC_EC_OP_M_ADDR) - (npc_at(NPC_EC_OP_Q_Y_ADDR) + one));
            ADD(ALL_EC_OP, npc_at(NPC_EC_OP_R_X_ADDR) - (npc_at(NPC_EC_OP_M_ADDR) + one));
            ADD(ALL_EC_OP, npc_at(NPC_EC_OP_R_Y_ADDR) - (npc_at(NPC_EC_OP_R_X_ADDR) + one));
            doubling("EC_OP_TRANSITION", EC_OP_TRANSITION, qx, qy, qslope);
            ADD(ALL_EC_OP, npc_at(NPC_EC_OP_Q_X_ADDR + 1) - qx(0));
            ADD(ALL_EC_OP, npc_at(NPC_EC_OP_Q_Y_ADDR + 1) - qy(0));
            bit_unpacking("ALL_EC_OP", ALL_EC_OP, msuffix, c8(OP_M_BIT251_AND_BIT196_AND_BIT192), c8(OP_M_BIT251_AND_BIT196));
            ADD(EC_OP_TRANSITION, b0 * (b0 - one));
            ADD(EC_OP_STEP_252, msuffix(0));
            ADD(EC_OP_STEP_255, msuffix(0));
            subset_sum("EC_OP_TRANSITION", EC_OP_TRANSITION, b0, rx, ry, rslope, qx(0), qy(0), &rinv);
            ADD(ALL_EC_OP, msuffix(0) - npc_at(NPC_EC_OP_M_ADDR + 1));
            ADD(ALL_EC_OP, npc_at(NPC_EC_OP_P_X_ADDR + 1) - rx(0));
            ADD(ALL_EC_OP, npc_at(NPC_EC_OP_P_Y_ADDR + 1) - ry(0));
            ADD(ALL_EC_OP, npc_at(NPC_EC_OP_R_X_ADDR + 1) - rx(255));
            ADD(ALL_EC_OP, npc_at(NPC_EC_OP_R_Y_ADDR + 1) - ry(255));
        }
        // ---- Poseidon builtin (air.rs:1868-2186)
        {
            const PoseidonKeys &keys = poseidon_keys();
            const uint64_t FULL[3][2] = {{53, 29}, {13, 61}, {45, 3}};         // state, state squared: column 8, step 64
            auto full = [&](int j, uint64_t k = 0) { return c8(64 * k + FULL[j][0]); };
            auto full_sq = [&](int j, uint64_t k = 0) { return c8(64 * k + FULL[j][1]); };
            auto part = [&](int which, uint64_t k = 0) { return which == 0 ? c7(8 * k + 3) : c8(16 * k + 6); };
            auto part_sq = [&](int which, uint64_t k = 0) { return which == 0 ? c7(8 * k + 7) : c8(16 * k + 14); };
            auto cubed = [&](int j, uint64_t k) { return full(j, k) * full_sq(j, k); };
            auto pcubed = [&](int which, uint64_t k) { return part(which, k) * part_sq(which, k); };
            auto addr = [&](int i) { return npc_at(NPC_POSEIDON_ADDRS[i]); };
            auto val = [&](int i) { return npc_at(NPC_POSEIDON_ADDRS[i] + 1); };
            const Domain EVERY_64 = every(64);
            const std::vector<Factor> D14{F(512, 3, 4), F(512, 7, 8)};
            std::vector<Factor> D15{F(512, 5, 8)}; D15.insert(D15.end(), D14.begin(), D14.end());
            const std::vector<Factor> D16{F(512, 31, 32)};
            std::vector<Factor> D17{F(512, 11, 16), F(512, 23, 32), F(512, 25, 32), F(512, 13, 16), F(512, 27, 32), F(512, 29, 32), F(512, 15, 16)};
            D17.insert(D17.end(), D16.begin(), D16.end());
            std::vector<Factor> D19{F(512, 61, 64), F(512, 63, 64)}; D19.insert(D19.end(), D16.begin(), D16.end());
            std::vector<Factor> D20{F(512, 19, 32), F(512, 21, 32)}; D20.insert(D20.end(), D15.begin(), D15.end()); D20.insert(D20.end(), D17.begin(), D17.end());
            std::vector<Factor> D14_17 = D14; D14_17.insert(D14_17.end(), D17.begin(), D17.end());
            const Domain POSEIDON_ADDR_STEP{D15, {F(64)}}, POSEIDON_PARTIAL1_SQUARING{D14_17, {F(16)}}, POSEIDON_HALF_FULL_ROUND_TRANSITION{{F(256, 3, 4)}, {F(64)}};
            const Domain POSEIDON_PARTIAL_ROUND0{D19, {F(8)}}, POSEIDON_PARTIAL_ROUND1{D20, {F(16)}};
            ADD(FIRST_ROW, addr(0) - HINT(initial_poseidon_addr));
            ADD(POSEIDON_ADDR_STEP, addr(1) - (addr(0) + one));
            ADD(EVERY_512_EXCEPT_LAST, npc_at(512 + NPC_POSEIDON_ADDRS[0]) - (addr(5) + one));
            for (int j = 0; j < 3; ++j) ADD(EVERY_64, full(j) * full(j) - full_sq(j));
            ADD(EVERY_8, part(0) * part(0) - part_sq(0));
            ADD(POSEIDON_PARTIAL1_SQUARING, part(1) * part(1) - part_sq(1));
            for (int j = 0; j < 3; ++j) ADD(EVERY_512, val(j) + CF(keys.round[0][j]) - full(j));
            {
                const E c0 = cubed(0, 0), c1 = cubed(1, 0), c2 = cubed(2, 0);
                ADD(POSEIDON_HALF_FULL_ROUND_TRANSITION, full(0, 1) - (c0 + c0 + c0 + c1 + c2 + column(g, T_POSEIDON_FULL_KEY0)));
                ADD(POSEIDON_HALF_FULL_ROUND_TRANSITION, full(1, 1) + c1 - (c0 + c2 + column(g, T_POSEIDON_FULL_KEY1)));
                ADD(POSEIDON_HALF_FULL_ROUND_TRANSITION, full(2, 1) + c2 + c2 - (c0 + c1 + column(g, T_POSEIDON_FULL_KEY2)));
                const E l0 = cubed(0, 7), l1 = cubed(1, 7), l2 = cubed(2, 7);
                ADD(EVERY_512, val(3) - (l0 + l0 + l0 + l1 + l2));
                ADD(EVERY_512, val(4) + l1 - (l0 + l2));
                ADD(EVERY_512, val(5) + l2 + l2 - (l0 + l1));
            }
            for (uint64_t i = 0; i < 3; ++i) ADD(EVERY_512, part(0, 61 + i) - part(1, i));
            {
                const E m0 = cubed(0, 3), m1 = cubed(1, 3), m2 = cubed(2, 3);
                const E p0c0 = pcubed(0, 0), p0c1 = pcubed(0, 1);
                ADD(EVERY_512, part(0, 0) + m2 + m2 - (m0 + m1 + CF(keys.margin_full_to_partial[0])));
                ADD(EVERY_512, part(0, 1) - (m1 * NEG(4) + m2 * C(10) + part(0, 0) * C(4) + p0c0 * NEG(2) + CF(keys.margin_full_to_partial[1])));
                ADD(EVERY_512, part(0, 2) - (m2 * C(8) + part(0, 0) * C(4) + p0c0 * C(6) + part(0, 1) + part(0, 1) + p0c1 * NEG(2) + CF(keys.margin_full_to_partial[2])));
            }
            for (int which = 0; which < 2; ++which) {
                const E pc0 = pcubed(which, 0), pc1 = pcubed(which, 1), pc2 = pcubed(which, 2);
                const E key = column(g, which == 0 ? T_POSEIDON_PARTIAL_KEY0 : T_POSEIDON_PARTIAL_KEY1);
                const E expr = part(which, 3) - (pc0 * C(8) + part(which, 1) * C(4) + pc1 * C(6) + part(which, 2) + part(which, 2) + pc2 * NEG(2) + key);
                if (which == 0) ADD(POSEIDON_PARTIAL_ROUND0, expr); else ADD(POSEIDON_PARTIAL_ROUND1, expr);
            }
            {
                const E q19 = pcubed(1, 19), q20 = pcubed(1, 20), q21 = pcubed(1, 21);
                ADD(EVERY_512, full(0, 4) - (q19 * C(16) + part(1, 20) * C(8) + q20 * C(16) + part(1, 21) * C(6) + q21 + CF(keys.margin_partial_to_full[0])));
                ADD(EVERY_512, full(1, 4) - (q20 * C(4) + part(1, 21) + part(1, 21) + q21 + CF(keys.margin_partial_to_full[1])));
                ADD(EVERY_512, full(2, 4) - (q19 * C(8) + part(1, 20) * C(4) + q20 * C(6) + part(1, 21) + part(1, 21) + q21 * NEG(2) + CF(keys.margin_partial_to_full[2])));
            }
        }
#undef ADD
        return composer.total();
    }
};

}  // namespace

std::unique_ptr<Air> make_starknet_air(ss_ctx *ctx, const AirPublicInput &pi, uint32_t log_n, uint32_t log_blowup, uint64_t lde_offset) {
    return std::unique_ptr<Air>(new StarknetAir(ctx, pi, log_n, log_blowup, lde_offset));
}

}  // namespace ssh
