// trace_starknet.cpp — see trace_starknet.hpp.  Section by section the mirror of layouts/starknet.py::base_trace.
#include "trace_starknet.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <exception>
#include <memory>
#include <parallel/algorithm>
#include <tuple>

#include <mutex>
#include <omp.h>

#include "../../include/sandstorm_hip.h"
#include "trace_common.hpp"
#include "device_trace.hpp"

namespace ssh {

using namespace tracedetail;

namespace {

constexpr uint64_t CYCLE_HEIGHT = 16, PUBLIC_MEMORY_STEP = 8, RANGE_CHECK_STEP = 4, DILUTED_CHECK_STEP = 8;
constexpr uint64_t PEDERSEN_BUILTIN_RATIO = 32, RANGE_CHECK_BUILTIN_RATIO = 16, RANGE_CHECK_BUILTIN_PARTS = 8, BITWISE_RATIO = 64;
constexpr uint64_t ECDSA_BUILTIN_RATIO = 2048, EC_OP_BUILTIN_RATIO = 1024, POSEIDON_RATIO = 32;
enum { COL_FLAGS, COL_PEDERSEN_X, COL_PEDERSEN_Y, COL_PEDERSEN_SUFFIX, COL_PEDERSEN_SLOPE, COL_NPC, COL_MEMORY, COL_RANGE_CHECK, COL_AUXILIARY, NUM_COLS };
enum { NPC_PC = 0, NPC_PUB_MEM_ADDR = 2, NPC_MEM_OP0_ADDR = 4, NPC_MEM_DST_ADDR = 8, NPC_MEM_OP1_ADDR = 12, NPC_UNUSED_ADDR = 14,
       NPC_PEDERSEN_INPUT0_ADDR = 6, NPC_PEDERSEN_INPUT1_ADDR = 262, NPC_PEDERSEN_OUTPUT_ADDR = 134, NPC_RANGE_CHECK128_ADDR = 70,
       NPC_ECDSA_PUBKEY_ADDR = 390, NPC_ECDSA_MESSAGE_ADDR = 16774, NPC_BITWISE_POOL_ADDR = 198, NPC_BITWISE_X_OR_Y_ADDR = 902 };
const uint64_t NPC_EC_OP_ADDRS[7] = {8582, 4486, 12678, 2438, 10630, 6534, 14726};            // p.x, p.y, q.x, q.y, m, r.x, r.y
const uint64_t NPC_POSEIDON_ADDRS[6] = {38, 102, 166, 230, 294, 358};
enum { RC_OFF_DST = 0, RC_ORDERED = 2, RC_OFF_OP1 = 4, RC_OFF_OP0 = 8, RC_UNUSED = 12, RC16_COMPONENT = 12 };
enum { AUX_AP = 0, AUX_TMP0 = 2, AUX_OP0_MUL_OP1 = 4, AUX_FP = 8, AUX_TMP1 = 10, AUX_RES = 12 };
enum { DC_UNORDERED = 1, DC_ORDERED = 5 };
enum { EC_PUBKEY_DOUBLING_X = 1, EC_PUBKEY_DOUBLING_Y = 33, EC_PUBKEY_DOUBLING_SLOPE = 35, EC_PUBKEY_PARTIAL_SUM_X = 17, EC_PUBKEY_PARTIAL_SUM_Y = 49,
       EC_PUBKEY_PARTIAL_SUM_X_DIFF_INV = 51, EC_PUBKEY_PARTIAL_SUM_SLOPE = 19, EC_R_SUFFIX = 9, EC_MESSAGE_SUFFIX = 59, EC_GENERATOR_PARTIAL_SUM_Y = 91,
       EC_GENERATOR_PARTIAL_SUM_X = 27, EC_GENERATOR_PARTIAL_SUM_X_DIFF_INV = 7, EC_GENERATOR_PARTIAL_SUM_SLOPE = 123, EC_R_POINT_SLOPE = 16331,
       EC_R_POINT_X_DIFF_INV = 32715, EC_R_INV = 16355, EC_W_INV = 32739, EC_MESSAGE_INV = 16363, EC_PUBKEY_X_SQUARED = 32747, EC_B_SLOPE = 32763,
       EC_B_X_DIFF_INV = 32647 };
enum { OP_Q_DOUBLING_X = 41, OP_Q_DOUBLING_Y = 25, OP_Q_DOUBLING_SLOPE = 57, OP_R_PARTIAL_SUM_X = 5, OP_R_PARTIAL_SUM_Y = 37, OP_R_PARTIAL_SUM_SLOPE = 11,
       OP_R_PARTIAL_SUM_X_DIFF_INV = 43, OP_M_SUFFIX = 21, OP_M_BIT251_AND_BIT196_AND_BIT192 = 16371, OP_M_BIT251_AND_BIT196 = 16339 };
const uint64_t BITWISE_SHIFTED_CELLS[4] = {9, 521, 265, 777};

// body(k) for k in [0, count) on all host threads; the first exception a body throws is rethrown on the caller's thread
template <class Body> void parallel_for(uint64_t count, const Body &body) {
    std::exception_ptr err;
#pragma omp parallel for schedule(static)
    for (uint64_t k = 0; k < count; ++k) {
        try { body(k); } catch (...) {
#pragma omp critical(ssh_trace_error)
            if (!err) err = std::current_exception();
        }
    }
    if (err) std::rethrow_exception(err);
}

// the curve's group order (builtins/src/utils.rs:134)
const U256 CURVE_ORDER{0x1e66a241adc64d2full, 0xb781126dcae7b232ull, 0xffffffffffffffffull, 0x0800000000000010ull};

bool is_zero(const U256 &v) { return (v[0] | v[1] | v[2] | v[3]) == 0; }
bool less(const U256 &a, const U256 &b) {
    for (int k = 3; k >= 0; --k) if (a[k] != b[k]) return a[k] < b[k];
    return false;
}
unsigned bit_length(const U256 &v) {
    for (int k = 3; k >= 0; --k) if (v[k]) return 64 * k + 64 - (unsigned)__builtin_clzll(v[k]);
    return 0;
}
U256 add_raw(const U256 &a, const U256 &b) {
    U256 r;
    unsigned __int128 c = 0;
    for (int k = 0; k < 4; ++k) { c += (unsigned __int128)a[k] + b[k]; r[k] = (uint64_t)c; c >>= 64; }
    return r;
}
U256 sub_raw(const U256 &a, const U256 &b) {
    U256 r;
    unsigned __int128 br = 0;
    for (int k = 0; k < 4; ++k) { const unsigned __int128 t = (unsigned __int128)a[k] - b[k] - (uint64_t)br; r[k] = (uint64_t)t; br = (t >> 64) & 1; }
    return r;
}
// arithmetic modulo the group order (only the dummy signature needs it): values below 2^252, so sums do not overflow 256 bits
U256 addmod_n(const U256 &a, const U256 &b) { const U256 s = add_raw(a, b); return less(s, CURVE_ORDER) ? s : sub_raw(s, CURVE_ORDER); }
U256 mulmod_n(const U256 &a, const U256 &b) {
    U256 acc{};
    for (int i = (int)bit_length(b) - 1; i >= 0; --i) { acc = addmod_n(acc, acc); if (bit(b, (unsigned)i)) acc = addmod_n(acc, a); }
    return acc;
}
U256 invmod_n(const U256 &a) {                           // a^(N - 2)
    const U256 e = sub_raw(CURVE_ORDER, U256{2, 0, 0, 0});
    U256 acc{1, 0, 0, 0};
    for (int i = (int)bit_length(e) - 1; i >= 0; --i) { acc = mulmod_n(acc, acc); if (bit(e, (unsigned)i)) acc = mulmod_n(acc, a); }
    return acc;
}
U256 canonical_of(const Felt &f) {
    const std::array<uint8_t, 32> be = canonical_be_bytes(f);
    U256 v{};
    for (int i = 0; i < 32; ++i) v[(31 - i) / 8] |= (uint64_t)be[i] << (8 * ((31 - i) % 8));
    return v;
}

Pt ec_neg(const Pt &p) { return Pt{p.x, felt_neg(p.y)}; }
// calculate_slope (builtins/src/utils.rs:163-181): the chord through two points, the tangent when they coincide
Felt slope_of(const Pt &p1, const Pt &p2) {
    if (felt_eq(p1.x, p2.x)) {
        if (!felt_eq(p1.y, p2.y)) fail("vertical chord");
        const Felt xx = felt_mul(p1.x, p1.x);
        return felt_mul(felt_add(felt_add(felt_add(xx, xx), xx), felt_from_u64(1)), felt_inv(felt_add(p1.y, p1.y)));
    }
    return felt_mul(felt_sub(p1.y, p2.y), felt_inv(felt_sub(p1.x, p2.x)));
}
Pt ec_mul_small(uint64_t k, const Pt &p) {
    bool have = false;
    Pt acc = p, addend = p;
    for (; k; k >>= 1) {
        if (k & 1) { acc = have ? ec_add(acc, addend) : addend; have = true; }
        addend = ec_double(addend);
    }
    return acc;
}
// Tonelli-Shanks: p - 1 = 2^192 * (2^59 + 17); 3 generates the multiplicative group
bool felt_sqrt(const Felt &a, Felt &root) {
    if (felt_is_zero(a)) { root = a; return true; }
    const uint64_t q = (1ull << 59) + 17;
    const Felt one = felt_from_u64(1);
    Felt legendre = felt_pow(a, q);
    for (int i = 0; i < 191; ++i) legendre = felt_mul(legendre, legendre);
    if (!felt_eq(legendre, one)) return false;
    unsigned m = 192;
    Felt c = felt_pow(felt_from_u64(3), q), t = felt_pow(a, q), r = felt_pow(a, (q + 1) / 2);
    while (!felt_eq(t, one)) {
        unsigned i = 0;
        for (Felt t2 = t; !felt_eq(t2, one); t2 = felt_mul(t2, t2)) ++i;
        Felt b = c;
        for (unsigned k = 0; k + i + 1 < m; ++k) b = felt_mul(b, b);
        m = i; c = felt_mul(b, b); t = felt_mul(t, c); r = felt_mul(r, b);
    }
    root = r;
    return true;
}

struct Doubling { Pt point; Felt slope; };
// ecdsa/mod.rs:192-206: p, 2 p, 4 p, ... with the tangents' slopes (3 x^2 + 1) / (2 y) - the chain in Jacobian coordinates, the
// affine points and the slopes' denominators from two batched inversions (the reference inverts per step)
std::vector<Doubling> doubling_steps(const Pt &p) {
    std::vector<Jac> chain(256);
    chain[0] = jac_of(p);
    for (int i = 1; i < 256; ++i) chain[i] = jac_double(chain[i - 1]);
    const std::vector<Pt> pts = batch_normalize(chain);
    std::vector<Felt> den(256);
    for (int i = 0; i < 256; ++i) den[i] = felt_add(pts[i].y, pts[i].y);
    batch_invert(den.data(), den.size());
    const Felt one = felt_from_u64(1);
    std::vector<Doubling> out(256);
    for (int i = 0; i < 256; ++i) {
        const Felt xx = felt_mul(pts[i].x, pts[i].x);
        out[i] = Doubling{pts[i], felt_mul(felt_add(felt_add(felt_add(xx, xx), xx), one), den[i])};
    }
    return out;
}
// The two chains of a multiply-add (gen_ec_mad_steps / mimic_ec_mad_air): point_i = 2^min(i, max_doublings) point and
// partial_i = partial + the point_j of the set bits j < i of x, for i < steps; `last` = the sum after the last step.  Both chains are
// accumulated in Jacobian coordinates and made affine by one batched inversion each.  met: the first step whose partial sum shares its
// x with the step's point (the reference divides by that difference: a panic there, `false` in the signature check), else -1; the
// chains are not filled in then.
struct MadChain { std::vector<Pt> point, partial; Pt last; int met = -1; };
MadChain mad_chain(const U256 &x, const Pt &point, const Pt &partial, unsigned max_doublings, unsigned steps) {
    MadChain c;
    if (!steps) { c.last = partial; return c; }
    std::vector<Jac> dbl(steps);
    dbl[0] = jac_of(point);
    for (unsigned i = 1; i < steps; ++i) dbl[i] = i - 1 < max_doublings ? jac_double(dbl[i - 1]) : dbl[i - 1];
    c.point = batch_normalize(dbl);
    std::vector<Jac> sums(1, jac_of(partial));                // the partial sum changes at the set bits only
    std::vector<uint32_t> which(steps);
    for (unsigned i = 0; i < steps; ++i) {
        const Jac &cur = sums.back();
        which[i] = (uint32_t)sums.size() - 1;
        const Felt zz = felt_mul(cur.Z, cur.Z), h = felt_sub(felt_mul(c.point[i].x, zz), cur.X);
        if (felt_is_zero(h)) { c.met = (int)i; return c; }
        if (bit(x, i)) sums.push_back(jac_add_affine(cur, c.point[i], zz, h));
    }
    const std::vector<Pt> affine = batch_normalize(sums);
    c.partial.resize(steps);
    for (unsigned i = 0; i < steps; ++i) c.partial[i] = affine[which[i]];
    c.last = affine.back();
    return c;
}
struct MadStep { Pt partial; Felt suffix, slope, x_diff_inv; };
// gen_ec_mad_steps (ecdsa/mod.rs:157-190, ec_op/mod.rs:98-130): ONE more batched inversion serves every step's three uses of
// 1 / (partial.x - point.x) - the cell itself and the chord's slope (point.y - partial.y) / (point.x - partial.x) (the sum along the
// chord is the chain's next partial sum)
std::vector<MadStep> ec_mad_steps(const U256 &x, const Pt &point, const Pt &partial, unsigned max_doublings) {
    const MadChain c = mad_chain(x, point, partial, max_doublings, 256);
    if (c.met >= 0) fail("a partial sum meets the fixed point");
    std::vector<Felt> x_diff_inv(256);
    for (unsigned i = 0; i < 256; ++i) x_diff_inv[i] = felt_sub(c.partial[i].x, c.point[i].x);
    batch_invert(x_diff_inv.data(), x_diff_inv.size());
    const Felt zero = felt_from_u64(0);
    std::vector<MadStep> out(256);
    for (unsigned i = 0; i < 256; ++i)
        out[i] = MadStep{c.partial[i], felt_from_canonical(shr(x, i)), bit(x, i) ? felt_mul(felt_sub(c.partial[i].y, c.point[i].y), x_diff_inv[i]) : zero,
                         x_diff_inv[i]};
    return out;
}
// mimic_ec_mad_air (ecdsa/mod.rs:278-301)
bool mimic_ec_mad(const U256 &m, const Pt &point, const Pt &partial, Pt &out) {
    const unsigned bits = bit_length(m);
    if (bits < 1 || bits >= 252) return false;
    const MadChain c = mad_chain(m, point, partial, 256, bits);
    if (c.met >= 0) return false;
    out = c.last;
    return true;
}

struct Curve { Pt generator; Felt beta; Pt shift; };
const Curve &curve() {
    static const Curve c = [] {
        Curve v;
        starknet_curve(v.generator.x, v.generator.y, v.beta);
        v.shift = pedersen_point(0);
        return v;
    }();
    return c;
}

struct EcdsaTrace {                                       // ecdsa::InstanceTrace::new (ecdsa/mod.rs:61-140)
    Pt pubkey, b;
    Felt message, b_slope, b_x_diff_inv, w_inv, r_inv, message_inv, r_point_slope, r_point_x_diff_inv;
    std::vector<MadStep> zg_steps, rq_steps, wb_steps;
    std::vector<Doubling> pubkey_doubling, b_doubling;
};
EcdsaTrace ecdsa_trace(const U256 &pubkey_x, const U256 &message, const U256 &r, const U256 &w) {
    const Curve &cv = curve();
    const Felt px = felt_from_canonical(pubkey_x);
    Felt y;
    if (!felt_sqrt(felt_add(felt_add(felt_mul(felt_mul(px, px), px), px), cv.beta), y)) fail("the public key is not on the curve");
    const Pt neg_shift = ec_neg(cv.shift);
    const U256 yc = canonical_of(y), ync = canonical_of(felt_neg(y));
    const Felt first = less(yc, ync) ? felt_neg(y) : y, second = less(yc, ync) ? y : felt_neg(y);      // the larger root first (verify, ecdsa/mod.rs:250-276)
    EcdsaTrace t;
    bool found = false;
    Pt zg, rq, wb;
    for (const Felt &cand : {first, second}) {
        const Pt q{px, cand};
        if (!mimic_ec_mad(message, cv.generator, neg_shift, zg) || !mimic_ec_mad(r, q, cv.shift, rq)) continue;
        if (!mimic_ec_mad(w, ec_add(zg, rq), cv.shift, wb)) continue;
        if (canonical_of(ec_add(wb, neg_shift).x) == r) { t.pubkey = q; found = true; break; }
    }
    if (!found) fail("signature is invalid");
    t.b = ec_add(zg, rq);
    t.b_slope = slope_of(zg, rq);
    t.b_x_diff_inv = felt_inv(felt_sub(zg.x, rq.x));
    t.zg_steps = ec_mad_steps(message, cv.generator, neg_shift, 250);
    t.rq_steps = ec_mad_steps(r, t.pubkey, cv.shift, 255);
    t.wb_steps = ec_mad_steps(w, t.b, cv.shift, 255);
    t.pubkey_doubling = doubling_steps(t.pubkey);
    t.b_doubling = doubling_steps(t.b);
    t.message = felt_from_canonical(message);
    t.w_inv = felt_inv(felt_from_canonical(w)); t.r_inv = felt_inv(felt_from_canonical(r)); t.message_inv = felt_inv(t.message);
    t.r_point_slope = slope_of(wb, neg_shift);
    t.r_point_x_diff_inv = felt_inv(felt_sub(wb.x, neg_shift.x));
    return t;
}
// gen_dummy_instance (ecdsa/mod.rs:208-248): private key 1, message pedersen(1, 0), the first nonce that gives r, w < 2^251
void ecdsa_dummy_instance(U256 &pubkey_x, U256 &message, U256 &r, U256 &w) {
    const Curve &cv = curve();
    const Felt one = felt_from_u64(1), zero = felt_from_u64(0);
    Felt h;
    if (ss_pedersen_hash_host(one.data(), zero.data(), h.data()) != SS_OK) fail("Pedersen hash failed");
    message = canonical_of(h);
    pubkey_x = canonical_of(cv.generator.x);
    for (uint64_t k = 1;; ++k) {
        r = canonical_of(ec_mul_small(k, cv.generator).x);
        if (is_zero(r) || bit_length(r) > 251) continue;
        const U256 s = addmod_n(message, r);              // message + r * private key
        if (is_zero(s)) continue;
        w = mulmod_n(U256{k, 0, 0, 0}, invmod_n(s));
        if (is_zero(w) || bit_length(w) > 251) continue;
        return;
    }
}

// (the squares are cells of the trace too: made with the instance's trace - once per DISTINCT instance, not once per block that holds it)
struct PoseidonTrace { std::array<Felt, 3> full[8], full_sq[8]; std::vector<Felt> partial, partial_sq; Felt out[3]; };
Felt cube(const Felt &v) { return felt_mul(felt_mul(v, v), v); }
PoseidonTrace poseidon_trace(const std::array<Felt, 3> &input) {       // poseidon::InstanceTrace::new (poseidon/mod.rs:45-98)
    const auto &rk = poseidon_round_keys();
    PoseidonTrace t;
    std::array<Felt, 3> st = input;
    unsigned r = 0, nfull = 0;
    for (int phase = 0; phase < 3; ++phase)
        for (int i = 0; i < (phase == 1 ? 83 : 4); ++i, ++r) {
            for (int j = 0; j < 3; ++j) st[j] = felt_add(st[j], rk[r][j]);
            if (phase == 1) { t.partial.push_back(st[2]); st[2] = cube(st[2]); }
            else { t.full[nfull++] = st; for (auto &v : st) v = cube(v); }
            st = {felt_add(felt_add(felt_add(felt_add(st[0], st[0]), st[0]), st[1]), st[2]), felt_add(felt_sub(st[0], st[1]), st[2]),
                  felt_sub(felt_add(st[0], st[1]), felt_add(st[2], st[2]))};
        }
    for (int j = 0; j < 3; ++j) t.out[j] = st[j];
    for (int rnd = 0; rnd < 8; ++rnd) for (int j = 0; j < 3; ++j) t.full_sq[rnd][j] = felt_mul(t.full[rnd][j], t.full[rnd][j]);
    t.partial_sq.resize(t.partial.size());
    for (size_t k = 0; k < t.partial.size(); ++k) t.partial_sq[k] = felt_mul(t.partial[k], t.partial[k]);
    return t;
}

// an EC-op instance's trace (ec_op::InstanceTrace::new, ec_op/mod.rs:40-130)
struct EcOpTrace { Pt p, q, r; Felt m; std::vector<Doubling> q_doubling; std::vector<MadStep> r_steps; bool b251_196, b251_196_192; };
using U256x2 = std::pair<U256, U256>;
using U256x3 = std::tuple<U256, U256, U256>;
using U256x4 = std::tuple<U256, U256, U256, U256>;
using U256x5 = std::tuple<U256, U256, U256, U256, U256>;

// the dummy signature (a constant: ecdsa/mod.rs gen_dummy_instance) and its trace - three scalar multiplications with their doubling
// chains, ~0.15 s - are made once per process
struct EcdsaDummy { U256x4 key; std::shared_ptr<const EcdsaTrace> trace; };
const EcdsaDummy &ecdsa_dummy() {
    static const EcdsaDummy d = [] {
        U256 v[4];
        ecdsa_dummy_instance(v[0], v[1], v[2], v[3]);
        return EcdsaDummy{U256x4{v[0], v[1], v[2], v[3]}, std::make_shared<const EcdsaTrace>(ecdsa_trace(v[0], v[1], v[2], v[3]))};
    }();
    return d;
}

const ss_trace_layout &cpu_layout() {                     // the CPU's cells in a cycle's 16 rows (starknet air.rs:2538-3250: Npc, RangeCheck, Auxiliary)
    static const ss_trace_layout l = [] {
        ss_trace_layout v;
        for (int j = 0; j < 8; ++j) v.npc_pair[j] = SS_TRACE_NPC_PAD;
        v.npc_pair[NPC_PC / 2] = SS_TRACE_NPC_PC; v.npc_pair[NPC_MEM_OP0_ADDR / 2] = SS_TRACE_NPC_OP0; v.npc_pair[NPC_MEM_DST_ADDR / 2] = SS_TRACE_NPC_DST;
        v.npc_pair[NPC_MEM_OP1_ADDR / 2] = SS_TRACE_NPC_OP1;
        for (uint64_t o = 0; o < CYCLE_HEIGHT; o += PUBLIC_MEMORY_STEP) v.npc_pair[(o + NPC_PUB_MEM_ADDR) / 2] = SS_TRACE_NPC_PUBLIC;
        for (int o = 0; o < 16; ++o) { v.rc_cell[o] = SS_TRACE_RC_FILL; v.aux_cell[o] = SS_TRACE_AUX_ZERO; }
        v.rc_cell[RC_OFF_DST] = SS_TRACE_RC_OFF_DST; v.rc_cell[RC_OFF_OP1] = SS_TRACE_RC_OFF_OP1; v.rc_cell[RC_OFF_OP0] = SS_TRACE_RC_OFF_OP0;
        for (uint64_t o = 0; o < CYCLE_HEIGHT; o += DILUTED_CHECK_STEP) v.rc_cell[o + DC_UNORDERED] = v.rc_cell[o + DC_ORDERED] = SS_TRACE_RC_ZERO;      // trace.rs:294-302
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
    const StarknetPrivateInput &priv;
    uint64_t num_cycles, n;
    Felt pad_value;
};

// ---- the host backend: the sections' cells straight into the caller's columns
struct HostBackend {
    const Inputs &in;
    Felt *const *out;
    const std::function<void(int)> *column_done;
    AddrArray npc_addr;
    std::chrono::steady_clock::time_point t_last = std::chrono::steady_clock::now();
    const bool timing = getenv("SSH_TRACE_TIMING") != nullptr;               // stage times on stderr
    HostBackend(const Inputs &in_, Felt *const *out_, const std::function<void(int)> *cd) : in(in_), out(out_), column_done(cd), npc_addr(in_.n / 2) {}
    void lap(const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[starknet trace] %-28s %8.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    }
    void done(std::initializer_list<int> cs) { if (column_done && *column_done) for (int c : cs) (*column_done)(c); }
    void set_pair(uint64_t row, uint64_t address, const Felt &value) { out[COL_NPC][row] = felt_from_u64(address); out[COL_NPC][row + 1] = value; npc_addr[row / 2] = address; }

    // The CPU's cells (trace.rs:177-244) and the range-check pool's (trace.rs:165-235, 294-302).  The generator is bound by the host's
    // memory traffic (4.8 GB of columns at 2^20 steps), so every column this section touches is written ONCE: a cycle's 16 rows of the
    // flags, the memory pool, the range-check column and the auxiliary column are made in a block on the stack - padding first, then
    // what the cycle puts there - and stored row after row.  Cells of these columns that a builtin owns get the padding here and their
    // values in the builtin's section.
    void cpu_section(const RcPoolPlan &pool, const std::vector<uint32_t> &rc_count, uint64_t rc_fill) {
        const uint64_t num_cycles = in.num_cycles;
        const RegisterStates &states = in.states;
        const Mem &mem = in.mem;
        Felt *const flags = out[COL_FLAGS], *const npc = out[COL_NPC], *const rc_col = out[COL_RANGE_CHECK], *const aux = out[COL_AUXILIARY];
        const Felt zero = felt_from_u64(0), pad_addr = felt_from_u64(1), pad_value = in.pad_value, rc_max_f = felt_from_u64(rc_fill);
        CountArray ordered_vals;
        ordered_runs(rc_count, pool.lo, pool.hi, ordered_vals);
        const uint64_t per = CYCLE_HEIGHT / RANGE_CHECK_STEP;
        constexpr uint64_t JNZ_BLOCK = 512;                  // cycles whose conditional jumps share one inversion (JnzInverses)
        parallel_for((num_cycles + JNZ_BLOCK - 1) / JNZ_BLOCK, [&](uint64_t block) {
            const uint64_t first_cycle = block * JNZ_BLOCK, end_cycle = std::min(num_cycles, first_cycle + JNZ_BLOCK);
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
                if (w.pc_update() == 4) res = felt_is_zero(dst) ? zero : jnz.take(dst);
                else if (w.res_logic() == 0) res = op1;
                else if (w.res_logic() == 1) res = felt_add(op0, op1);
                else if (w.res_logic() == 2) res = felt_mul(op0, op1);
                else fail("invalid res logic");
                const Felt tmp0 = w.flag(F_PC_JNZ) ? dst : zero;
                Felt blk[CYCLE_HEIGHT];
                // flags
                for (int f = 0; f < 16; ++f) flags[r + f] = felt_from_u64(w.flag_prefix(f));
                // memory pool: (address, value) pairs, the padding pair where the CPU has none (a builtin's pair comes with its section)
                for (uint64_t o = 0; o < CYCLE_HEIGHT; o += 2) { blk[o] = pad_addr; blk[o + 1] = pad_value; npc_addr[(r + o) / 2] = 1; }
                auto pair = [&](uint64_t o, uint64_t address, const Felt &value) { blk[o] = felt_from_u64(address); blk[o + 1] = value; npc_addr[(r + o) / 2] = address; };
                pair(NPC_PC, pc, felt_from_canonical(iw));
                pair(NPC_MEM_OP0_ADDR, op0_addr, op0);
                pair(NPC_MEM_DST_ADDR, dst_addr, dst);
                pair(NPC_MEM_OP1_ADDR, op1_addr, op1);
                for (uint64_t o = 0; o < CYCLE_HEIGHT; o += PUBLIC_MEMORY_STEP) pair(o + NPC_PUB_MEM_ADDR, 0, zero);
                for (uint64_t o = 0; o < CYCLE_HEIGHT; ++o) npc[r + o] = blk[o];
                // auxiliary column: zero where no section writes
                for (uint64_t o = 0; o < CYCLE_HEIGHT; ++o) blk[o] = zero;
                blk[AUX_TMP0] = tmp0; blk[AUX_TMP1] = felt_mul(tmp0, res);
                blk[AUX_AP] = felt_from_u64(ap); blk[AUX_FP] = felt_from_u64(fp);
                blk[AUX_OP0_MUL_OP1] = felt_mul(op0, op1); blk[AUX_RES] = res;
                for (uint64_t o = 0; o < CYCLE_HEIGHT; ++o) aux[r + o] = blk[o];
                // range-check column: the padding value, the instruction's offsets, the odd cycles' next padding value, the cycle's ordered
                // values, zeros where the diluted check's cells are
                for (uint64_t o = 0; o < CYCLE_HEIGHT; ++o) blk[o] = rc_max_f;
                blk[RC_OFF_DST] = felt_from_u64(w.off_dst()); blk[RC_OFF_OP1] = felt_from_u64(w.off_op1()); blk[RC_OFF_OP0] = felt_from_u64(w.off_op0());
                if (cycle % 2 == 1) blk[RC_UNUSED] = felt_from_u64(pool.pad(pool.pad0 + cycle / 2));
                for (uint64_t o = 0; o < CYCLE_HEIGHT; o += RANGE_CHECK_STEP) {
                    const uint64_t at = cycle * per + o / RANGE_CHECK_STEP;
                    blk[o + RC_ORDERED] = felt_from_u64(at < ordered_vals.size() ? ordered_vals[at] : pool.hi);
                }
                for (uint64_t o = 0; o < CYCLE_HEIGHT; o += DILUTED_CHECK_STEP) blk[o + DC_UNORDERED] = blk[o + DC_ORDERED] = zero;      // trace.rs:294-302
                for (uint64_t o = 0; o < CYCLE_HEIGHT; ++o) rc_col[r + o] = blk[o];
            }
        });
        lap("cpu cells + range-check pool");
        done({COL_FLAGS});
    }
    // a builtin's blocks: block i holds template of_block[i]
    template <class Place> void builtin(const char *what, const std::vector<uint32_t> &of_block, uint32_t, uint64_t step, uint64_t begin, uint64_t per_block, const Place &place) {
        parallel_for(of_block.size(), [&](uint64_t i) {
            HostSink sink{out, &npc_addr, COL_NPC, i * step, begin + per_block * i};
            place(sink, of_block[i]);
        });
        lap(what);
    }
    // the range-check builtin's slots (trace.rs:388-426): the given instances, then dummies made of the pool's padding values
    void rc_builtin(const ss_trace_rc_plan &plan, const RcPoolPlan &pool, const std::vector<uint64_t> &given3) {
        Felt *const rc_col = out[COL_RANGE_CHECK];
        parallel_for(plan.n_slots, [&](uint64_t s) {
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
        lap("range-check builtin");
    }
    void diluted(const DilutedPlan &dp, const std::vector<uint64_t> &pad_rows, const std::vector<uint64_t> &pad_values, uint64_t slots) {
        Felt *const rc_col = out[COL_RANGE_CHECK];
        for (size_t k = 0; k < pad_rows.size(); ++k) rc_col[pad_rows[k]] = felt_from_u64(pad_values[k]);
        parallel_for(1u << DILUTED_N_BITS, [&](uint64_t v) {                // the short runs value by value
            if (dp.first[v + 1] - dp.first[v] >= 4096) return;
            const Felt f = felt_from_u64(dilute((uint32_t)v));
            for (uint64_t k = dp.first[v]; k < dp.first[v + 1]; ++k) rc_col[DILUTED_CHECK_STEP * k + DC_ORDERED] = f;
        });
        for (uint32_t v = 0; v < (1u << DILUTED_N_BITS); ++v)                 // the long ones (value 0 of the dummy instances) by all threads
            if (dp.first[v + 1] - dp.first[v] >= 4096) {
                const Felt f = felt_from_u64(dilute(v));
                const uint64_t k0 = dp.first[v];
                parallel_for(dp.first[v + 1] - k0, [&](uint64_t j) { rc_col[DILUTED_CHECK_STEP * (k0 + j) + DC_ORDERED] = f; });
            }
        (void)slots;
        lap("diluted pool");
    }
    // gap fillers (trace.rs:890-925) and the ordered memory (get_ordered_memory_accesses, utils.rs:112-152)
    void memory() {
        done({COL_RANGE_CHECK, COL_AUXILIARY});
        const std::vector<uint64_t> gaps = memory_gaps(npc_addr, in.pi.public_memory, in.num_cycles);
        if (gaps.size() > in.num_cycles) fail("more memory gaps than cycles to hold them");
        for (size_t k = 0; k < gaps.size(); ++k) set_pair(k * CYCLE_HEIGHT + NPC_UNUSED_ADDR, gaps[k], felt_from_u64(0));
        lap("gap fillers");
        done({COL_NPC});
        ordered_memory_into(out[COL_MEMORY], in.n, npc_addr, out[COL_NPC], in.n / PUBLIC_MEMORY_STEP, in.pi.public_memory, in.pad_value);
        lap("sorted memory");
        done({COL_MEMORY});
    }
    void pedersen_done() { done({COL_PEDERSEN_X, COL_PEDERSEN_Y, COL_PEDERSEN_SUFFIX, COL_PEDERSEN_SLOPE}); }
};

// ---- the device backend: the same sections as uploads of plans / templates and kernel launches (device_trace.hpp, csrc/trace.hip)
struct DeviceBackend {
    const Inputs &in;
    DeviceTrace &dt;
    void cpu_section(const RcPoolPlan &pool, const std::vector<uint32_t> &rc_count, uint64_t rc_fill) {
        dt.cpu_cells(cpu_layout(), COL_FLAGS, COL_NPC, COL_RANGE_CHECK, COL_AUXILIARY, in.pad_value, rc_fill);
        ss_trace_rc_plan plan{};
        plan.ordered_step = RANGE_CHECK_STEP; plan.ordered_off = RC_ORDERED; plan.unused_off = RC_UNUSED;
        dt.rc_pool(plan, pool, rc_count, COL_RANGE_CHECK);
    }
    template <class Place> void builtin(const char *, const std::vector<uint32_t> &of_block, uint32_t n_templates, uint64_t step, uint64_t begin, uint64_t per_block,
                                        const Place &place) {
        dt.builtin(n_templates, of_block, step, begin, per_block, place);
    }
    void rc_builtin(const ss_trace_rc_plan &plan, const RcPoolPlan &, const std::vector<uint64_t> &given3) { dt.rc_builtin(plan, given3, COL_RANGE_CHECK); }
    void diluted(const DilutedPlan &dp, const std::vector<uint64_t> &pad_rows, const std::vector<uint64_t> &pad_values, uint64_t slots) {
        dt.patch(COL_RANGE_CHECK, pad_rows, pad_values);
        dt.ordered_runs(COL_RANGE_CHECK, DILUTED_CHECK_STEP, DC_ORDERED, slots, dp.first, 0, true);
    }
    void memory() { dt.ordered_memory(COL_MEMORY, in.pi.public_memory, in.n / PUBLIC_MEMORY_STEP, in.pad_value, NPC_UNUSED_ADDR); }
    void pedersen_done() {}
};

// ExecutionTrace::new (layouts/src/starknet/trace.rs:99-987) section by section, for either backend: where a builtin's cells go is
// said ONCE (the `place` lambdas), the host stores them per block, the device per distinct instance
template <class Backend> void generate(Backend &be, const Inputs &in) {
    const uint64_t num_cycles = in.num_cycles, n = in.n;
    const AirPublicInput &pi = in.pi;
    const StarknetPrivateInput &priv = in.priv;
    // ---- the range-check pool (trace.rs:142-165): the offsets of every instruction and the builtin's parts counted first
    std::vector<uint32_t> rc_count(1 << 16, 0);
    {
        std::vector<std::vector<uint32_t>> rc_count_of((size_t)omp_get_max_threads());
        parallel_for(num_cycles, [&](uint64_t cycle) {
            std::vector<uint32_t> &my_count = rc_count_of[(size_t)omp_get_thread_num()];
            if (my_count.empty()) my_count.assign(1 << 16, 0);
            const Word w{in.mem.at(in.states[cycle].pc)[0]};
            for (uint64_t v : {w.off_dst(), w.off_op0(), w.off_op1()}) ++my_count[v];
        });
        for (auto &part : rc_count_of) for (size_t v = 0; v < part.size(); ++v) rc_count[v] += part[v];
    }
    ss_trace_rc_plan rc_plan{};
    rc_plan.n_slots = num_cycles / RANGE_CHECK_BUILTIN_RATIO; rc_plan.n_given = priv.range_check.size(); rc_plan.slot_rows = RANGE_CHECK_BUILTIN_RATIO * CYCLE_HEIGHT;
    rc_plan.addr_begin = pi.segments[4].begin_addr; rc_plan.part_stride = 32; rc_plan.part_off = RC16_COMPONENT; rc_plan.pair_off = NPC_RANGE_CHECK128_ADDR;
    rc_plan.ordered_step = RANGE_CHECK_STEP; rc_plan.ordered_off = RC_ORDERED; rc_plan.unused_off = RC_UNUSED;
    if (rc_plan.n_given > rc_plan.n_slots) fail("more range-check instances than the trace has slots for");
    std::vector<uint64_t> rc_given;                       // index, value low, value high
    for (auto &inst : priv.range_check) {
        if (inst.value[2] | inst.value[3]) fail("range-check value does not fit 128 bits");
        rc_given.insert(rc_given.end(), {(uint64_t)inst.index, inst.value[0], inst.value[1]});
        for (unsigned k = 0; k < RANGE_CHECK_BUILTIN_PARTS; ++k) ++rc_count[(uint32_t)(shr(inst.value, 16 * (RANGE_CHECK_BUILTIN_PARTS - 1 - k))[0] & 0xffff)];
    }
    RcPoolPlan pool;
    pool.from_counts(rc_count, rc_plan.n_slots - rc_plan.n_given);          // the dummy instances take the first padding values (trace.rs:246-261)
    // the odd cycles take the next padding values in order, every cycle its CYCLE_HEIGHT / RANGE_CHECK_STEP ordered values
    if (pool.pad0 + num_cycles / 2 < pool.padding.size() || num_cycles * (CYCLE_HEIGHT / RANGE_CHECK_STEP) < pool.total) fail("range-check values do not fit the trace");
    rc_plan.rc_lo = pool.lo; rc_plan.rc_hi = pool.hi; rc_plan.n_padding = pool.padding.size(); rc_plan.pad0 = pool.pad0;
    be.cpu_section(pool, rc_count, pool.hi);

    // ---- Pedersen (trace.rs:304-386)
    {
        const uint64_t step = PEDERSEN_BUILTIN_RATIO * CYCLE_HEIGHT;
        const auto given = instances_by_index(priv.pedersen, n / step, "pedersen");
        Instances<U256x2, PedersenTrace> inst;
        inst.assign(n / step, [&](uint64_t i) { auto it = given.find((uint32_t)i); return it != given.end() ? U256x2{it->second->a, it->second->b} : U256x2{}; });
        inst.trace_all([](const U256x2 &k) { return pedersen_instance_trace_cached(k.first, k.second); });
        be.builtin("pedersen", inst.of_block, (uint32_t)inst.keys.size(), step, pi.segments[3].begin_addr, 3, [&](auto &s, uint32_t t) {
            const PedersenTrace &c = *inst.traces[t];
            for (uint64_t j = 0; j < 512; ++j) {
                s.cell(COL_PEDERSEN_X, j, c.steps[j].point.x); s.cell(COL_PEDERSEN_Y, j, c.steps[j].point.y);
                s.cell(COL_PEDERSEN_SUFFIX, j, c.steps[j].suffix); s.cell(COL_PEDERSEN_SLOPE, j, c.steps[j].slope);
            }
            const U256 *in2[2] = {&inst.keys[t].first, &inst.keys[t].second};
            for (int half = 0; half < 2; ++half) {
                const bool b251 = bit(*in2[half], 251), b196 = bit(*in2[half], 196), b192 = bit(*in2[half], 192);
                s.cell(COL_PEDERSEN_SLOPE, 256 * half + 255, felt_from_u64(b251 && b196));
                s.cell(COL_AUXILIARY, 256 * half + 71, felt_from_u64(b251 && b196 && b192));
            }
            s.pair(NPC_PEDERSEN_INPUT0_ADDR, 0, felt_from_canonical(inst.keys[t].first));
            s.pair(NPC_PEDERSEN_INPUT1_ADDR, 1, felt_from_canonical(inst.keys[t].second));
            s.pair(NPC_PEDERSEN_OUTPUT_ADDR, 2, c.out);
        });
        be.pedersen_done();
    }
    // ---- range-check builtin (trace.rs:388-426)
    be.rc_builtin(rc_plan, pool, rc_given);
    // ---- ECDSA (trace.rs:428-523)
    {
        const uint64_t step = ECDSA_BUILTIN_RATIO * CYCLE_HEIGHT;
        const auto given = instances_by_index(priv.ecdsa, n / step, "ecdsa");
        Instances<U256x4, EcdsaTrace> inst;
        inst.assign(n / step, [&](uint64_t i) {
            auto it = given.find((uint32_t)i);
            return it != given.end() ? U256x4{it->second->pubkey_x, it->second->message, it->second->r, it->second->w} : ecdsa_dummy().key;
        });
        inst.trace_all([](const U256x4 &k) {               // (three scalar multiplications with their doubling chains per signature)
            if (k == ecdsa_dummy().key) return ecdsa_dummy().trace;
            return std::make_shared<const EcdsaTrace>(ecdsa_trace(std::get<0>(k), std::get<1>(k), std::get<2>(k), std::get<3>(k)));
        });
        be.builtin("ecdsa", inst.of_block, (uint32_t)inst.keys.size(), step, pi.segments[5].begin_addr, 2, [&](auto &s, uint32_t ti) {
            const EcdsaTrace &t = *inst.traces[ti];
            for (int half = 0; half < 2; ++half) {
                const std::vector<MadStep> &mad = half ? t.wb_steps : t.rq_steps;
                const std::vector<Doubling> &dbl = half ? t.b_doubling : t.pubkey_doubling;
                for (uint64_t j = 0; j < 256; ++j) {
                    const uint64_t r = 64 * (256 * half + j);
                    s.cell(COL_AUXILIARY, r + EC_PUBKEY_DOUBLING_X, dbl[j].point.x); s.cell(COL_AUXILIARY, r + EC_PUBKEY_DOUBLING_Y, dbl[j].point.y);
                    s.cell(COL_AUXILIARY, r + EC_PUBKEY_DOUBLING_SLOPE, dbl[j].slope);
                    s.cell(COL_AUXILIARY, r + EC_PUBKEY_PARTIAL_SUM_X, mad[j].partial.x); s.cell(COL_AUXILIARY, r + EC_PUBKEY_PARTIAL_SUM_Y, mad[j].partial.y);
                    s.cell(COL_AUXILIARY, r + EC_PUBKEY_PARTIAL_SUM_SLOPE, mad[j].slope); s.cell(COL_AUXILIARY, r + EC_PUBKEY_PARTIAL_SUM_X_DIFF_INV, mad[j].x_diff_inv);
                    s.cell(COL_AUXILIARY, r + EC_R_SUFFIX, mad[j].suffix);
                }
            }
            for (uint64_t j = 0; j < 256; ++j) {
                const uint64_t r = 128 * j;
                const MadStep &st = t.zg_steps[j];
                s.cell(COL_AUXILIARY, r + EC_GENERATOR_PARTIAL_SUM_X, st.partial.x); s.cell(COL_AUXILIARY, r + EC_GENERATOR_PARTIAL_SUM_Y, st.partial.y);
                s.cell(COL_AUXILIARY, r + EC_GENERATOR_PARTIAL_SUM_SLOPE, st.slope); s.cell(COL_AUXILIARY, r + EC_GENERATOR_PARTIAL_SUM_X_DIFF_INV, st.x_diff_inv);
                s.cell(COL_AUXILIARY, r + EC_MESSAGE_SUFFIX, st.suffix);
            }
            s.cell(COL_AUXILIARY, EC_B_SLOPE, t.b_slope); s.cell(COL_AUXILIARY, EC_B_X_DIFF_INV, t.b_x_diff_inv); s.cell(COL_AUXILIARY, EC_W_INV, t.w_inv);
            s.cell(COL_AUXILIARY, EC_R_INV, t.r_inv); s.cell(COL_AUXILIARY, EC_R_POINT_SLOPE, t.r_point_slope); s.cell(COL_AUXILIARY, EC_R_POINT_X_DIFF_INV, t.r_point_x_diff_inv);
            s.cell(COL_AUXILIARY, EC_MESSAGE_INV, t.message_inv);
            s.cell(COL_AUXILIARY, EC_PUBKEY_X_SQUARED, felt_mul(t.pubkey.x, t.pubkey.x));
            s.pair(NPC_ECDSA_PUBKEY_ADDR, 0, t.pubkey.x);
            s.pair(NPC_ECDSA_MESSAGE_ADDR, 1, t.message);
        });
    }
    // ---- bitwise and the diluted check (trace.rs:525-705)
    {
        const uint64_t step = BITWISE_RATIO * CYCLE_HEIGHT;
        const auto given = instances_by_index(priv.bitwise, n / step, "bitwise");
        Instances<U256x2, BitwiseTrace> inst;
        inst.assign(n / step, [&](uint64_t i) { auto it = given.find((uint32_t)i); return it != given.end() ? U256x2{it->second->x, it->second->y} : U256x2{}; });
        inst.trace_all([](const U256x2 &k) { return bitwise_instance_trace(k.first, k.second); });
        be.builtin("bitwise", inst.of_block, (uint32_t)inst.keys.size(), step, pi.segments[6].begin_addr, 5, [&](auto &s, uint32_t ti) {
            const BitwiseTrace &t = *inst.traces[ti];
            for (unsigned k = 0; k < 4; ++k) s.cell(COL_RANGE_CHECK, BITWISE_SHIFTED_CELLS[k], t.shifted[k]);
            for (int p = 0; p < 4; ++p) for (int c = 0; c < 4; ++c) for (int sg = 0; sg < 4; ++sg) s.cell(COL_RANGE_CHECK, 256 * p + 16 * (4 * c + sg) + 1, t.parts[p][c][sg]);
            for (int k = 0; k < 4; ++k) s.pair(NPC_BITWISE_POOL_ADDR + 256 * k, k, t.memory[k]);
            s.pair(NPC_BITWISE_X_OR_Y_ADDR, 4, t.memory[4]);
        });
        // the diluted pool: every instance's 68 diluted cells counted (by template: nearly every block holds the dummy instance)
        std::vector<uint32_t> dil_count(1u << DILUTED_N_BITS, 0), blocks_of(inst.keys.size(), 0);
        for (uint32_t t : inst.of_block) ++blocks_of[t];
        for (size_t t = 0; t < inst.keys.size(); ++t) for (uint32_t v : inst.traces[t]->undiluted) dil_count[v] += blocks_of[t];
        const uint64_t slots = n / DILUTED_CHECK_STEP;
        DilutedPlan dp;
        dp.from_counts(dil_count, slots);
        std::vector<uint64_t> pad_rows, pad_values;
        size_t pi_d = 0;
        for (uint64_t blk = 0; blk < n / 1024 && pi_d < dp.padding.size(); ++blk)               // the free cells 8 i + 1, i odd (trace.rs:668-693)
            for (uint64_t i = 1; i < 1024 / DILUTED_CHECK_STEP && pi_d < dp.padding.size(); i += 2) {
                const uint64_t off = 8 * i + DC_UNORDERED;
                if (std::find(std::begin(BITWISE_SHIFTED_CELLS), std::end(BITWISE_SHIFTED_CELLS), off) != std::end(BITWISE_SHIFTED_CELLS)) continue;
                pad_rows.push_back(blk * 1024 + off);
                pad_values.push_back(dilute(dp.padding[pi_d++]));
            }
        if (pi_d != dp.padding.size()) fail("diluted-check values do not fit the trace");
        be.diluted(dp, pad_rows, pad_values, slots);
    }
    // ---- EC op (trace.rs:707-777)
    {
        const Curve &cv = curve();
        const uint64_t step = EC_OP_BUILTIN_RATIO * CYCLE_HEIGHT;
        const auto given = instances_by_index(priv.ec_op, n / step, "ec_op");
        // gen_dummy_instance (ec_op/mod.rs:84-96): P0 + 1 * G
        const U256x5 dummy{canonical_of(cv.shift.x), canonical_of(cv.shift.y), canonical_of(cv.generator.x), canonical_of(cv.generator.y), U256{1, 0, 0, 0}};
        Instances<U256x5, EcOpTrace> inst;
        inst.assign(n / step, [&](uint64_t i) {
            auto it = given.find((uint32_t)i);
            return it != given.end() ? U256x5{it->second->p_x, it->second->p_y, it->second->q_x, it->second->q_y, it->second->m} : dummy;
        });
        auto trace_of = [](const U256x5 &in5) {              // (a scalar multiplication with its doubling chain per instance)
            auto t = std::make_shared<EcOpTrace>();
            t->p = Pt{felt_from_canonical(std::get<0>(in5)), felt_from_canonical(std::get<1>(in5))};
            t->q = Pt{felt_from_canonical(std::get<2>(in5)), felt_from_canonical(std::get<3>(in5))};
            t->m = felt_from_canonical(std::get<4>(in5));
            t->q_doubling = doubling_steps(t->q);
            t->r_steps = ec_mad_steps(std::get<4>(in5), t->q, t->p, 255);
            t->r = t->r_steps.back().partial;
            t->b251_196 = bit(std::get<4>(in5), 251) && bit(std::get<4>(in5), 196); t->b251_196_192 = t->b251_196 && bit(std::get<4>(in5), 192);
            return std::shared_ptr<const EcOpTrace>(t);
        };
        inst.trace_all([&](const U256x5 &in5) {
            if (in5 == dummy) {                              // the dummy instance (a constant) is traced once per process
                static const std::shared_ptr<const EcOpTrace> cached = trace_of(in5);
                return cached;
            }
            return trace_of(in5);
        });
        be.builtin("ec op", inst.of_block, (uint32_t)inst.keys.size(), step, pi.segments[7].begin_addr, 7, [&](auto &s, uint32_t ti) {
            const EcOpTrace &t = *inst.traces[ti];
            for (uint64_t j = 0; j < 256; ++j) {
                const uint64_t r = 64 * j;
                s.cell(COL_AUXILIARY, r + OP_Q_DOUBLING_X, t.q_doubling[j].point.x); s.cell(COL_AUXILIARY, r + OP_Q_DOUBLING_Y, t.q_doubling[j].point.y);
                s.cell(COL_AUXILIARY, r + OP_Q_DOUBLING_SLOPE, t.q_doubling[j].slope);
                s.cell(COL_AUXILIARY, r + OP_R_PARTIAL_SUM_X, t.r_steps[j].partial.x); s.cell(COL_AUXILIARY, r + OP_R_PARTIAL_SUM_Y, t.r_steps[j].partial.y);
                s.cell(COL_AUXILIARY, r + OP_M_SUFFIX, t.r_steps[j].suffix);
                if (j != 255) {                              // the ECDSA builtin owns the last ones
                    s.cell(COL_AUXILIARY, r + OP_R_PARTIAL_SUM_SLOPE, t.r_steps[j].slope); s.cell(COL_AUXILIARY, r + OP_R_PARTIAL_SUM_X_DIFF_INV, t.r_steps[j].x_diff_inv);
                }
            }
            s.cell(COL_AUXILIARY, OP_M_BIT251_AND_BIT196, felt_from_u64(t.b251_196)); s.cell(COL_AUXILIARY, OP_M_BIT251_AND_BIT196_AND_BIT192, felt_from_u64(t.b251_196_192));
            const Felt values[7] = {t.p.x, t.p.y, t.q.x, t.q.y, t.m, t.r.x, t.r.y};
            for (int k = 0; k < 7; ++k) s.pair(NPC_EC_OP_ADDRS[k], k, values[k]);
        });
    }
    // ---- Poseidon (trace.rs:779-888)
    {
        const uint64_t step = POSEIDON_RATIO * CYCLE_HEIGHT;
        const uint64_t FULL[3][2] = {{53, 29}, {13, 61}, {45, 3}};
        const auto given = instances_by_index(priv.poseidon, n / step, "poseidon");
        Instances<U256x3, PoseidonTrace> inst;
        inst.assign(n / step, [&](uint64_t i) {
            auto it = given.find((uint32_t)i);
            return it != given.end() ? U256x3{it->second->input[0], it->second->input[1], it->second->input[2]} : U256x3{};
        });
        inst.trace_all([](const U256x3 &k) {
            return std::make_shared<const PoseidonTrace>(poseidon_trace(std::array<Felt, 3>{felt_from_canonical(std::get<0>(k)), felt_from_canonical(std::get<1>(k)), felt_from_canonical(std::get<2>(k))}));
        });
        be.builtin("poseidon", inst.of_block, (uint32_t)inst.keys.size(), step, pi.segments[8].begin_addr, 6, [&](auto &s, uint32_t ti) {
            const PoseidonTrace &t = *inst.traces[ti];
            const U256x3 &k3 = inst.keys[ti];
            const Felt input[3] = {felt_from_canonical(std::get<0>(k3)), felt_from_canonical(std::get<1>(k3)), felt_from_canonical(std::get<2>(k3))};
            for (uint64_t rnd = 0; rnd < 8; ++rnd)
                for (int j = 0; j < 3; ++j) { s.cell(COL_AUXILIARY, 64 * rnd + FULL[j][0], t.full[rnd][j]); s.cell(COL_AUXILIARY, 64 * rnd + FULL[j][1], t.full_sq[rnd][j]); }
            for (uint64_t k = 0; k < 64; ++k) { s.cell(COL_RANGE_CHECK, 8 * k + 3, t.partial[k]); s.cell(COL_RANGE_CHECK, 8 * k + 7, t.partial_sq[k]); }
            for (uint64_t k = 0; k + 61 < t.partial.size(); ++k) { s.cell(COL_AUXILIARY, 16 * k + 6, t.partial[61 + k]); s.cell(COL_AUXILIARY, 16 * k + 14, t.partial_sq[61 + k]); }
            for (int k = 0; k < 3; ++k) { s.pair(NPC_POSEIDON_ADDRS[k], k, input[k]); s.pair(NPC_POSEIDON_ADDRS[3 + k], 3 + k, t.out[k]); }
        });
    }
    // ---- gap fillers (trace.rs:890-925), sorted memory (get_ordered_memory_accesses, utils.rs:112-152)
    be.memory();
}

Inputs check_inputs(const RegisterStates &states, const Mem &mem, const AirPublicInput &pi, const StarknetPrivateInput &priv) {
    const uint64_t num_cycles = states.size();
    if (!num_cycles || (num_cycles & (num_cycles - 1))) fail("the number of cycles must be a power of two");
    if (num_cycles < ECDSA_BUILTIN_RATIO) fail("the starknet layout needs at least 2048 cycles");
    for (int k : {3, 4, 5, 6, 7, 8}) if (!pi.segments[k].present) fail("the starknet layout requires every builtin segment");
    const MemoryEntry *padding = nullptr;
    for (auto &e : pi.public_memory) if (e.address == 1) { padding = &e; break; }
    if (!padding) fail("public memory has no entry at address 1");
    return Inputs{states, mem, pi, priv, num_cycles, num_cycles * CYCLE_HEIGHT, felt_from_canonical(padding->value)};
}

}  // namespace

void starknet_base_trace_into(Felt *const out[9], const RegisterStates &states, const std::vector<U256> &memory,
                              const std::vector<uint8_t> &present, const AirPublicInput &pi, const StarknetPrivateInput &priv,
                              const std::function<void(int)> *column_done) {
    const HostThreadsScope host_threads_scope;              // OpenMP threads by the cgroup's CPU quota (trace_common.hpp)
    const Mem mem{memory, present};
    const Inputs in = check_inputs(states, mem, pi, priv);
    HostBackend be(in, out, column_done);
    be.lap("allocation");
    generate(be, in);
}

// the same columns made in HBM (d_cols: 9 device columns of 16 * cycles felts): trace.bin / memory.bin go up as they are, the cells are
// made by csrc/trace.hip; only what the host needs for the plans is read here (the instructions' offsets for the range-check pool)
void starknet_base_trace_device(ss_ctx *ctx, uint64_t *const d_cols[9], const uint8_t *trace_bin, uint64_t trace_len, const uint8_t *memory_bin, uint64_t memory_len,
                                const std::vector<U256> &memory, const std::vector<uint8_t> &present, const AirPublicInput &pi, const StarknetPrivateInput &priv) {
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

std::vector<std::vector<Felt>> starknet_base_trace(const RegisterStates &states, const std::vector<U256> &memory,
                                                   const std::vector<uint8_t> &present, const AirPublicInput &pi, const StarknetPrivateInput &priv) {
    std::vector<std::vector<Felt>> cols(NUM_COLS);
    Felt *out[NUM_COLS];
    for (int c = 0; c < NUM_COLS; ++c) { cols[c].resize(states.size() * CYCLE_HEIGHT); out[c] = cols[c].data(); }
    starknet_base_trace_into(out, states, memory, present, pi, priv);
    return cols;
}

}  // namespace ssh
