#include "coin.hpp"

#include <algorithm>
#include <cstring>
#include <stdexcept>

#include "../../include/sandstorm_hip.h"

namespace ssh {

static void ok(ss_status s) {
    if (s != SS_OK) throw std::runtime_error(ss_last_error());
}

Digest keccak256(const uint8_t *msg, size_t len) {
    Digest d;
    ok(ss_keccak256_host(msg, len, d.data()));
    return d;
}

// ---- Blake2s-256 (RFC 7693), host
static const uint32_t B2IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const uint8_t B2SIG[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
static inline uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void b2_compress(uint32_t h[8], const uint8_t blk[64], uint64_t t, bool last) {
    uint32_t m[16], v[16];
    for (int i = 0; i < 16; ++i) memcpy(&m[i], blk + 4 * i, 4);
    for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[i + 8] = B2IV[i]; }
    v[12] ^= (uint32_t)t; v[13] ^= (uint32_t)(t >> 32);
    if (last) v[14] = ~v[14];
    auto G = [&](int a, int b, int c, int d, uint32_t x, uint32_t y) {
        v[a] = v[a] + v[b] + x; v[d] = ror(v[d] ^ v[a], 16); v[c] = v[c] + v[d]; v[b] = ror(v[b] ^ v[c], 12);
        v[a] = v[a] + v[b] + y; v[d] = ror(v[d] ^ v[a], 8);  v[c] = v[c] + v[d]; v[b] = ror(v[b] ^ v[c], 7);
    };
    for (int r = 0; r < 10; ++r) {
        const uint8_t *s = B2SIG[r];
        G(0, 4, 8, 12, m[s[0]], m[s[1]]);  G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        G(2, 6, 10, 14, m[s[4]], m[s[5]]); G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        G(0, 5, 10, 15, m[s[8]], m[s[9]]); G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
    for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
}
Digest blake2s256(const uint8_t *msg, size_t len) {
    uint32_t h[8];
    for (int i = 0; i < 8; ++i) h[i] = B2IV[i];
    h[0] ^= 0x01010020u;
    uint64_t t = 0;
    while (len > 64) { t += 64; b2_compress(h, msg, t, false); msg += 64; len -= 64; }
    uint8_t blk[64] = {0};
    if (len) memcpy(blk, msg, len);
    t += len;
    b2_compress(h, blk, t, true);
    Digest d;
    memcpy(d.data(), h, 32);
    return d;
}

// ---- tiny field helpers on transcript values (host Pedersen gives us a Montgomery multiplier-free
// path only for hashing; products of transcript scalars use 256-bit schoolbook + Montgomery here)
typedef unsigned __int128 u128;
static const uint64_t P[4] = {1ull, 0ull, 0ull, 0x0800000000000011ull};
static bool geq_p(const uint64_t a[4]) {
    for (int i = 3; i >= 0; --i) { if (a[i] > P[i]) return true; if (a[i] < P[i]) return false; }
    return true;
}
static void sub_p(uint64_t a[4]) {
    u128 br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a[i] - P[i] - br; a[i] = (uint64_t)d; br = (d >> 64) & 1; }
}
Felt felt_mul(const Felt &a, const Felt &b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)a[j] * b[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        const uint64_t m = 0 - t[0];                       // -p^-1 = -1 mod 2^64
        c = (u128)m * P[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; ++j) { c += (u128)m * P[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    uint64_t r[4] = {t[0], t[1], t[2], t[3]};
    if (t[4] || geq_p(r)) sub_p(r);
    return {r[0], r[1], r[2], r[3]};
}
static const Felt R2 = {0xfffffd737e000401ull, 0x00000001330fffffull, 0xffffffffff6f8000ull, 0x07ffd4ab5e008810ull};
// v 2^256 mod p.  2^256 = 32 2^251 = -(544 2^192 + 32) mod p, so for v < 2^49 (every address, offset, flag and count the trace
// generators convert by the million) the image is p - (544 v 2^192 + 32 v) - the subtrahend is below p and has two non-zero
// limbs: a handful of 64-bit operations where the general path is a Montgomery product with 2^512 mod p
Felt felt_from_u64(uint64_t v) {
    if (v >> 49) return felt_mul(Felt{v, 0, 0, 0}, R2);
    if (!v) return Felt{0, 0, 0, 0};
    return Felt{1ull - 32ull * v, ~0ull, ~0ull, P[3] - 544ull * v - 1ull};      // the low limb borrows (32 v > 1) through limbs 1, 2
}
Felt felt_from_canonical(const Felt &value) { return felt_mul(value, R2); }
Felt felt_pow(const Felt &a, uint64_t e) {
    Felt r = felt_from_u64(1), b = a;
    while (e) { if (e & 1) r = felt_mul(r, b); b = felt_mul(b, b); e >>= 1; }
    return r;
}
// more field helpers on transcript-sized data (host AIRs, trace generation)
Felt felt_inv(const Felt &a) {
    // a^(p-2): p - 2 = 2^251 + 2^196 + (2^192 - 1).  Left to right: bit 251, 55 squarings down to bit 196 (x a), 4 squarings, then the
    // 192 one-bits six at a time (six squarings, x a^63): 251 squarings + 43 products where bit-by-bit takes 251 + 193 - the trace
    // generators invert per elliptic-curve step of a builtin instance, by the thousand per real instance
    const Felt a2 = felt_mul(a, a), a3 = felt_mul(a2, a), a7 = felt_mul(felt_mul(a3, a3), a);
    const Felt a15 = felt_mul(felt_mul(a7, a7), a), a31 = felt_mul(felt_mul(a15, a15), a), a63 = felt_mul(felt_mul(a31, a31), a);
    Felt r = a;
    for (int i = 0; i < 55; ++i) r = felt_mul(r, r);
    r = felt_mul(r, a);
    for (int i = 0; i < 4; ++i) r = felt_mul(r, r);
    for (int g = 0; g < 32; ++g) {
        for (int i = 0; i < 6; ++i) r = felt_mul(r, r);
        r = felt_mul(r, a63);
    }
    return r;
}
Felt felt_neg(const Felt &a) {
    static const uint64_t P[4] = {1ull, 0ull, 0ull, 0x0800000000000011ull};
    if ((a[0] | a[1] | a[2] | a[3]) == 0) return a;
    Felt r;
    unsigned __int128 br = 0;
    for (int i = 0; i < 4; ++i) { unsigned __int128 d = (unsigned __int128)P[i] - a[i] - br; r[i] = (uint64_t)d; br = (d >> 64) & 1; }
    return r;
}
Felt felt_sub(const Felt &a, const Felt &b) {
    // a - b = a + (-b) with one conditional subtraction
    static const uint64_t P[4] = {1ull, 0ull, 0ull, 0x0800000000000011ull};
    Felt nb = felt_neg(b), r;
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (unsigned __int128)a[i] + nb[i]; r[i] = (uint64_t)c; c >>= 64; }
    bool ge = true;
    for (int i = 3; i >= 0; --i) { if (r[i] > P[i]) break; if (r[i] < P[i]) { ge = false; break; } }
    if (ge) { unsigned __int128 br = 0; for (int i = 0; i < 4; ++i) { unsigned __int128 d = (unsigned __int128)r[i] - P[i] - br; r[i] = (uint64_t)d; br = (d >> 64) & 1; } }
    return r;
}
Felt root_of_unity(uint32_t log_n) {
    Felt c = felt_pow(felt_from_u64(3), (1ull << 59) + 17ull);
    for (uint32_t i = 0; i < 192 - log_n; ++i) c = felt_mul(c, c);
    return c;
}

Felt felt_add(const Felt &a, const Felt &b) { return felt_sub(a, felt_neg(b)); }
std::array<uint8_t, 32> mont_be_bytes(const Felt &f) {
    std::array<uint8_t, 32> o;
    for (int i = 0; i < 4; ++i) for (int b = 0; b < 8; ++b) o[i * 8 + b] = (uint8_t)(f[3 - i] >> (56 - 8 * b));
    return o;
}
std::array<uint8_t, 32> canonical_be_bytes(const Felt &f) { return mont_be_bytes(felt_mul(f, Felt{1, 0, 0, 0})); }

// ---- the coin
Digest PublicCoin::hash(const uint8_t *m, size_t n) const { return kind_ == SS_COIN_SOLIDITY ? keccak256(m, n) : blake2s256(m, n); }

void PublicCoin::reseed_with_bytes(const uint8_t *bytes, size_t len) {
    std::vector<uint8_t> buf(32 + len);
    memcpy(buf.data(), digest_.data(), 32);
    for (int i = 31; i >= 0; --i) if (++buf[i] != 0) break;          // U256 digest + 1, big-endian
    if (len) memcpy(buf.data() + 32, bytes, len);
    digest_ = hash(buf.data(), buf.size());
    counter_ = 0;
}
void PublicCoin::reseed_with_field_elements(const std::vector<Felt> &v) {
    if (kind_ == SS_COIN_SOLIDITY) {
        for (const Felt &e : v) { auto b = mont_be_bytes(e); reseed_with_bytes(b.data(), 32); }
    } else {
        Felt cur = {0, 0, 0, 0};                                     // PedersenHashFn::hash_elements
        for (const Felt &e : v) { Felt o; ok(ss_pedersen_hash_host(cur.data(), e.data(), o.data())); cur = o; }
        Felt n = felt_from_u64(v.size()), o;
        ok(ss_pedersen_hash_host(cur.data(), n.data(), o.data()));
        auto b = canonical_be_bytes(o);
        reseed_with_bytes(b.data(), 32);
    }
}
void PublicCoin::reseed_with_field_element_vector(const std::vector<Felt> &v) {
    std::vector<uint8_t> buf(32 * v.size());
    for (size_t i = 0; i < v.size(); ++i) { auto b = mont_be_bytes(v[i]); memcpy(buf.data() + 32 * i, b.data(), 32); }
    reseed_with_bytes(buf.data(), buf.size());
}
void PublicCoin::reseed_with_int(uint64_t v) {
    uint8_t b[8];
    for (int i = 0; i < 8; ++i) b[i] = (uint8_t)(v >> (56 - 8 * i));
    reseed_with_bytes(b, 8);
}
Digest PublicCoin::draw_bytes() {
    uint8_t buf[64];
    memcpy(buf, digest_.data(), 32);
    memset(buf + 32, 0, 24);
    for (int b = 0; b < 8; ++b) buf[56 + b] = (uint8_t)(counter_ >> (56 - 8 * b));
    ++counter_;
    return hash(buf, 64);
}
Felt PublicCoin::draw() {
    const u128 top = (u128)31 * P[3];
    const uint64_t bound[4] = {31, 0, 0, (uint64_t)top};             // 31 p
    for (;;) {
        Digest d = draw_bytes();
        uint64_t v[4];
        for (int i = 0; i < 4; ++i) { uint64_t w = 0; for (int b = 0; b < 8; ++b) w = (w << 8) | d[i * 8 + b]; v[3 - i] = w; }
        bool lt = false;
        for (int i = 3; i >= 0; --i) { if (v[i] < bound[i]) { lt = true; break; } if (v[i] > bound[i]) break; }
        if (!lt) continue;
        while (geq_p(v)) sub_p(v);
        return {v[0], v[1], v[2], v[3]};                             // from_montgomery: the limbs ARE the element
    }
}
std::vector<uint64_t> PublicCoin::draw_queries(size_t max_n, uint64_t domain_size) {
    const size_t want = kind_ == SS_COIN_CAIRO ? ((max_n + 3) / 4) * 4 : max_n;
    std::vector<uint64_t> vals;
    while (vals.size() < want) {
        Digest d = draw_bytes();
        for (int k = 0; k < 4 && vals.size() < want; ++k) {
            uint64_t w = 0;
            for (int b = 0; b < 8; ++b) w = (w << 8) | d[8 * k + b];
            vals.push_back(w % domain_size);
        }
    }
    vals.resize(max_n);
    std::set<uint64_t> s(vals.begin(), vals.end());
    return std::vector<uint64_t>(s.begin(), s.end());
}

// PublicCoin::verify_proof_of_work (crypto/src/public_coin/solidity.rs:143-156, cairo.rs:156-169)
bool verify_proof_of_work(int coin_kind, const Digest &digest, uint32_t bits, uint64_t nonce) {
    if (!bits) return true;
    auto h = [&](const std::vector<uint8_t> &m) { return coin_kind == SS_COIN_SOLIDITY ? keccak256(m.data(), m.size()) : blake2s256(m.data(), m.size()); };
    std::vector<uint8_t> m = {0x01, 0x23, 0x45, 0x67, 0x89, 0xAB, 0xCD, 0xED};
    m.insert(m.end(), digest.begin(), digest.end());
    m.push_back((uint8_t)bits);
    const Digest prefix = h(m);
    std::vector<uint8_t> m2(prefix.begin(), prefix.end());
    for (int i = 7; i >= 0; --i) m2.push_back((uint8_t)(nonce >> (8 * i)));
    const Digest out = h(m2);
    for (uint32_t b = 0; b < bits; ++b) if ((out[b / 8] >> (7 - b % 8)) & 1) return false;
    return true;
}

}  // namespace ssh
