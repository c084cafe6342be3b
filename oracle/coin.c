/* ORACLE — TEST INFRASTRUCTURE ONLY (see fp252.h).
 *
 * Rows C1/C2 of SURVEY.md §8(a): the two Fiat-Shamir coins.
 *   kind 0: SolidityVerifierPublicCoin (crypto/src/public_coin/solidity.rs:36-161)
 *   kind 1: CairoVerifierPublicCoin    (crypto/src/public_coin/cairo.rs:42-174)
 * Pinned by draw_matches_solidity_verifier (solidity.rs:172-193) and
 * reseed_with_field_element (cairo.rs:189-208) — tests/golden/coins.json.
 */
#include "oracle.h"
#include <stdlib.h>

static void coin_hash(int kind, const uint8_t *m, size_t n, uint8_t out[32]) {
    if (kind == 0) or_keccak256(m, n, out); else or_blake2s256(m, n, out);
}

void or_coin_new(or_coin *c, int kind, const uint8_t digest[32]) {
    c->kind = kind; memcpy(c->digest, digest, 32); c->counter = 0;
}

/* solidity.rs:37-44 / cairo.rs:43-50: digest <- H(be32(digest + 1) || bytes) */
void or_coin_reseed_bytes(or_coin *c, const uint8_t *bytes, size_t len) {
    uint8_t *buf = (uint8_t *)malloc(32 + len);
    memcpy(buf, c->digest, 32);
    for (int i = 31; i >= 0; --i) { if (++buf[i] != 0) break; } /* wraps like U256 add */
    memcpy(buf + 32, bytes, len);
    coin_hash(c->kind, buf, 32 + len, c->digest);
    c->counter = 0;
    free(buf);
}

static void coin_draw_bytes(or_coin *c, uint8_t out[32]) {
    uint8_t buf[64];
    memcpy(buf, c->digest, 32);
    memset(buf + 32, 0, 24);
    for (int b = 0; b < 8; ++b) buf[56 + b] = (uint8_t)(c->counter >> (56 - 8 * b));
    c->counter += 1;
    coin_hash(c->kind, buf, 64, out);
}

void or_coin_reseed_felts(or_coin *c, const fp_t *v, size_t n) {
    if (c->kind == 0) {
        /* solidity.rs:66-71: one reseed per element, Montgomery BE bytes */
        for (size_t i = 0; i < n; ++i) { uint8_t b[32]; fp_mont_be_bytes(v[i], b); or_coin_reseed_bytes(c, b, 32); }
    } else {
        /* cairo.rs:76-80: Pedersen chain hash, canonical BE bytes */
        uint8_t b[32]; fp_canonical_be_bytes(or_pedersen_hash_elements(v, n), b);
        or_coin_reseed_bytes(c, b, 32);
    }
}
void or_coin_reseed_felt_vector(or_coin *c, const fp_t *v, size_t n) {
    uint8_t *buf = (uint8_t *)malloc(32 * (n ? n : 1));
    for (size_t i = 0; i < n; ++i) fp_mont_be_bytes(v[i], buf + 32 * i);
    or_coin_reseed_bytes(c, buf, 32 * n);
    free(buf);
}
void or_coin_reseed_int(or_coin *c, uint64_t v) {
    uint8_t b[8];
    for (int i = 0; i < 8; ++i) b[i] = (uint8_t)(v >> (56 - 8 * i));
    or_coin_reseed_bytes(c, b, 8);
}

/* draw(): reject samples >= 31p, then from_montgomery = raw limbs mod p
 * (solidity.rs:93-102, crypto/src/utils.rs:8-12) */
fp_t or_coin_draw(or_coin *c) {
    for (;;) {
        uint8_t d[32]; coin_draw_bytes(c, d);
        uint64_t v[4];
        for (int i = 0; i < 4; ++i) { uint64_t w = 0; for (int b = 0; b < 8; ++b) w = (w << 8) | d[i * 8 + b]; v[3 - i] = w; }
        /* bound = 31 p = 31 + 31*(2^59+17)*2^192 */
        u128 top = (u128)31 * FP_P[3];
        uint64_t bound[4] = {31, 0, 0, (uint64_t)top};
        int lt = 0;
        for (int i = 3; i >= 0; --i) { if (v[i] < bound[i]) { lt = 1; break; } if (v[i] > bound[i]) break; }
        if (!lt) continue;
        while (fp_geq_p(v)) fp_sub_p(v);
        fp_t r = {{v[0], v[1], v[2], v[3]}};
        return r;
    }
}

void or_coin_draw_queries(or_coin *c, size_t max_n, uint64_t domain_size, uint64_t *out) {
    /* solidity.rs:104-122 takes max_n ints; cairo.rs:108-131 draws a multiple
     * of 4 and truncates.  Both consume whole 32-byte draws lazily. */
    size_t want = c->kind == 1 ? ((max_n + 3) / 4) * 4 : max_n;
    size_t got = 0;
    while (got < want) {
        uint8_t d[32]; coin_draw_bytes(c, d);
        for (int k = 0; k < 4 && got < want; ++k, ++got) {
            uint64_t w = 0; for (int b = 0; b < 8; ++b) w = (w << 8) | d[8 * k + b];
            if (got < max_n) out[got] = w % domain_size;
        }
    }
}

static unsigned leading_zeros(const uint8_t h[32]) {
    unsigned z = 0;
    for (int i = 0; i < 32; ++i) {
        if (h[i] == 0) { z += 8; continue; }
        uint8_t b = h[i]; while (!(b & 0x80)) { ++z; b <<= 1; }
        break;
    }
    return z;
}
static void pow_prefix(const or_coin *c, unsigned bits, uint8_t prefix[32]) {
    uint8_t buf[41];
    const uint64_t magic = 0x0123456789ABCDEDull;
    for (int i = 0; i < 8; ++i) buf[i] = (uint8_t)(magic >> (56 - 8 * i));
    memcpy(buf + 8, c->digest, 32);
    buf[40] = (uint8_t)bits;
    coin_hash(c->kind, buf, 41, prefix);
}
int or_coin_verify_pow(const or_coin *c, unsigned bits, uint64_t nonce) {
    uint8_t buf[40], h[32];
    pow_prefix(c, bits, buf);
    for (int i = 0; i < 8; ++i) buf[32 + i] = (uint8_t)(nonce >> (56 - 8 * i));
    coin_hash(c->kind, buf, 40, h);
    return leading_zeros(h) >= bits;
}
/* non-parallel `find` semantics (solidity.rs:138): smallest nonce in 1.. */
uint64_t or_coin_grind(const or_coin *c, unsigned bits) {
    uint8_t buf[40], h[32];
    pow_prefix(c, bits, buf);
    for (uint64_t nonce = 1;; ++nonce) {
        for (int i = 0; i < 8; ++i) buf[32 + i] = (uint8_t)(nonce >> (56 - 8 * i));
        coin_hash(c->kind, buf, 40, h);
        if (leading_zeros(h) >= bits) return nonce;
    }
}
