/* ORACLE — TEST INFRASTRUCTURE ONLY (see fp252.h).
 *
 * Keccak-256 (sha3 crate 0.10.8, Cargo.lock:1533-1534) and Blake2s-256
 * (blake2 crate 0.10.6, Cargo.lock:308-309) restated from their published
 * specifications (Keccak team submission v3 with 0x01 padding; RFC 7693), plus
 * the reference's element/row hashing rules:
 *   - hash_elements: concat of to_montgomery(e).to_be_bytes::<32>()
 *     (crypto/src/hash/keccak.rs:50-58, blake2s.rs:50-61)
 *   - masks: Keccak keeps the FIRST n bytes (hash/mod.rs:5-13), Blake2s keeps
 *     the LAST n bytes (hash/mod.rs:15-23)
 *   - hash_rows over a column-major matrix (crypto/src/merkle/utils.rs:19-46)
 * Pinned by draw_matches_solidity_verifier (public_coin/solidity.rs:172-193)
 * and reseed_with_field_element (public_coin/cairo.rs:189-208).
 */
#include "oracle.h"
#include <stdlib.h>

/* ------------------------------------------------------------- Keccak-256 */
static const uint64_t KRC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull,
    0x000000000000808bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
    0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
    0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull,
    0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
static const int KROT[25] = {0,  1,  62, 28, 27, 36, 44, 6,  55, 20, 3,  10, 43,
                             25, 39, 41, 45, 15, 21, 8,  18, 2,  61, 56, 14};
static inline uint64_t rol64(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

static void keccak_f(uint64_t s[25]) {
    for (int r = 0; r < 24; ++r) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; ++x) c[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
        for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; ++i) s[i] ^= d[i % 5];
        /* rho + pi: B[y][2x+3y] = rot(A[x][y]) with index = x + 5y */
        for (int x = 0; x < 5; ++x)
            for (int y = 0; y < 5; ++y)
                b[y + 5 * ((2 * x + 3 * y) % 5)] = rol64(s[x + 5 * y], KROT[x + 5 * y]);
        for (int y = 0; y < 5; ++y)
            for (int x = 0; x < 5; ++x)
                s[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        s[0] ^= KRC[r];
    }
}

void or_keccak256(const uint8_t *msg, size_t len, uint8_t out[32]) {
    uint64_t s[25];
    memset(s, 0, sizeof s);
    const size_t rate = 136;
    while (len >= rate) {
        for (int i = 0; i < 17; ++i) { uint64_t w; memcpy(&w, msg + 8 * i, 8); s[i] ^= w; }
        keccak_f(s);
        msg += rate; len -= rate;
    }
    uint8_t blk[136];
    memset(blk, 0, sizeof blk);
    memcpy(blk, msg, len);
    blk[len] ^= 0x01;
    blk[135] ^= 0x80;
    for (int i = 0; i < 17; ++i) { uint64_t w; memcpy(&w, blk + 8 * i, 8); s[i] ^= w; }
    keccak_f(s);
    memcpy(out, s, 32);
}

/* ------------------------------------------------------------ Blake2s-256 */
static const uint32_t B2IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                                 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const uint8_t B2SIG[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15},
    {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4},
    {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13},
    {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11},
    {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5},
    {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
static inline uint32_t ror32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static void b2s_compress(uint32_t h[8], const uint8_t blk[64], uint64_t t, int last) {
    uint32_t m[16], v[16];
    for (int i = 0; i < 16; ++i) memcpy(&m[i], blk + 4 * i, 4);
    for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[i + 8] = B2IV[i]; }
    v[12] ^= (uint32_t)t; v[13] ^= (uint32_t)(t >> 32);
    if (last) v[14] = ~v[14];
#define B2G(a, b, c, d, x, y)                                   \
    v[a] = v[a] + v[b] + (x); v[d] = ror32(v[d] ^ v[a], 16);     \
    v[c] = v[c] + v[d];       v[b] = ror32(v[b] ^ v[c], 12);     \
    v[a] = v[a] + v[b] + (y); v[d] = ror32(v[d] ^ v[a], 8);      \
    v[c] = v[c] + v[d];       v[b] = ror32(v[b] ^ v[c], 7);
    for (int r = 0; r < 10; ++r) {
        const uint8_t *s = B2SIG[r];
        B2G(0, 4, 8, 12, m[s[0]], m[s[1]]);
        B2G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        B2G(2, 6, 10, 14, m[s[4]], m[s[5]]);
        B2G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        B2G(0, 5, 10, 15, m[s[8]], m[s[9]]);
        B2G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        B2G(2, 7, 8, 13, m[s[12]], m[s[13]]);
        B2G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
#undef B2G
    for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
}

void or_blake2s256(const uint8_t *msg, size_t len, uint8_t out[32]) {
    uint32_t h[8];
    for (int i = 0; i < 8; ++i) h[i] = B2IV[i];
    h[0] ^= 0x01010020u; /* digest length 32, no key, fanout = depth = 1 */
    uint64_t t = 0;
    while (len > 64) {
        t += 64;
        b2s_compress(h, msg, t, 0);
        msg += 64; len -= 64;
    }
    uint8_t blk[64];
    memset(blk, 0, sizeof blk);
    memcpy(blk, msg, len);
    t += len;
    b2s_compress(h, blk, t, 1);
    memcpy(out, h, 32);
}

/* ------------------------------------------------------------- SHA-256
 * FIPS 180-4 sections 4.1.2, 4.2.2, 5.1.1, 5.3.3, 6.2.  The reference names ministark's Sha256HashFn for its 64-bit-field claim
 * (cli/src/main.rs:105,119-120; the `sha2` crate behind it is the same function): pinned by the standard's own vectors
 * (tests/test_oracle_golden.py) - the one part of that claim a public document pins. */
static const uint32_t SHA_K[64] = {
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u,
    0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
    0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u,
    0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
    0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
    0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
static uint32_t sha_rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void sha256_block(uint32_t h[8], const uint8_t blk[64]) {
    uint32_t w[64];
    for (int t = 0; t < 16; ++t) w[t] = ((uint32_t)blk[4 * t] << 24) | ((uint32_t)blk[4 * t + 1] << 16) | ((uint32_t)blk[4 * t + 2] << 8) | blk[4 * t + 3];
    for (int t = 16; t < 64; ++t) {
        const uint32_t s0 = sha_rotr(w[t - 15], 7) ^ sha_rotr(w[t - 15], 18) ^ (w[t - 15] >> 3);
        const uint32_t s1 = sha_rotr(w[t - 2], 17) ^ sha_rotr(w[t - 2], 19) ^ (w[t - 2] >> 10);
        w[t] = w[t - 16] + s0 + w[t - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int t = 0; t < 64; ++t) {
        const uint32_t t1 = hh + (sha_rotr(e, 6) ^ sha_rotr(e, 11) ^ sha_rotr(e, 25)) + ((e & f) ^ (~e & g)) + SHA_K[t] + w[t];
        const uint32_t t2 = (sha_rotr(a, 2) ^ sha_rotr(a, 13) ^ sha_rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
void or_sha256(const uint8_t *msg, size_t len, uint8_t out[32]) {
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    const uint64_t bits = (uint64_t)len * 8;
    while (len >= 64) { sha256_block(h, msg); msg += 64; len -= 64; }
    uint8_t blk[128];
    memset(blk, 0, sizeof blk);
    memcpy(blk, msg, len);
    blk[len] = 0x80;
    const size_t total = len + 9 <= 64 ? 64 : 128;
    for (int i = 0; i < 8; ++i) blk[total - 1 - i] = (uint8_t)(bits >> (8 * i));
    sha256_block(h, blk);
    if (total == 128) sha256_block(h, blk + 64);
    for (int i = 0; i < 8; ++i) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
}

/* --------------------------------------------------- reference hash rules */
void or_apply_mask(int kind, uint8_t d[32]) {
    if (kind == OR_HASH_KECCAK_M20) memset(d + 20, 0, 12);      /* hash/mod.rs:5-13  */
    else if (kind == OR_HASH_BLAKE2S_M20) memset(d, 0, 12);     /* hash/mod.rs:15-23 */
}
void or_hash_bytes(int kind, const uint8_t *msg, size_t len, uint8_t out[32]) {
    if (kind == OR_HASH_KECCAK || kind == OR_HASH_KECCAK_M20) or_keccak256(msg, len, out);
    else or_blake2s256(msg, len, out);
    or_apply_mask(kind, out);
}
/* H::hash_elements */
void or_hash_elements(int kind, const fp_t *e, size_t n, uint8_t out[32]) {
    uint8_t *buf = (uint8_t *)malloc(32 * (n ? n : 1));
    for (size_t i = 0; i < n; ++i) fp_mont_be_bytes(e[i], buf + 32 * i);
    or_hash_bytes(kind, buf, 32 * n, out);
    free(buf);
}
/* H::merge(n0, n1) */
void or_hash_merge(int kind, const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) {
    uint8_t buf[64];
    memcpy(buf, a, 32); memcpy(buf + 32, b, 32);
    or_hash_bytes(kind, buf, 64, out);
}
/* crypto/src/merkle/utils.rs:19-46 — one digest per matrix row */
void or_hash_rows(int kind, const fp_t *const *cols, size_t ncols, size_t nrows, uint8_t *out) {
#pragma omp parallel for schedule(static) if (nrows >= 1024)
    for (size_t r = 0; r < nrows; ++r) {
        fp_t row[64];
        for (size_t c = 0; c < ncols; ++c) row[c] = cols[c][r];
        or_hash_elements(kind, row, ncols, out + 32 * r);
    }
}
