/* ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the arithmetic on Sandstorm's proving hot path.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or
 * call anything under oracle/.  The shipped HIP path never does.
 *
 * Fp = integers mod p, p = 2^251 + 17*2^192 + 1 (the StarkWare prime,
 * reference cli/src/main.rs:25-26).  An element is 4 x u64 little-endian limbs
 * in Montgomery form with R = 2^256 — the byte image the reference hashes
 * (crypto/src/utils.rs:8-22 reads/writes the `.0.0` limbs directly).
 *
 * The field arithmetic itself lives in the un-vendored ark-ff 0.4.2
 * (Cargo.lock:94-95); this file restates the published Montgomery algorithm
 * (CIOS) and is pinned by the known-answer tests listed in oracle/README.md.
 */
#ifndef ORACLE_FP252_H
#define ORACLE_FP252_H
#include <stdint.h>
#include <string.h>

typedef struct { uint64_t l[4]; } fp_t;
typedef unsigned __int128 u128;

static const uint64_t FP_P[4] = {1ull, 0ull, 0ull, 0x0800000000000011ull};
/* -p^{-1} mod 2^64; p == 1 (mod 2^64) so this is 2^64 - 1 */
#define FP_INV 0xffffffffffffffffull

static inline int fp_geq_p(const uint64_t a[4]) {
    for (int i = 3; i >= 0; --i) {
        if (a[i] > FP_P[i]) return 1;
        if (a[i] < FP_P[i]) return 0;
    }
    return 1;
}
static inline void fp_sub_p(uint64_t a[4]) {
    u128 br = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a[i] - FP_P[i] - br;
        a[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
}
static inline fp_t fp_add(fp_t a, fp_t b) {
    fp_t r; u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    /* a,b < p < 2^252 so no carry out of 256 bits */
    if (fp_geq_p(r.l)) fp_sub_p(r.l);
    return r;
}
static inline fp_t fp_sub(fp_t a, fp_t b) {
    fp_t r; u128 br = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a.l[i] - b.l[i] - br;
        r.l[i] = (uint64_t)d; br = (d >> 64) & 1;
    }
    if (br) { u128 c = 0; for (int i = 0; i < 4; ++i) { c += (u128)r.l[i] + FP_P[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
    return r;
}
static inline fp_t fp_neg(fp_t a) { fp_t z = {{0,0,0,0}}; return fp_sub(z, a); }

/* Montgomery product a*b*R^-1 mod p: the 4 x 4 schoolbook product, then four reduction steps that use the prime's shape -
 * p = 1 + P3 2^192 and -p^-1 = -1 mod 2^64, so a step is m = -t_i, t += m (limb i becomes zero) and t += m P3 2^(64 (i + 3)):
 * 20 64 x 64 multiplications where the generic CIOS loop (rounds 1-5) did 32.  The same function, value for value (Montgomery
 * reduction has one canonical result): every known-answer test of oracle/README.md holds it. */
static inline fp_t fp_mul(fp_t a, fp_t b) {
    uint64_t t[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) {
            c += (u128)a.l[j] * b.l[i] + t[i + j];
            t[i + j] = (uint64_t)c; c >>= 64;
        }
        t[i + 4] = (uint64_t)c;
    }
    for (int i = 0; i < 4; ++i) {
        const uint64_t m = 0 - t[i];
        const u128 mp = (u128)m * FP_P[3];
        u128 c = ((u128)t[i] + m) >> 64;
        c += t[i + 1]; t[i + 1] = (uint64_t)c; c >>= 64;
        c += t[i + 2]; t[i + 2] = (uint64_t)c; c >>= 64;
        c += (u128)t[i + 3] + (uint64_t)mp; t[i + 3] = (uint64_t)c; c >>= 64;
        c += (u128)t[i + 4] + (uint64_t)(mp >> 64); t[i + 4] = (uint64_t)c; c >>= 64;
        for (int k = i + 5; k < 9; ++k) { c += t[k]; t[k] = (uint64_t)c; c >>= 64; }
    }
    fp_t r = {{t[4], t[5], t[6], t[7]}};
    if (t[8] || fp_geq_p(r.l)) fp_sub_p(r.l);
    return r;
}
static inline int fp_eq(fp_t a, fp_t b) { return memcmp(&a, &b, sizeof a) == 0; }
static inline int fp_is_zero(fp_t a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }

/* constants / conversions, defined in fp252.c */
extern fp_t FP_ONE;      /* R mod p */
extern fp_t FP_R2;       /* R^2 mod p */
extern fp_t FP_GEN;      /* 3 in Montgomery form: multiplicative generator */
void fp_init(void);
fp_t fp_from_canonical(const uint64_t c[4]);     /* integer < p  -> Montgomery */
void fp_to_canonical(fp_t a, uint64_t c[4]);     /* Montgomery -> integer < p  */
fp_t fp_from_u64(uint64_t v);
fp_t fp_pow_u64(fp_t a, uint64_t e);
fp_t fp_pow(fp_t a, const uint64_t e[4]);
fp_t fp_inv(fp_t a);
/* primitive 2^k-th root of unity 3^((p-1)/2^k) (ark-poly
 * Radix2EvaluationDomain::new(n).group_gen(); reference call site
 * layouts/src/recursive/air.rs:66-67) */
fp_t fp_root_of_unity(unsigned log_n);
/* 32-byte big-endian image of the Montgomery limbs — what the reference's
 * hash functions absorb (crypto/src/hash/keccak.rs:50-58) */
void fp_mont_be_bytes(fp_t a, uint8_t out[32]);
/* big-endian 32 bytes -> integer reduced mod p -> Montgomery
 * (crypto/src/merkle/mixed.rs:152-153: Fp::from(BigUint::from_bytes_be)) */
fp_t fp_from_be_bytes_reduce(const uint8_t in[32]);
/* canonical value as 32 big-endian bytes (PedersenDigest::as_bytes,
 * crypto/src/hash/pedersen.rs:23-28) */
void fp_canonical_be_bytes(fp_t a, uint8_t out[32]);
#endif
