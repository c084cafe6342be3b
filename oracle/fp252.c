/* ORACLE — TEST INFRASTRUCTURE ONLY (see fp252.h). */
#include "fp252.h"

fp_t FP_ONE, FP_R2, FP_GEN;
static int fp_ready = 0;

static fp_t fp_double_raw(fp_t a) { return fp_add(a, a); }

void fp_init(void) {
    if (fp_ready) return;
    /* R mod p = 2^256 mod p by 256 modular doublings of 1; R^2 by 256 more. */
    fp_t x = {{1, 0, 0, 0}};
    for (int i = 0; i < 256; ++i) x = fp_double_raw(x);
    FP_ONE = x;
    for (int i = 0; i < 256; ++i) x = fp_double_raw(x);
    FP_R2 = x;
    FP_GEN = fp_from_u64(3);
    fp_ready = 1;
}

fp_t fp_from_canonical(const uint64_t c[4]) {
    fp_t a = {{c[0], c[1], c[2], c[3]}};
    return fp_mul(a, FP_R2);
}
void fp_to_canonical(fp_t a, uint64_t c[4]) {
    fp_t one = {{1, 0, 0, 0}};
    fp_t r = fp_mul(a, one);
    memcpy(c, r.l, 32);
}
fp_t fp_from_u64(uint64_t v) {
    uint64_t c[4] = {v, 0, 0, 0};
    /* FP_R2 may not be final while fp_init runs FP_GEN last; callers after init only */
    return fp_from_canonical(c);
}
fp_t fp_pow(fp_t a, const uint64_t e[4]) {
    fp_t r = FP_ONE;
    for (int i = 255; i >= 0; --i) {
        r = fp_mul(r, r);
        if ((e[i >> 6] >> (i & 63)) & 1) r = fp_mul(r, a);
    }
    return r;
}
fp_t fp_pow_u64(fp_t a, uint64_t e) {
    uint64_t ee[4] = {e, 0, 0, 0};
    return fp_pow(a, ee);
}
fp_t fp_inv(fp_t a) {
    /* a^(p-2) */
    /* p[0] = 1, so p - 2 borrows through the three low limbs */
    uint64_t e[4] = {0xffffffffffffffffull, 0xffffffffffffffffull, 0xffffffffffffffffull, FP_P[3] - 1};
    return fp_pow(a, e);
}
fp_t fp_root_of_unity(unsigned log_n) {
    /* (p-1) = 2^192 * (2^59 + 17); exponent (p-1) >> log_n, log_n <= 192 */
    uint64_t pm1[4] = {0, 0, 0, FP_P[3]};
    uint64_t e[4] = {0, 0, 0, 0};
    unsigned ws = log_n / 64, bs = log_n % 64;
    for (int i = 0; i < 4; ++i) {
        unsigned src = i + ws;
        uint64_t lo = src < 4 ? pm1[src] : 0, hi = (src + 1) < 4 ? pm1[src + 1] : 0;
        e[i] = bs ? ((lo >> bs) | (hi << (64 - bs))) : lo;
    }
    return fp_pow(FP_GEN, e);
}
void fp_mont_be_bytes(fp_t a, uint8_t out[32]) {
    for (int i = 0; i < 4; ++i)
        for (int b = 0; b < 8; ++b) out[i * 8 + b] = (uint8_t)(a.l[3 - i] >> (56 - 8 * b));
}
void fp_canonical_be_bytes(fp_t a, uint8_t out[32]) {
    uint64_t c[4]; fp_to_canonical(a, c);
    fp_t t = {{c[0], c[1], c[2], c[3]}};
    fp_mont_be_bytes(t, out);
}
fp_t fp_from_be_bytes_reduce(const uint8_t in[32]) {
    uint64_t c[4];
    for (int i = 0; i < 4; ++i) {
        uint64_t v = 0;
        for (int b = 0; b < 8; ++b) v = (v << 8) | in[i * 8 + b];
        c[3 - i] = v;
    }
    /* reduce mod p: value < 2^256 < 32p, subtract while >= p */
    while (fp_geq_p(c)) fp_sub_p(c);
    return fp_from_canonical(c);
}

/* bench.py's cpu_baseline leg: `iters` dependent Montgomery products on the calling core -> the last value (so that nothing is
 * optimised away); the caller times it.  What the port's products cost, to be read beside ark-ff's (the reference's field). */
fp_t or_mulmod_chain(fp_t x, fp_t y, uint64_t iters) {
    for (uint64_t i = 0; i < iters; ++i) x = fp_mul(x, y);
    return x;
}

void or_pedersen_init(void);
void or_init(void) { fp_init(); or_pedersen_init(); }
