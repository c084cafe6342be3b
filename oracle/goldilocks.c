/* ORACLE — TEST INFRASTRUCTURE ONLY (see fp252.h).
 *
 * The 64-bit field variant (BASELINE.json configs[4]): p = 2^64 - 2^32 + 1 ("Goldilocks") and its
 * cubic extension Fq3 = Fp[X] / (X^3 - 2), the pair the reference instantiates for its experimental claim
 * (cli/src/main.rs:103-133: ministark_gpu::fields::p18446744069414584321::ark::{Fp, Fq3} with
 * layouts::plain::AirConfig<Fp, Fq3>).  The field crate is un-vendored and the reference holds NO test vector, proof or
 * constant for this field: PARITY UNPINNED.  What is restated here are the mathematical definitions, with the same
 * conventions as the 252-bit path (natural-order evaluation at offset * w^k, w = g^((p-1)/n), g = 7 the smallest generator
 * of the multiplicative group; fold = interpolate the 2^k points of a row and evaluate at alpha), so that the HIP kernels
 * have a bit-exact checker.  The non-residue 2 of Fq3 is an assumption (2 is a cubic non-residue mod p).
 *
 * Elements are plain u64 < p here and in the kernels: every operation on this path is linear in the data (NTT, LDE, FRI
 * fold multiply data by coefficients the code owns), so images in Montgomery form (arkworks' in-memory Fp64) pass through
 * unchanged - the kernels never multiply two data values.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GL_P 0xFFFFFFFF00000001ull

static inline uint64_t gl_add(uint64_t a, uint64_t b) { unsigned __int128 s = (unsigned __int128)a + b; return (uint64_t)(s >= GL_P ? s - GL_P : s); }
static inline uint64_t gl_sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a + (GL_P - b); }
static inline uint64_t gl_mul(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) % GL_P); }
static uint64_t gl_pow(uint64_t a, uint64_t e) { uint64_t r = 1; while (e) { if (e & 1) r = gl_mul(r, a); a = gl_mul(a, a); e >>= 1; } return r; }
static uint64_t gl_inv(uint64_t a) { return gl_pow(a, GL_P - 2); }
uint64_t or_gl_root_of_unity(unsigned log_n) { return gl_pow(7, (GL_P - 1) >> log_n); }
uint64_t or_gl_mul(uint64_t a, uint64_t b) { return gl_mul(a % GL_P, b % GL_P); }
uint64_t or_gl_pow(uint64_t a, uint64_t e) { return gl_pow(a % GL_P, e); }

static void gl_ntt_core(uint64_t *a, unsigned log_n, uint64_t w) {
    const size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; ++i) {                          /* bit-reversal, then decimation in time */
        size_t j = 0;
        for (unsigned b = 0; b < log_n; ++b) j |= ((i >> b) & 1) << (log_n - 1 - b);
        if (i < j) { uint64_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (unsigned s = 0; s < log_n; ++s) {
        const size_t half = (size_t)1 << s;
        const uint64_t ws = gl_pow(w, n >> (s + 1));
#pragma omp parallel for schedule(static) if (n >= 65536)
        for (size_t blk = 0; blk < n; blk += 2 * half) {
            uint64_t t = 1;
            for (size_t j = 0; j < half; ++j) {
                const uint64_t u = a[blk + j], v = gl_mul(a[blk + j + half], t);
                a[blk + j] = gl_add(u, v);
                a[blk + j + half] = gl_sub(u, v);
                t = gl_mul(t, ws);
            }
        }
    }
}
/* coefficients -> evaluations on offset * <w_n>, natural order */
void or_gl_ntt_forward(uint64_t *a, unsigned log_n, uint64_t offset) {
    const size_t n = (size_t)1 << log_n;
    if (offset != 1) { uint64_t g = 1; for (size_t i = 0; i < n; ++i) { a[i] = gl_mul(a[i], g); g = gl_mul(g, offset); } }
    gl_ntt_core(a, log_n, or_gl_root_of_unity(log_n));
}
/* evaluations on offset * <w_n> (natural order) -> coefficients */
void or_gl_ntt_inverse(uint64_t *a, unsigned log_n, uint64_t offset) {
    const size_t n = (size_t)1 << log_n;
    gl_ntt_core(a, log_n, gl_inv(or_gl_root_of_unity(log_n)));
    uint64_t g = gl_inv((uint64_t)n % GL_P);
    const uint64_t ginv = gl_inv(offset);
    for (size_t i = 0; i < n; ++i) { a[i] = gl_mul(a[i], g); g = gl_mul(g, ginv); }
}
/* one column of a low-degree extension: interpolate over <w_n>, evaluate over offset * <w_{n << log_blowup}> */
void or_gl_lde(const uint64_t *in, unsigned log_n, unsigned log_blowup, uint64_t offset, uint64_t *evals_out, uint64_t *coeffs_out) {
    const size_t n = (size_t)1 << log_n, N = n << log_blowup;
    uint64_t *c = coeffs_out ? coeffs_out : (uint64_t *)malloc(8 * n);
    memcpy(c, in, 8 * n);
    or_gl_ntt_inverse(c, log_n, 1);
    memcpy(evals_out, c, 8 * n);
    memset(evals_out + n, 0, 8 * (N - n));
    or_gl_ntt_forward(evals_out, log_n + log_blowup, offset);
    if (!coeffs_out) free(c);
}

/* ---- Fq3 = Fp[X] / (X^3 - 2): (a0, a1, a2) = a0 + a1 X + a2 X^2 */
typedef struct { uint64_t c[3]; } gl3_t;
static gl3_t gl3_mul(gl3_t a, gl3_t b) {
    gl3_t r;
    const uint64_t d0 = gl_mul(a.c[0], b.c[0]), d1 = gl_add(gl_mul(a.c[0], b.c[1]), gl_mul(a.c[1], b.c[0]));
    const uint64_t d2 = gl_add(gl_add(gl_mul(a.c[0], b.c[2]), gl_mul(a.c[1], b.c[1])), gl_mul(a.c[2], b.c[0]));
    const uint64_t d3 = gl_add(gl_mul(a.c[1], b.c[2]), gl_mul(a.c[2], b.c[1])), d4 = gl_mul(a.c[2], b.c[2]);
    r.c[0] = gl_add(d0, gl_add(d3, d3));            /* X^3 = 2 */
    r.c[1] = gl_add(d1, gl_add(d4, d4));
    r.c[2] = d2;
    return r;
}
static gl3_t gl3_add(gl3_t a, gl3_t b) { gl3_t r = {{gl_add(a.c[0], b.c[0]), gl_add(a.c[1], b.c[1]), gl_add(a.c[2], b.c[2])}}; return r; }
static gl3_t gl3_scale(gl3_t a, uint64_t s) { gl3_t r = {{gl_mul(a.c[0], s), gl_mul(a.c[1], s), gl_mul(a.c[2], s)}}; return r; }
void or_gl3_mul(const uint64_t a[3], const uint64_t b[3], uint64_t out[3]) {
    gl3_t x, y; memcpy(&x, a, 24); memcpy(&y, b, 24);
    gl3_t r = gl3_mul(x, y); memcpy(out, &r, 24);
}

/* One FRI layer over Fq3-valued evaluations on the Fp domain offset * <w_L> (natural order, elements interleaved
 * [L][3]): row j = {evals[j + k L/fold]}, out[j] = (degree < fold interpolant of row j)(alpha), alpha in Fq3; with
 * unnormalised != 0 the value is multiplied by fold (the 252-bit path's SS_FRI_UNNORMALISED).  Definition: the size-fold
 * inverse transform of the row over x_j * <w_fold>, then Horner in alpha. */
void or_gl3_fri_fold(const uint64_t *evals, unsigned log_len, unsigned fold, const uint64_t alpha[3], uint64_t offset,
                     int unnormalised, uint64_t *out) {
    const size_t L = (size_t)1 << log_len, rows = L / fold;
    unsigned log_fold = 0;
    while ((1u << log_fold) < fold) ++log_fold;
    const uint64_t wL = or_gl_root_of_unity(log_len);
    gl3_t al; memcpy(&al, alpha, 24);
#pragma omp parallel for schedule(static) if (rows >= 256)
    for (size_t j = 0; j < rows; ++j) {
        const uint64_t xj = gl_mul(offset, gl_pow(wL, j));
        uint64_t comp[3][16];
        for (unsigned k = 0; k < fold; ++k) for (int c = 0; c < 3; ++c) comp[c][k] = evals[3 * (j + k * rows) + c];
        gl3_t coef[16];
        for (int c = 0; c < 3; ++c) or_gl_ntt_inverse(comp[c], log_fold, xj);          /* the row's domain is x_j * <w_fold> */
        for (unsigned k = 0; k < fold; ++k) for (int c = 0; c < 3; ++c) coef[k].c[c] = comp[c][k];
        gl3_t acc = coef[fold - 1];
        for (unsigned k = fold - 1; k-- > 0;) acc = gl3_add(gl3_mul(acc, al), coef[k]);
        if (unnormalised) acc = gl3_scale(acc, fold);
        memcpy(out + 3 * j, &acc, 24);
    }
}

/* ---- DEEP over Fq3: the definitions, term by term -----------------------------------------------------------------------
 * (the kernels use one shifted inverse table and a norm-based inverse; here every denominator is inverted on its own, by
 * Fermat in Fq3: a^-1 = a^(p^3 - 2)) */
static gl3_t gl3_sub(gl3_t a, gl3_t b) { gl3_t r = {{gl_sub(a.c[0], b.c[0]), gl_sub(a.c[1], b.c[1]), gl_sub(a.c[2], b.c[2])}}; return r; }
static gl3_t gl3_inv(gl3_t a) {
    /* p^3 - 2 as a 192-bit integer, little-endian 64-bit words */
    unsigned __int128 p = GL_P;
    unsigned __int128 p2 = p * p;                                    /* < 2^128 */
    uint64_t w[3];
    {   /* p2 * p */
        const uint64_t lo = (uint64_t)p2, hi = (uint64_t)(p2 >> 64);
        unsigned __int128 t0 = (unsigned __int128)lo * GL_P, t1 = (unsigned __int128)hi * GL_P + (uint64_t)(t0 >> 64);
        w[0] = (uint64_t)t0; w[1] = (uint64_t)t1; w[2] = (uint64_t)(t1 >> 64);
    }
    /* minus 2 (w[0] = p^3 mod 2^64 = 1: borrow) */
    if (w[0] >= 2) w[0] -= 2; else { w[0] = w[0] - 2; if (w[1]-- == 0) w[2]--; }
    gl3_t r = {{1, 0, 0}};
    for (int k = 191; k >= 0; --k) {
        r = gl3_mul(r, r);
        if ((w[k / 64] >> (k % 64)) & 1) r = gl3_mul(r, a);
    }
    return r;
}
void or_gl3_inv(const uint64_t a[3], uint64_t out[3]) { gl3_t x; memcpy(&x, a, 24); gl3_t r = gl3_inv(x); memcpy(out, &r, 24); }

/* out[3 j ..) = P_{cell_col[j]}(z w_n^{cell_off[j]}) for NATURAL-order coefficient columns (Horner in Fq3) */
void or_gl3_ood_eval(const uint64_t *const *coeffs, unsigned log_n, const uint32_t *cell_col, const uint32_t *cell_off, unsigned ncells,
                     const uint64_t z[3], uint64_t *out) {
    const size_t n = (size_t)1 << log_n;
    const uint64_t wn = or_gl_root_of_unity(log_n);
    gl3_t zz; memcpy(&zz, z, 24);
    for (unsigned j = 0; j < ncells; ++j) {
        const gl3_t pt = gl3_scale(zz, gl_pow(wn, cell_off[j] & (n - 1)));
        const uint64_t *c = coeffs[cell_col[j]];
        gl3_t acc = {{0, 0, 0}};
        for (size_t k = n; k-- > 0;) { acc = gl3_mul(acc, pt); acc.c[0] = gl_add(acc.c[0], c[k]); }
        memcpy(out + 3 * (size_t)j, &acc, 24);
    }
}

/* out[i] (interleaved [N][3]) = sum_j coeff_t[j] (T_{col_j}[i] - ood_t[j]) / (x_i - z w_n^{off_j}) + sum_k coeff_c[k] (H_k[i] - ood_c[k]) / (x_i - zc),
 * x_i = offset * w_N^i, N = n << log_blowup */
void or_gl3_deep_compose(const uint64_t *const *trace, const uint64_t *const *comp, unsigned ncomp, unsigned log_n, unsigned log_blowup,
                         uint64_t offset, const uint32_t *mask_col, const uint32_t *mask_off, unsigned nmask, const uint64_t *ood_t,
                         const uint64_t *coeff_t, const uint64_t *ood_c, const uint64_t *coeff_c, const uint64_t z[3], const uint64_t zc[3],
                         uint64_t *out) {
    const size_t n = (size_t)1 << log_n, N = n << log_blowup;
    const uint64_t wn = or_gl_root_of_unity(log_n), wN = or_gl_root_of_unity(log_n + log_blowup);
    gl3_t zz, zcc = {{0, 0, 0}};
    memcpy(&zz, z, 24);
    if (ncomp) memcpy(&zcc, zc, 24);
#pragma omp parallel for schedule(static) if (N >= 64)
    for (size_t i = 0; i < N; ++i) {
        const uint64_t x = gl_mul(offset, gl_pow(wN, i));
        gl3_t acc = {{0, 0, 0}};
        for (unsigned j = 0; j < nmask; ++j) {
            gl3_t den = gl3_scale(zz, gl_pow(wn, mask_off[j] & (n - 1)));
            den.c[0] = gl_sub(x, den.c[0]); den.c[1] = gl_sub(0, den.c[1]); den.c[2] = gl_sub(0, den.c[2]);
            gl3_t num, o, c;
            memcpy(&o, ood_t + 3 * (size_t)j, 24); memcpy(&c, coeff_t + 3 * (size_t)j, 24);
            num.c[0] = gl_sub(trace[mask_col[j]][i], o.c[0]); num.c[1] = gl_sub(0, o.c[1]); num.c[2] = gl_sub(0, o.c[2]);
            acc = gl3_add(acc, gl3_mul(gl3_mul(c, num), gl3_inv(den)));
        }
        if (ncomp) {
            gl3_t den = {{gl_sub(x, zcc.c[0]), gl_sub(0, zcc.c[1]), gl_sub(0, zcc.c[2])}};
            const gl3_t dinv = gl3_inv(den);
            for (unsigned k = 0; k < ncomp; ++k) {
                gl3_t num, o, c;
                memcpy(&o, ood_c + 3 * (size_t)k, 24); memcpy(&c, coeff_c + 3 * (size_t)k, 24);
                num.c[0] = gl_sub(comp[k][i], o.c[0]); num.c[1] = gl_sub(0, o.c[1]); num.c[2] = gl_sub(0, o.c[2]);
                acc = gl3_add(acc, gl3_mul(gl3_mul(c, num), dinv));
            }
        }
        memcpy(out + 3 * i, &acc, 24);
    }
}

/* ---- the constraint program over Fq3 (include/sandstorm_hip.h ss_air_program; ss_eval_quotient_gl64x3): accumulators, slots
 * and constants in Fq3, trace cells / tables / x in Fp.  One point at a time. */
void or_gl3_eval_program(const uint32_t *code, uint32_t n_instr, const uint64_t *consts3, uint32_t n_slots, const uint64_t *tables,
                         const uint32_t *table_desc, const uint64_t *const *lde_cols, unsigned log_n, unsigned log_blowup, uint64_t offset,
                         uint64_t *out) {
    const size_t N = (size_t)1 << (log_n + log_blowup);
    const uint64_t wN = or_gl_root_of_unity(log_n + log_blowup);
#pragma omp parallel if (N >= 256)
    {
        gl3_t *slots = (gl3_t *)calloc(n_slots ? n_slots : 1, sizeof(gl3_t));
#pragma omp for schedule(static)
        for (size_t i = 0; i < N; ++i) {
            gl3_t acc[4];
            memset(acc, 0, sizeof acc);
            const uint64_t x = gl_mul(offset, gl_pow(wN, i));
            for (uint32_t pc = 0; pc < n_instr; ++pc) {
                const uint32_t w0 = code[2 * pc], w1 = code[2 * pc + 1];
                const unsigned op = w0 & 0xff, d = (w0 >> 8) & 0xf, kind = (w0 >> 12) & 0xf;
                gl3_t src = {{0, 0, 0}};
                if (op <= 4) {
                    switch (kind) {
                    case 0: src = acc[w1 & 3]; break;
                    case 1: src = slots[w1]; break;
                    case 2: memcpy(&src, consts3 + 3 * (size_t)w1, 24); break;
                    case 3: src.c[0] = lde_cols[w1 >> 24][(i + ((size_t)(w1 & 0xffffff) << log_blowup)) & (N - 1)]; break;
                    case 4: src.c[0] = tables[table_desc[2 * w1] + (i & (((size_t)1 << table_desc[2 * w1 + 1]) - 1))]; break;
                    default: src.c[0] = x; break;
                    }
                }
                switch (op) {
                case 0: acc[d] = src; break;
                case 1: acc[d] = gl3_add(acc[d], src); break;
                case 2: acc[d] = gl3_sub(acc[d], src); break;
                case 3: acc[d] = gl3_sub(src, acc[d]); break;
                case 4: acc[d] = gl3_mul(acc[d], src); break;
                case 5: { const gl3_t z = {{0, 0, 0}}; acc[d] = memcmp(&acc[d], &z, 24) ? gl3_inv(acc[d]) : z; } break;
                case 6: slots[w1] = acc[d]; break;
                case 7: memcpy(out + 3 * i, &acc[d], 24); break;
                default: break;
                }
            }
        }
        free(slots);
    }
}
