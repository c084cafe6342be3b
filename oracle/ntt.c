/* ORACLE — TEST INFRASTRUCTURE ONLY (see fp252.h).
 *
 * Restates rows N1/N2 of SURVEY.md §8(a): ministark's Matrix::interpolate /
 * Matrix::evaluate (un-vendored, git 875fb385; call sites src/lib.rs:17-26)
 * which delegate to ark-poly 0.4.2 Radix2EvaluationDomain fft/ifft.  The
 * convention is pinned by the reference's own periodic-column tests
 * (builtins/src/pedersen/periodic.rs:1183-1209): fft(coeffs)[k] = P(w^k) in
 * natural order with w = 3^((p-1)/n).
 */
#include "oracle.h"
#include <stdlib.h>

static unsigned bitrev(unsigned x, unsigned bits) {
    unsigned r = 0;
    for (unsigned i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

void or_bitrev_permute(fp_t *a, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
#pragma omp parallel for schedule(static) if (n >= 4096)
    for (size_t i = 0; i < n; ++i) {
        size_t j = bitrev((unsigned)i, log_n);
        if (i < j) { fp_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
}

/* out[k] = first * w^k, k < count: chunks by all threads (a chunk's first power by square-and-multiply, the rest by one product each) -
 * the twiddle table of a 2^25-point transform is 2^24 powers: one after the other they were a fifth of the transform's time */
static void powers_into(fp_t *out, size_t count, fp_t first, fp_t w) {
    const size_t chunk = 1 << 12;
#pragma omp parallel for schedule(static) if (count >= 4 * chunk)
    for (size_t c0 = 0; c0 < count; c0 += chunk) {
        fp_t g = fp_mul(first, fp_pow_u64(w, (uint64_t)c0));
        const size_t end = c0 + chunk < count ? c0 + chunk : count;
        for (size_t k = c0; k < end; ++k) { out[k] = g; g = fp_mul(g, w); }
    }
}
/* a[i] *= first * w^i */
static void scale_by_powers(fp_t *a, size_t n, fp_t first, fp_t w) {
    const size_t chunk = 1 << 12;
#pragma omp parallel for schedule(static) if (n >= 4 * chunk)
    for (size_t c0 = 0; c0 < n; c0 += chunk) {
        fp_t g = fp_mul(first, fp_pow_u64(w, (uint64_t)c0));
        const size_t end = c0 + chunk < n ? c0 + chunk : n;
        for (size_t i = c0; i < end; ++i) { a[i] = fp_mul(a[i], g); g = fp_mul(g, w); }
    }
}

/* in-place radix-2 DIT, natural in / natural out, root w (Montgomery) */
static void ntt_core(fp_t *a, unsigned log_n, fp_t w) {
    size_t n = (size_t)1 << log_n;
    if (log_n == 0) return;
    or_bitrev_permute(a, log_n);
    fp_t *tw = (fp_t *)malloc(sizeof(fp_t) * (n / 2 ? n / 2 : 1));
    powers_into(tw, n / 2 ? n / 2 : 1, FP_ONE, w);
    for (unsigned s = 0; s < log_n; ++s) {
        size_t half = (size_t)1 << s, step = n >> (s + 1);
#pragma omp parallel for schedule(static) if (n >= 4096)
        for (size_t b = 0; b < n / 2; ++b) {
            size_t blk = b >> s, j = b & (half - 1);
            size_t i0 = (blk << (s + 1)) + j, i1 = i0 + half;
            fp_t t = fp_mul(a[i1], tw[j * step]);
            fp_t u = a[i0];
            a[i0] = fp_add(u, t);
            a[i1] = fp_sub(u, t);
        }
    }
    free(tw);
}

/* evaluate: coefficients -> evaluations on offset*<w_n>, natural order */
void or_ntt_forward(fp_t *a, unsigned log_n, const fp_t *offset) {
    size_t n = (size_t)1 << log_n;
    if (offset && !fp_eq(*offset, FP_ONE)) {
        scale_by_powers(a, n, FP_ONE, *offset);
    }
    ntt_core(a, log_n, fp_root_of_unity(log_n));
}

/* interpolate: evaluations on offset*<w_n> (natural order) -> coefficients */
void or_ntt_inverse(fp_t *a, unsigned log_n, const fp_t *offset) {
    size_t n = (size_t)1 << log_n;
    fp_t winv = fp_inv(fp_root_of_unity(log_n));
    ntt_core(a, log_n, winv);
    fp_t ninv = fp_inv(fp_from_u64((uint64_t)n));
    if (offset && !fp_eq(*offset, FP_ONE)) {
        scale_by_powers(a, n, ninv, fp_inv(*offset));
    } else {
#pragma omp parallel for schedule(static) if (n >= 4096)
        for (size_t i = 0; i < n; ++i) a[i] = fp_mul(a[i], ninv);
    }
}

/* One column of the low-degree extension (pipeline steps 3+4):
 * interpolate over <w_n>, then evaluate over offset*<w_{n*2^log_blowup}>.
 * coeffs_out (n) and evals_out (n << log_blowup) may not alias `in`. */
void or_lde(const fp_t *in, unsigned log_n, unsigned log_blowup, const fp_t *offset,
            fp_t *evals_out, fp_t *coeffs_out) {
    size_t n = (size_t)1 << log_n, N = n << log_blowup;
    fp_t *c = coeffs_out ? coeffs_out : (fp_t *)malloc(sizeof(fp_t) * n);
    memcpy(c, in, sizeof(fp_t) * n);
    or_ntt_inverse(c, log_n, NULL);
    memcpy(evals_out, c, sizeof(fp_t) * n);
    memset(evals_out + n, 0, sizeof(fp_t) * (N - n));
    or_ntt_forward(evals_out, log_n + log_blowup, offset);
    if (!coeffs_out) free(c);
}

/* Horner evaluation of a coefficient vector at x (all Montgomery). */
fp_t or_poly_eval(const fp_t *coeffs, size_t n, fp_t x) {
    fp_t r = {{0, 0, 0, 0}};
    for (size_t i = n; i-- > 0;) r = fp_add(fp_mul(r, x), coeffs[i]);
    return r;
}
