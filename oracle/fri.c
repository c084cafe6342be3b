/* ORACLE — TEST INFRASTRUCTURE ONLY (see fp252.h).
 *
 * Rows F1 and D1 of SURVEY.md §8(a).  Both live in the un-vendored ministark
 * (FriProver::build_layers, DeepPolyComposer; git 875fb385) and are called with
 * the reference's defaults fri_folding_factor = 8, fri_max_remainder_coeffs =
 * 16 (cli/src/main.rs:57-60) and the DEEP coefficient rule of
 * src/lib.rs:102-116 (powers of one alpha, degree adjustment (1,0) = none).
 * The FRI fold (M7/M8) is PINNED by data: the queried rows of consecutive layers in the proof files the
 * reference ships satisfy or_fri_fold_ex under flags = 3 (current code path) resp. 0 (older path) - see
 * tests/golden/make_fri_golden.py and fri_saved_proofs.json.  DEEP term order (M6) stays unpinned.
 * Written as naive O(fold^2) sums on purpose so that the HIP butterflies are checked against something
 * structurally different.
 */
#include "oracle.h"
#include <stdlib.h>

/* flags: OR_FRI_BITREV_ROWS (1): the vector is in bit-reversed order - row r = evals[fold r .. fold r + fold),
 * entry j of the row at x_r w_fold^bitrev(j), x_r = offset w^bitrev(r), output in bit-reversed order too;
 * OR_FRI_UNNORMALISED (2): out = fold * interpolant(alpha).  The two conventions found in the proof files the
 * reference ships (tests/golden/make_fri_golden.py): flags = 3 for the current code path, 0 for the older one. */
static unsigned bitrev_u(unsigned x, unsigned bits) {
    unsigned r = 0;
    for (unsigned i = 0; i < bits; ++i) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

void or_fri_fold_ex(const fp_t *evals, unsigned log_len, unsigned fold, fp_t alpha, fp_t offset,
                    unsigned flags, fp_t *out) {
    size_t len = (size_t)1 << log_len, rows = len / fold;
    unsigned log_fold = 0;
    while ((1u << log_fold) < fold) ++log_fold;
    const unsigned row_bits = log_len - log_fold;
    fp_t w = fp_root_of_unity(log_len);
    fp_t wf_inv = fp_inv(fp_root_of_unity(log_fold));
    fp_t fold_inv = (flags & 2u) ? FP_ONE : fp_inv(fp_from_u64(fold));
    /* table of wf_inv^e, e < fold */
    fp_t wfp[64];
    wfp[0] = FP_ONE;
    for (unsigned e = 1; e < fold; ++e) wfp[e] = fp_mul(wfp[e - 1], wf_inv);
#pragma omp parallel for schedule(static) if (rows >= 256)
    for (size_t j = 0; j < rows; ++j) {
        const uint64_t e = (flags & 1u) ? bitrev_u((unsigned)j, row_bits) : (uint64_t)j;
        fp_t xj = fp_mul(offset, fp_pow_u64(w, e));
        fp_t t = fp_mul(alpha, fp_inv(xj)); /* alpha / x_j */
        fp_t acc = {{0, 0, 0, 0}}, tm = FP_ONE;
        for (unsigned m = 0; m < fold; ++m) {
            /* c_m * x_j^m = (1/fold) sum_k v_k wf^(-k m),  v_k = f(x_j wf^k) */
            fp_t s = {{0, 0, 0, 0}};
            for (unsigned k = 0; k < fold; ++k) {
                const fp_t vk = (flags & 1u) ? evals[j * fold + bitrev_u(k, log_fold)] : evals[j + k * rows];
                s = fp_add(s, fp_mul(vk, wfp[(k * m) % fold]));
            }
            acc = fp_add(acc, fp_mul(fp_mul(s, fold_inv), tm));
            tm = fp_mul(tm, t);
        }
        out[j] = acc;
    }
}

void or_fri_fold(const fp_t *evals, unsigned log_len, unsigned fold, fp_t alpha, fp_t offset,
                 fp_t *out) {
    or_fri_fold_ex(evals, log_len, fold, alpha, offset, 0, out);
}

void or_deep_compose(const fp_t *const *trace_lde, const fp_t *const *comp_lde, unsigned log_n,
                     unsigned log_blowup, fp_t offset, const uint32_t *mask_col,
                     const uint32_t *mask_off, size_t nmask, const fp_t *ood_trace,
                     const fp_t *coeff_trace, size_t ncomp, const fp_t *ood_comp,
                     const fp_t *coeff_comp, fp_t z, fp_t *out) {
    unsigned log_N = log_n + log_blowup;
    size_t N = (size_t)1 << log_N;
    fp_t wn = fp_root_of_unity(log_n), wN = fp_root_of_unity(log_N);
    fp_t *zs = (fp_t *)malloc(sizeof(fp_t) * (nmask ? nmask : 1));
    for (size_t j = 0; j < nmask; ++j) zs[j] = fp_mul(z, fp_pow_u64(wn, mask_off[j]));
    fp_t zc = fp_pow_u64(z, (uint64_t)ncomp);
#pragma omp parallel for schedule(static) if (N >= 256)
    for (size_t i = 0; i < N; ++i) {
        fp_t x = fp_mul(offset, fp_pow_u64(wN, (uint64_t)i));
        fp_t acc = {{0, 0, 0, 0}};
        for (size_t j = 0; j < nmask; ++j) {
            fp_t num = fp_sub(trace_lde[mask_col[j]][i], ood_trace[j]);
            fp_t den = fp_inv(fp_sub(x, zs[j]));
            acc = fp_add(acc, fp_mul(coeff_trace[j], fp_mul(num, den)));
        }
        fp_t denc = fp_inv(fp_sub(x, zc));
        for (size_t k = 0; k < ncomp; ++k) {
            fp_t num = fp_sub(comp_lde[k][i], ood_comp[k]);
            acc = fp_add(acc, fp_mul(coeff_comp[k], fp_mul(num, denc)));
        }
        out[i] = acc;
    }
    free(zs);
}

/* The same DEEP composition with ONE field inversion per point (Montgomery's trick over the point's denominators)
 * instead of one per (point, cell): the definition above is what the tests hold the GPU kernel to at small sizes; this
 * is what the whole-pipeline CPU runs use (bench.py's cpu_baseline, the multi-rank gloo tests), checked equal to the
 * definition in tests/test_oracle_defs.py.  Row-block form: out[j], j < nrows, is the value at LDE row row0 + j * stride,
 * read from trace_lde[c][j * stride] (columns that start at row row0; stride = blowup gives the sub-coset offset * <w_n>). */
void or_deep_compose_rows(const fp_t *const *trace_lde, const fp_t *const *comp_lde, unsigned log_n,
                          unsigned log_blowup, fp_t offset, const uint32_t *mask_col,
                          const uint32_t *mask_off, size_t nmask, const fp_t *ood_trace,
                          const fp_t *coeff_trace, size_t ncomp, const fp_t *ood_comp,
                          const fp_t *coeff_comp, fp_t z, uint64_t row0, uint64_t nrows, uint64_t stride, fp_t *out) {
    unsigned log_N = log_n + log_blowup;
    fp_t wn = fp_root_of_unity(log_n), wN = fp_root_of_unity(log_N);
    /* distinct denominators: one per distinct row offset, plus the composition point */
    size_t ndist = 0;
    uint32_t *dist_off = (uint32_t *)malloc(sizeof(uint32_t) * (nmask + 1));
    size_t *which = (size_t *)malloc(sizeof(size_t) * (nmask + 1));
    for (size_t j = 0; j < nmask; ++j) {
        size_t k = 0;
        while (k < ndist && dist_off[k] != mask_off[j]) ++k;
        if (k == ndist) dist_off[ndist++] = mask_off[j];
        which[j] = k;
    }
    fp_t *zs = (fp_t *)malloc(sizeof(fp_t) * (ndist + 1));
    for (size_t k = 0; k < ndist; ++k) zs[k] = fp_mul(z, fp_pow_u64(wn, dist_off[k]));
    zs[ndist] = fp_pow_u64(z, (uint64_t)ncomp);
    const size_t nd = ndist + 1;
#pragma omp parallel
    {
        fp_t *den = (fp_t *)malloc(sizeof(fp_t) * nd), *pre = (fp_t *)malloc(sizeof(fp_t) * nd);
#pragma omp for schedule(static)
        for (uint64_t r = 0; r < nrows; ++r) {
            const uint64_t k = r * stride;
            fp_t x = fp_mul(offset, fp_pow_u64(wN, row0 + k));
            fp_t run = FP_ONE;
            for (size_t q = 0; q < nd; ++q) { den[q] = fp_sub(x, zs[q]); pre[q] = run; run = fp_mul(run, den[q]); }
            fp_t inv = fp_inv(run);
            for (size_t q = nd; q-- > 0;) { fp_t d = den[q]; den[q] = fp_mul(inv, pre[q]); inv = fp_mul(inv, d); }
            fp_t acc = {{0, 0, 0, 0}};
            for (size_t j = 0; j < nmask; ++j) {
                fp_t num = fp_sub(trace_lde[mask_col[j]][k], ood_trace[j]);
                acc = fp_add(acc, fp_mul(coeff_trace[j], fp_mul(num, den[which[j]])));
            }
            for (size_t kk = 0; kk < ncomp; ++kk) {
                fp_t num = fp_sub(comp_lde[kk][k], ood_comp[kk]);
                acc = fp_add(acc, fp_mul(coeff_comp[kk], fp_mul(num, den[ndist])));
            }
            out[r] = acc;
        }
        free(den); free(pre);
    }
    free(zs); free(which); free(dist_off);
}

/* out[j] = T_{mask_col[j]}(z * w_n^{mask_off[j]}) from natural-order coefficient columns (Horner per cell) */
void or_ood_eval(const fp_t *const *coeffs, unsigned log_n, const uint32_t *mask_col, const uint32_t *mask_off,
                 size_t nmask, fp_t z, fp_t *out) {
    const size_t n = (size_t)1 << log_n;
    fp_t wn = fp_root_of_unity(log_n);
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t j = 0; j < nmask; ++j)
        out[j] = or_poly_eval(coeffs[mask_col[j]], n, fp_mul(z, fp_pow_u64(wn, mask_off[j])));
}

/* out[i] = 1 / (offset * w_N^i - c), i < 2^log_N (the full-length zerofier tables); batch inversion per chunk */
void or_inverse_table(unsigned log_N, fp_t offset, fp_t c, fp_t *out) {
    const size_t N = (size_t)1 << log_N, chunk = 1024;
    fp_t w = fp_root_of_unity(log_N);
#pragma omp parallel for schedule(static)
    for (size_t b = 0; b < (N + chunk - 1) / chunk; ++b) {
        const size_t lo = b * chunk, hi = lo + chunk < N ? lo + chunk : N;
        fp_t pre[1024], x = fp_mul(offset, fp_pow_u64(w, (uint64_t)lo)), run = FP_ONE;
        for (size_t i = lo; i < hi; ++i) { out[i] = fp_sub(x, c); pre[i - lo] = run; run = fp_mul(run, out[i]); x = fp_mul(x, w); }
        fp_t inv = fp_inv(run);
        for (size_t i = hi; i-- > lo;) { fp_t d = out[i]; out[i] = fp_mul(inv, pre[i - lo]); inv = fp_mul(inv, d); }
    }
}
