/* ORACLE — TEST INFRASTRUCTURE ONLY (see fp252.h).
 *
 * Extension-trace columns, row A2 of SURVEY.md §8(a): Trace::build_extension_columns
 * (layouts/src/recursive/trace.rs:699-814, layouts/src/starknet/trace.rs:997-1100), restated as the
 * reference writes it: sequential running products, ark-ff batch_inversion of the denominator
 * products, n * d_inv scattered at the permutation's row offsets; and the diluted-check aggregate
 * recurrence.  The algorithm is entirely in the vendored source (no ministark convention involved);
 * no reference test pins it, so the pin is this line-by-line restatement plus the big-integer
 * definition in tests/test_oracle_defs.py.
 */
#include "oracle.h"
#include <stdlib.h>

/* ark-ff 0.4.2 batch_inversion (fields/mod.rs, serial_batch_inversion_and_mul): every non-zero element is replaced by
 * its inverse, zeros are left untouched - Montgomery's trick as ark-ff writes it: prefix products of the non-zero
 * elements, one inversion, a backward pass. */
static void batch_inversion(fp_t *v, uint64_t n) {
    fp_t *prod = (fp_t *)malloc((n ? n : 1) * sizeof(fp_t));
    fp_t tmp = FP_ONE;
    uint64_t m = 0;
    for (uint64_t i = 0; i < n; ++i)
        if (!fp_is_zero(v[i])) { tmp = fp_mul(tmp, v[i]); prod[m++] = tmp; }
    tmp = fp_inv(tmp);
    for (uint64_t i = n; i-- > 0;) {
        if (fp_is_zero(v[i])) continue;
        --m;
        const fp_t new_tmp = fp_mul(tmp, v[i]);
        v[i] = m ? fp_mul(tmp, prod[m - 1]) : tmp;
        tmp = new_tmp;
    }
    free(prod);
}

static fp_t perm_term(const fp_t *col, uint64_t stride, uint64_t a_off, int64_t v_off, uint64_t k, fp_t z, fp_t alpha) {
    const fp_t *item = col + k * stride;
    if (v_off < 0) return fp_sub(z, item[a_off]);                                /* z - chunk[..]        trace.rs:727-728 */
    return fp_sub(z, fp_add(fp_mul(alpha, item[v_off]), item[a_off]));           /* z - (alpha * v + a)   trace.rs:713-714 */
}

void or_permutation_product(const fp_t *num, uint64_t num_stride, uint64_t num_a, int64_t num_v,
                            const fp_t *den, uint64_t den_stride, uint64_t den_a, int64_t den_v,
                            uint64_t count, fp_t z, fp_t alpha, fp_t *out, uint64_t out_stride, uint64_t out_off) {
    fp_init();
    fp_t *numerators = (fp_t *)malloc(sizeof(fp_t) * count), *denominators = (fp_t *)malloc(sizeof(fp_t) * count);
    fp_t numerator_acc = FP_ONE, denominator_acc = FP_ONE;
    for (uint64_t k = 0; k < count; ++k) {                                       /* trace.rs:712-719 */
        numerator_acc = fp_mul(numerator_acc, perm_term(num, num_stride, num_a, num_v, k, z, alpha));
        denominator_acc = fp_mul(denominator_acc, perm_term(den, den_stride, den_a, den_v, k, z, alpha));
        numerators[k] = numerator_acc;
        denominators[k] = denominator_acc;
    }
    batch_inversion(denominators, count);                                        /* trace.rs:720 */
    for (uint64_t i = 0; i < count; ++i)                                         /* trace.rs:766-769 */
        out[i * out_stride + out_off] = fp_mul(numerators[i], denominators[i]);
    free(numerators); free(denominators);
}

void or_diluted_aggregate(const fp_t *ordered, uint64_t stride, uint64_t off, uint64_t count, fp_t z, fp_t alpha,
                          fp_t *out, uint64_t out_stride, uint64_t out_off) {
    fp_init();
    if (!count) return;
    fp_t acc = FP_ONE;                                                           /* trace.rs:792-793 */
    out[out_off] = acc;
    for (uint64_t i = 1; i < count; ++i) {                                       /* trace.rs:797-803 */
        const fp_t u = fp_sub(ordered[i * stride + off], ordered[(i - 1) * stride + off]);
        acc = fp_add(fp_mul(acc, fp_add(FP_ONE, fp_mul(z, u))), fp_mul(alpha, fp_mul(u, u)));
        out[i * out_stride + out_off] = acc;
    }
}
