// Host check of the extension-trace scans (sandstorm_amd/csrc/ext_scan.h): the lane bodies and the level driver
// the HIP kernels run, executed here with a loop in place of each launch and compared with the reference's
// sequential loops (layouts/src/recursive/trace.rs:699-814) written with the same field arithmetic.
//   ext_scan_test             -> prints "ok <cases>" or the first mismatch
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../sandstorm_amd/csrc/ext_scan.h"

using namespace ss;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t splitmix() {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static Fp random_fp() {
    Fp r;
    for (int i = 0; i < 8; ++i) r.v[i] = (uint32_t)splitmix();
    r.v[7] &= 0x03ffffffu;                      // < 2^250 < p
    return r;
}

struct LoopExec {                               // a "launch" = the lane body for every lane, in any order (here: descending)
    ScanShape shape;
    template <class Op> int reduce(uint64_t lanes, const Fp *data, uint64_t n, Fp *agg, uint32_t lc) {
        for (uint64_t c = lanes + 3; c-- > 0;) scan_reduce_lane<Op>(c, data, n, agg, lc);
        return 0;
    }
    template <class Op> int apply(uint64_t lanes, Fp *data, uint64_t n, const Fp *aggscan, uint32_t lc) {
        for (uint64_t c = lanes + 3; c-- > 0;) scan_apply_lane<Op>(c, data, n, aggscan, lc);
        return 0;
    }
    template <class Op> int apply_out(uint64_t lanes, const Fp *data, uint64_t n, const Fp *aggscan, uint32_t lc, Fp *out, uint64_t os, uint64_t oo) {
        for (uint64_t c = lanes + 3; c-- > 0;) scan_apply_out_lane<Op>(c, data, n, aggscan, lc, out, os, oo);
        return 0;
    }
    int perm_quotients(const PermOperand &num, const PermOperand &den, uint64_t count, const Fp &z, const Fp &alpha, Fp *q, Fp *tmp, uint32_t lc) {
        for (uint64_t c = scan_chunks(count, lc) + 3; c-- > 0;) perm_quotients_lane(c, num, den, count, z, alpha, q, tmp, lc);
        return 0;
    }
    int dil_terms(const Fp *x, uint64_t stride, uint64_t off, uint64_t count, const Fp &z, const Fp &alpha, Fp *mc) {
        for (uint64_t k = count + 3; k-- > 0;) dil_terms_lane(k, x, stride, off, count, z, alpha, mc);
        return 0;
    }
    int dil_terms_block(const Fp *x, uint64_t stride, uint64_t off, uint64_t count, bool starts_column, const Fp &z, const Fp &alpha, Fp *mc) {
        for (uint64_t k = count + 3; k-- > 0;) dil_terms_block_lane(k, x, stride, off, count, starts_column, z, alpha, mc);
        return 0;
    }
};

static Fp term(const Fp *col, uint64_t stride, uint64_t a, int64_t v, uint64_t k, const Fp &z, const Fp &alpha) {
    const Fp *item = col + k * stride;
    if (v < 0) return fp_sub(z, item[a]);
    return fp_sub(z, fp_add(fp_mul(alpha, item[v]), item[a]));
}

int main() {
    const uint64_t counts[] = {1, 2, 3, 4, 5, 15, 16, 17, 63, 64, 65, 255, 256, 257, 4095, 4096, 4097, 20000};
    int cases = 0;
    const ScanShape shapes[] = {ScanShape{}, ScanShape{2, 1}, ScanShape{6, 6}, ScanShape{3, 7}};
    for (const ScanShape &shape : shapes)
    for (uint64_t count : counts) {
        LoopExec ex{shape};
        if (&shape != &shapes[0] && count > 5000) continue;
        for (int variant = 0; variant < 3; ++variant) {
            // variant 0: (address, value) terms, stride 2, out stride 2 offset 0
            // variant 1: single-value terms out of one column at offsets 0 / 2 of stride 4, out stride 4 offset 1
            // variant 2: like 1 with a zero denominator term in the middle (batch_inversion leaves zeros)
            const uint64_t stride = variant == 0 ? 2 : 4, os = variant == 0 ? 2 : 4, oo = variant == 0 ? 0 : 1;
            std::vector<Fp> a(count * stride), b(count * stride);
            for (auto &x : a) x = random_fp();
            for (auto &x : b) x = random_fp();
            const Fp z = random_fp(), alpha = random_fp();
            if (variant == 2) a[(count / 2) * stride + 2] = z;           // z - x = 0 at item count/2
            PermOperand num, den;
            if (variant == 0) { num = PermOperand{a.data(), 2, 0, 1}; den = PermOperand{b.data(), 2, 0, 1}; }
            else { num = PermOperand{a.data(), 4, 0, -1}; den = PermOperand{a.data(), 4, 2, -1}; }
            const Fp sentinel = random_fp();
            std::vector<Fp> out(count * os, sentinel), want(count * os, sentinel);
            std::vector<Fp> scratch(permutation_product_scratch_felts(count));
            if (permutation_product(ex, num, den, count, z, alpha, out.data(), os, oo, scratch.data())) { printf("exec error\n"); return 1; }
            Fp nacc = fp_one(), dacc = fp_one();
            for (uint64_t k = 0; k < count; ++k) {
                nacc = fp_mul(nacc, term(num.data, num.stride, num.a_off, num.v_off, k, z, alpha));
                dacc = fp_mul(dacc, term(den.data, den.stride, den.a_off, den.v_off, k, z, alpha));
                const Fp dinv = fp_is_zero(dacc) ? dacc : fp_inv(dacc);
                want[k * os + oo] = fp_mul(nacc, dinv);
            }
            for (uint64_t i = 0; i < out.size(); ++i)
                if (!fp_eq(out[i], want[i])) { printf("permutation product mismatch: count %llu variant %d index %llu\n", (unsigned long long)count, variant, (unsigned long long)i); return 1; }
            ++cases;
        }
        // diluted aggregate: stride 8 offset 5 -> out stride 8 offset 3 (the starknet shape), and the dense recursive shape
        for (int dense = 0; dense < 2; ++dense) {
            const uint64_t stride = dense ? 1 : 8, off = dense ? 0 : 5, os = dense ? 1 : 8, oo = dense ? 0 : 3;
            std::vector<Fp> x(count * stride);
            for (auto &v : x) v = random_fp();
            const Fp z = random_fp(), alpha = random_fp(), sentinel = random_fp();
            std::vector<Fp> out(count * os, sentinel), want(count * os, sentinel);
            std::vector<Fp> scratch(diluted_aggregate_scratch_felts(count));
            if (diluted_aggregate(ex, x.data(), stride, off, count, z, alpha, out.data(), os, oo, scratch.data())) { printf("exec error\n"); return 1; }
            Fp acc = fp_one();
            want[oo] = acc;
            for (uint64_t i = 1; i < count; ++i) {
                const Fp u = fp_sub(x[i * stride + off], x[(i - 1) * stride + off]);
                acc = fp_add(fp_mul(acc, fp_add(fp_one(), fp_mul(z, u))), fp_mul(alpha, fp_sqr(u)));
                want[i * os + oo] = acc;
            }
            for (uint64_t i = 0; i < out.size(); ++i)
                if (!fp_eq(out[i], want[i])) { printf("diluted aggregate mismatch: count %llu dense %d index %llu\n", (unsigned long long)count, dense, (unsigned long long)i); return 1; }
            ++cases;
        }
        // the same column as row blocks (one scan over several devices): every block's own maps, the term between two blocks
        // composed from the two boundary values, the value before the block applied - the cells of the single scan (`want`)
        for (uint64_t nblocks : {2ull, 5ull}) {
            if (count < nblocks || count > 5000) continue;
            const uint64_t stride = 8, off = 5, os = 8, oo = 3, per = count / nblocks, total = per * nblocks;
            std::vector<Fp> x(total * stride);
            for (auto &v : x) v = random_fp();
            const Fp z = random_fp(), alpha = random_fp(), sentinel = random_fp();
            std::vector<Fp> out(total * os, sentinel), want(total * os, sentinel), scratch(diluted_aggregate_scratch_felts(total));
            if (diluted_aggregate(ex, x.data(), stride, off, total, z, alpha, want.data(), os, oo, scratch.data())) { printf("exec error\n"); return 1; }
            Fp value = fp_zero();
            for (uint64_t b = 0; b < nblocks; ++b) {
                const Fp *xb = x.data() + b * per * stride;
                std::vector<Fp> maps(2 * per), aggs(scan_agg_felts(per, 2));
                if (diluted_aggregate_maps(ex, xb, stride, off, per, b == 0, z, alpha, maps.data(), aggs.data())) { printf("exec error\n"); return 1; }
                Fp start = value;
                if (b) {
                    const Fp u = fp_sub(xb[off], xb[off - stride]);
                    start = fp_add(fp_mul(value, fp_add(fp_one(), fp_mul(z, u))), fp_mul(alpha, fp_sqr(u)));
                }
                for (uint64_t k = per + 3; k-- > 0;) affine_apply_lane(k, maps.data(), per, start, out.data() + b * per * os, os, oo);
                value = fp_add(fp_mul(maps[2 * (per - 1)], start), maps[2 * (per - 1) + 1]);
            }
            for (uint64_t i = 0; i < out.size(); ++i)
                if (!fp_eq(out[i], want[i])) { printf("blocked aggregate mismatch: count %llu blocks %llu index %llu\n", (unsigned long long)count, (unsigned long long)nblocks, (unsigned long long)i); return 1; }
            // a running product over the same blocks: the block's own scan, scaled by the product of the blocks before it
            std::vector<Fp> a(total * 4);
            for (auto &v : a) v = random_fp();
            std::vector<Fp> pout(total * 4, sentinel), pwant(total * 4, sentinel), ps(permutation_product_scratch_felts(total));
            const PermOperand num{a.data(), 4, 0, -1}, den{a.data(), 4, 2, -1};
            if (permutation_product(ex, num, den, total, z, alpha, pwant.data(), 4, 1, ps.data())) { printf("exec error\n"); return 1; }
            Fp before = fp_one();
            for (uint64_t b = 0; b < nblocks; ++b) {
                const PermOperand nb{a.data() + 4 * b * per, 4, 0, -1}, db{a.data() + 4 * b * per, 4, 2, -1};
                Fp *ob = pout.data() + 4 * b * per;
                if (permutation_product(ex, nb, db, per, z, alpha, ob, 4, 1, ps.data())) { printf("exec error\n"); return 1; }
                const Fp last = ob[4 * (per - 1) + 1];
                if (b) for (uint64_t k = per + 3; k-- > 0;) scale_strided_lane(k, ob, 4, 1, per, before);
                before = fp_mul(before, last);
            }
            for (uint64_t i = 0; i < pout.size(); ++i)
                if (!fp_eq(pout[i], pwant[i])) { printf("blocked product mismatch: count %llu blocks %llu index %llu\n", (unsigned long long)count, (unsigned long long)nblocks, (unsigned long long)i); return 1; }
            ++cases;
        }
    }
    printf("ok %d\n", cases);
    return 0;
}
