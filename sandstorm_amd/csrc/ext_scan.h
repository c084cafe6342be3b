// ext_scan.h — lane bodies and the level driver of the extension-trace scans (ext.hip), host+device so that
// tests/cpp/ext_scan_test.cpp can run the very same code on the CPU with a loop in place of a launch.
#pragma once
#include "fp252.h"
#include "fl252.h"
#include "inv252.h"

namespace ss {

// items per lane: the scans and the inversion are chains of dependent multiplications fed by independent loads, so
// short chunks (more lanes in flight) hide the load latency; the inversion pays one safegcd per chunk and wants
// longer ones.  Runtime values so that ext.hip can take them from the environment when tuning.
struct ScanShape { uint32_t log_scan = 3, log_inv = 5; };   // measured: tools/ext_bench.py, DESIGN.md
constexpr uint32_t SCAN_MIN_LOG_CHUNK = 2;

struct PermOperand {            // ss_perm_operand with device pointers typed
    const Fp *data;
    uint64_t stride, a_off;
    int64_t v_off;              // < 0: single-value term z - a_k
};

// ---- the two monoids --------------------------------------------------------------------
// A lane's running value stays in the hot-loop form (fl252.h: 9 x 28-bit limbs, 173 G products/s where the 8 x 32 interchange form
// makes 93): items are re-limbed as they are loaded, what is stored is the canonical image again - the same bits either way.
struct MulOp {                       // field multiplication
    typedef Fl T;
    static constexpr uint32_t FELTS = 1;
    static SS_HD T identity() { return fl_one(); }
    static SS_HD T load(const Fp *base, uint64_t i) { return fl_from_fp(base[i]); }
    static SS_HD void store(Fp *base, uint64_t i, const T &x) { base[i] = fl_to_fp(x); }
    static SS_HD T combine(const T &acc, const T &x) { return fn_mul(acc, x); }           // x: a loaded item (normalised limbs)
    static SS_HD Fp value(const T &x) { return fl_to_fp(x); }
};
struct Affine { Fp m, c; };          // t -> m t + c
struct AffineL { Fl m, c; };         // the same in limbs; c may be lazy (a product < 1.25 p plus an item < p)
struct AffineOp {                    // "then": (acc then x)(t) = x.m (acc.m t + acc.c) + x.c
    typedef AffineL T;
    static constexpr uint32_t FELTS = 2;
    static SS_HD T identity() { return AffineL{fl_one(), fl_zero()}; }
    static SS_HD T load(const Fp *base, uint64_t i) { return AffineL{fl_from_fp(base[2 * i]), fl_from_fp(base[2 * i + 1])}; }
    static SS_HD void store(Fp *base, uint64_t i, const T &x) { base[2 * i] = fl_to_fp(x.m); base[2 * i + 1] = fl_to_fp(x.c); }
    static SS_HD void store(Fp *base, uint64_t i, const Affine &x) { base[2 * i] = x.m; base[2 * i + 1] = x.c; }
    static SS_HD T combine(const T &acc, const T &x) { return AffineL{fn_mul(acc.m, x.m), fl_add(fn_mul(acc.c, x.m), x.c)}; }
    static SS_HD Fp value(const T &x) { return fl_to_fp(x.c); }          // of a constant map (the aggregate's item 0 is one, so every prefix is)
};

// ---- lane bodies (c = chunk index = global lane id) -------------------------------------------
template <class Op>
SS_HD void scan_reduce_lane(uint64_t c, const Fp *data, uint64_t n, Fp *agg, uint32_t lc) {
    const uint64_t i0 = c << lc;
    if (i0 >= n) return;
    const uint64_t i1 = i0 + (1ull << lc) < n ? i0 + (1ull << lc) : n;
    typename Op::T acc = Op::load(data, i0);
    for (uint64_t i = i0 + 1; i < i1; ++i) acc = Op::combine(acc, Op::load(data, i));
    Op::store(agg, c, acc);
}

// in place: data[i] <- data[0] . data[1] ... data[i]; aggscan = inclusive scan of the chunk aggregates (null: one chunk)
template <class Op>
SS_HD void scan_apply_lane(uint64_t c, Fp *data, uint64_t n, const Fp *aggscan, uint32_t lc) {
    const uint64_t i0 = c << lc;
    if (i0 >= n) return;
    const uint64_t i1 = i0 + (1ull << lc) < n ? i0 + (1ull << lc) : n;
    typename Op::T acc = (c && aggscan) ? Op::load(aggscan, c - 1) : Op::identity();
    for (uint64_t i = i0; i < i1; ++i) {
        acc = Op::combine(acc, Op::load(data, i));
        Op::store(data, i, acc);
    }
}

// the last level of a scan whose cells go to a strided column: out[i * out_stride + out_off] <- value(data[0] . ... . data[i]) (data is
// only read: one pass and one array less than scanning in place and copying)
template <class Op>
SS_HD void scan_apply_out_lane(uint64_t c, const Fp *data, uint64_t n, const Fp *aggscan, uint32_t lc, Fp *out, uint64_t out_stride, uint64_t out_off) {
    const uint64_t i0 = c << lc;
    if (i0 >= n) return;
    const uint64_t i1 = i0 + (1ull << lc) < n ? i0 + (1ull << lc) : n;
    typename Op::T acc = (c && aggscan) ? Op::load(aggscan, c - 1) : Op::identity();
    for (uint64_t i = i0; i < i1; ++i) {
        acc = Op::combine(acc, Op::load(data, i));
        out[i * out_stride + out_off] = Op::value(acc);
    }
}

SS_HD Fp perm_term(const PermOperand &o, uint64_t k, const Fp &z, const Fp &alpha) {
    const Fp *item = o.data + k * o.stride;
    const Fp a = item[o.a_off];
    if (o.v_off < 0) return fp_sub(z, a);                                    // z - x_k           (trace.rs:727-728)
    return fp_sub(z, fp_add(fp_mul(alpha, item[o.v_off]), a));               // z - (alpha v + a)  (trace.rs:713-714)
}
// q_k = n_k / d_k for the chunk's items (chunk c = items c, c + m, c + 2 m, ...: neighbouring lanes touch neighbouring items) in ONE
// pass: the denominators' prefix products forward (tmp: count felts), one inversion, and on the way back the terms again - they are
// a subtraction and at most one product of cells that are still in the cache - times the inverse.  A zero denominator gives q_k = 0
// and stays out of the chunk's product (batch_inversion's zeros).  perm_terms + inverse_dense + perm_finish in one launch and
// 11 instead of 17 element transfers per item.
SS_HD void perm_quotients_lane(uint64_t c, const PermOperand &num, const PermOperand &den, uint64_t count, const Fp &z, const Fp &alpha,
                               Fp *q, Fp *tmp, uint32_t lc) {
    const uint64_t m = (count + (1ull << lc) - 1) >> lc;
    if (c >= m) return;
    Fl run = fl_one();
    for (uint64_t k = c; k < count; k += m) {
        const Fl d = fl_from_fp(perm_term(den, k, z, alpha));
        tmp[k] = fl_pack(run);
        if (!fn_is_zero(d)) run = fn_mul(run, d);
    }
    Fl inv = fl_from_fp(fp_inv_safegcd(fl_to_fp(run)));
    const uint64_t last = c + ((count - 1 - c) / m) * m;
    for (uint64_t k = last;; k -= m) {
        const Fl d = fl_from_fp(perm_term(den, k, z, alpha));
        if (fn_is_zero(d)) q[k] = fp_zero();
        else {
            const Fl dinv = fn_mul(inv, fl_from_fp(tmp[k]));
            q[k] = fl_to_fp(fn_mul(fl_from_fp(perm_term(num, k, z, alpha)), dinv));
            inv = fn_mul(inv, d);
        }
        if (k == c) break;
    }
}
// item 0 is the constant map t -> 1 (the initial value), item k the map t -> t (1 + z u_k) + alpha u_k^2
SS_HD void dil_terms_lane(uint64_t k, const Fp *x, uint64_t stride, uint64_t off, uint64_t count, const Fp &z,
                          const Fp &alpha, Fp *mc) {
    if (k >= count) return;
    Affine f;
    if (k == 0) {
        f.m = fp_zero(); f.c = fp_one();
    } else {
        const Fp u = fp_sub(x[k * stride + off], x[(k - 1) * stride + off]);         // curr - prev   (trace.rs:800)
        f.m = fp_add(fp_one(), fp_mul(z, u));
        f.c = fp_mul(alpha, fp_sqr(u));
    }
    AffineOp::store(mc, k, f);
}
// ---- one scan over the row blocks of several devices (ss_scale_strided, ss_diluted_aggregate_block, ss_affine_apply) ----------
// A running product over rows [lo, hi) is the block's own scan times the product of the blocks before it; the aggregate's maps
// compose the same way.  A block that does not start the column leaves its item 0 the IDENTITY: the term between two blocks
// needs the row before the block, which its owner holds - the caller composes it from the two boundary values (dil_term_map).
SS_HD void scale_strided_lane(uint64_t k, Fp *data, uint64_t stride, uint64_t off, uint64_t count, const Fp &factor) {
    if (k >= count) return;
    data[k * stride + off] = fp_mul(data[k * stride + off], factor);
}
SS_HD void dil_terms_block_lane(uint64_t k, const Fp *x, uint64_t stride, uint64_t off, uint64_t count, bool starts_column, const Fp &z,
                                const Fp &alpha, Fp *mc) {
    if (k >= count) return;
    if (k == 0 && !starts_column) { AffineOp::store(mc, 0, Affine{fp_one(), fp_zero()}); return; }
    dil_terms_lane(k, x, stride, off, count, z, alpha, mc);
}
// the block's cells from its scanned maps and the value before the block: acc_k = m_k start + c_k
SS_HD void affine_apply_lane(uint64_t k, const Fp *mc, uint64_t count, const Fp &start, Fp *out, uint64_t out_stride, uint64_t out_off) {
    if (k >= count) return;
    out[k * out_stride + out_off] = fp_add(fp_mul(mc[2 * k], start), mc[2 * k + 1]);
}

inline uint64_t scan_chunks(uint64_t n, uint32_t lc) { return (n + (1ull << lc) - 1) >> lc; }
// aggregates of every level, for any chunk size >= 2^SCAN_MIN_LOG_CHUNK: sum_l ceil(n / 4^l) < n/3 + (levels <= 32)
inline uint64_t scan_agg_felts(uint64_t count, uint32_t felts) { return felts * (count / ((1ull << SCAN_MIN_LOG_CHUNK) - 1) + 64); }

// ---- level driver; Exec runs a lane body for lanes 0 .. nlanes-1 (a kernel launch, or a host loop) ----
// tmp: room for the aggregates of every level: scan_agg_felts(n, Op::FELTS) felts.  Returns Exec's status (0 = ok).
template <class Op, class Exec>
int scan_inclusive(Exec &ex, Fp *data, uint64_t n, Fp *tmp) {
    const uint32_t lc = ex.shape.log_scan;
    if (n <= (1ull << lc)) return ex.template apply<Op>(1, data, n, (const Fp *)nullptr, lc);
    const uint64_t m = scan_chunks(n, lc);
    int e = ex.template reduce<Op>(m, (const Fp *)data, n, tmp, lc);
    if (e) return e;
    if ((e = scan_inclusive<Op>(ex, tmp, m, tmp + Op::FELTS * m))) return e;
    return ex.template apply<Op>(m, data, n, (const Fp *)tmp, lc);
}

// the same scan with its cells written to a strided column (data keeps the items below the top level's prefixes: scratch)
template <class Op, class Exec>
int scan_inclusive_out(Exec &ex, Fp *data, uint64_t n, Fp *tmp, Fp *out, uint64_t out_stride, uint64_t out_off) {
    const uint32_t lc = ex.shape.log_scan;
    if (n <= (1ull << lc)) return ex.template apply_out<Op>(1, (const Fp *)data, n, (const Fp *)nullptr, lc, out, out_stride, out_off);
    const uint64_t m = scan_chunks(n, lc);
    int e = ex.template reduce<Op>(m, (const Fp *)data, n, tmp, lc);
    if (e) return e;
    if ((e = scan_inclusive<Op>(ex, tmp, m, tmp + Op::FELTS * m))) return e;
    return ex.template apply_out<Op>(m, (const Fp *)data, n, (const Fp *)tmp, lc, out, out_stride, out_off);
}

inline uint64_t permutation_product_scratch_felts(uint64_t count) { return 3 * count + scan_agg_felts(count, 1); }
inline uint64_t diluted_aggregate_scratch_felts(uint64_t count) { return 2 * count + scan_agg_felts(count, 2); }

// the whole permutation product / aggregate in terms of Exec
template <class Exec>
int permutation_product(Exec &ex, const PermOperand &num, const PermOperand &den, uint64_t count, const Fp &z, const Fp &alpha,
                        Fp *out, uint64_t out_stride, uint64_t out_off, Fp *scratch) {
    if (count == 0) return 0;
    Fp *q = scratch, *tmp = scratch + count, *aggs = scratch + 3 * count;
    // The reference scans numerators and denominators apart and divides the running products (trace.rs:752-769).  The field is
    // exact, so prod n_k / prod d_k = prod (n_k / d_k) bit for bit: invert the TERMS' denominators (the same batch inversion, the
    // same count), form the quotient terms and scan once - one scan (~ 15 launches) less per product.  A zero denominator d_k:
    // the reference's denominator_acc is zero from k on, batch_inversion leaves zeros, the column is zero from k on; here the
    // term k is zero and so is every product from k on.
    int e = ex.perm_quotients(num, den, count, z, alpha, q, tmp, ex.shape.log_inv);                 // terms, batch_inversion, n_k / d_k
    if (e) return e;
    return scan_inclusive_out<MulOp>(ex, q, count, aggs, out, out_stride, out_off);
}
// the block's scanned maps (mc: 2 count felts, kept by the caller between the two phases); aggs: scan_agg_felts(count, 2)
template <class Exec>
int diluted_aggregate_maps(Exec &ex, const Fp *x, uint64_t stride, uint64_t off, uint64_t count, bool starts_column, const Fp &z,
                           const Fp &alpha, Fp *mc, Fp *aggs) {
    if (count == 0) return 0;
    int e = ex.dil_terms_block(x, stride, off, count, starts_column, z, alpha, mc);
    if (e) return e;
    return scan_inclusive<AffineOp>(ex, mc, count, aggs);
}
template <class Exec>
int diluted_aggregate(Exec &ex, const Fp *x, uint64_t stride, uint64_t off, uint64_t count, const Fp &z, const Fp &alpha, Fp *out,
                      uint64_t out_stride, uint64_t out_off, Fp *scratch) {
    if (count == 0) return 0;
    Fp *mc = scratch, *aggs = scratch + 2 * count;
    int e = ex.dil_terms(x, stride, off, count, z, alpha, mc);
    if (e) return e;
    return scan_inclusive_out<AffineOp>(ex, mc, count, aggs, out, out_stride, out_off);      // item 0 is a constant map: so is every prefix
}

}  // namespace ss
