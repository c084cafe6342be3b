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
struct MulOp {                       // field multiplication
    typedef Fp T;
    static constexpr uint32_t FELTS = 1;
    static SS_HD T identity() { return fp_one(); }
    static SS_HD T load(const Fp *base, uint64_t i) { return base[i]; }
    static SS_HD void store(Fp *base, uint64_t i, const T &x) { base[i] = x; }
    static SS_HD T combine(const T &acc, const T &x) { return fp_mul(acc, x); }
};
struct Affine { Fp m, c; };          // t -> m t + c
struct AffineOp {                    // "then": (acc then x)(t) = x.m (acc.m t + acc.c) + x.c
    typedef Affine T;
    static constexpr uint32_t FELTS = 2;
    static SS_HD T identity() { return Affine{fp_one(), fp_zero()}; }
    static SS_HD T load(const Fp *base, uint64_t i) { return Affine{base[2 * i], base[2 * i + 1]}; }
    static SS_HD void store(Fp *base, uint64_t i, const T &x) { base[2 * i] = x.m; base[2 * i + 1] = x.c; }
    static SS_HD T combine(const T &acc, const T &x) { return Affine{fp_mul(acc.m, x.m), fp_add(fp_mul(acc.c, x.m), x.c)}; }
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

// zero-preserving element-wise inversion (ark-ff batch_inversion semantics); tmp: n felts.
// Prefix products of the non-zero entries of the chunk, one inversion, back-substitution.
SS_HD void inverse_dense_lane(uint64_t c, Fp *data, uint64_t n, Fp *tmp, uint32_t lc) {
    // chunk c = elements c, c + m, c + 2 m, ... (m chunks): neighbouring lanes touch neighbouring elements
    const uint64_t m = (n + (1ull << lc) - 1) >> lc;
    if (c >= m) return;
    Fl run = fl_one();
    for (uint64_t i = c; i < n; i += m) {
        const Fl v = fl_from_fp(data[i]);
        tmp[i] = fl_pack(run);                    // (a weakly reduced image: only read back below)
        if (!fn_is_zero(v)) run = fn_mul(run, v);
    }
    Fl inv = fl_from_fp(fp_inv_safegcd(fl_to_fp(run)));   // run is a product of non-zero entries (or 1)
    uint64_t last = c + ((n - 1 - c) / m) * m;            // the chunk's last element
    for (uint64_t i = last;; i -= m) {
        const Fl v = fl_from_fp(data[i]);
        if (!fn_is_zero(v)) {                     // a zero stays zero
            data[i] = fl_to_fp(fn_mul(inv, fl_from_fp(tmp[i])));
            inv = fn_mul(inv, v);
        }
        if (i == c) break;
    }
}

SS_HD Fp perm_term(const PermOperand &o, uint64_t k, const Fp &z, const Fp &alpha) {
    const Fp *item = o.data + k * o.stride;
    const Fp a = item[o.a_off];
    if (o.v_off < 0) return fp_sub(z, a);                                    // z - x_k           (trace.rs:727-728)
    return fp_sub(z, fp_add(fp_mul(alpha, item[o.v_off]), a));               // z - (alpha v + a)  (trace.rs:713-714)
}
SS_HD void perm_terms_lane(uint64_t k, const PermOperand &num, const PermOperand &den, uint64_t count, const Fp &z,
                           const Fp &alpha, Fp *tn, Fp *td) {
    if (k >= count) return;
    tn[k] = perm_term(num, k, z, alpha);
    td[k] = perm_term(den, k, z, alpha);
}
SS_HD void perm_finish_lane(uint64_t k, const Fp *pn, const Fp *pd_inv, uint64_t count, Fp *out, uint64_t out_stride,
                            uint64_t out_off) {
    if (k >= count) return;
    // n * d_inv (trace.rs:767-769); without pd_inv: the strided copy of a finished column.  (out may be pn itself when the strides
    // are 1: a lane reads and writes the same element)
    out[k * out_stride + out_off] = pd_inv ? fp_mul(pn[k], pd_inv[k]) : pn[k];
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
SS_HD void dil_finish_lane(uint64_t k, const Fp *mc, uint64_t count, Fp *out, uint64_t out_stride, uint64_t out_off) {
    if (k >= count) return;
    out[k * out_stride + out_off] = mc[2 * k + 1];                           // the map is constant: its value is c
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
    if (k == 0 && !starts_column) { AffineOp::store(mc, 0, AffineOp::identity()); return; }
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

inline uint64_t permutation_product_scratch_felts(uint64_t count) { return 3 * count + scan_agg_felts(count, 1); }
inline uint64_t diluted_aggregate_scratch_felts(uint64_t count) { return 2 * count + scan_agg_felts(count, 2); }

// the whole permutation product / aggregate in terms of Exec
template <class Exec>
int permutation_product(Exec &ex, const PermOperand &num, const PermOperand &den, uint64_t count, const Fp &z, const Fp &alpha,
                        Fp *out, uint64_t out_stride, uint64_t out_off, Fp *scratch) {
    if (count == 0) return 0;
    Fp *tn = scratch, *td = scratch + count, *tmp = scratch + 2 * count, *aggs = scratch + 3 * count;
    int e = ex.perm_terms(num, den, count, z, alpha, tn, td);
    if (e) return e;
    // The reference scans numerators and denominators apart and divides the running products (trace.rs:752-769).  The field is
    // exact, so prod n_k / prod d_k = prod (n_k / d_k) bit for bit: invert the TERMS' denominators (the same batch inversion, the
    // same count), form the quotient terms and scan once - one scan (~ 15 launches) less per product.  A zero denominator d_k:
    // the reference's denominator_acc is zero from k on, batch_inversion leaves zeros, the column is zero from k on; here the
    // term k is zero and so is every product from k on.
    if ((e = ex.inverse_dense(scan_chunks(count, ex.shape.log_inv), td, count, tmp, ex.shape.log_inv))) return e;     // batch_inversion
    if ((e = ex.perm_finish((const Fp *)tn, (const Fp *)td, count, tn, 1, 0))) return e;                               // n_k / d_k
    if ((e = scan_inclusive<MulOp>(ex, tn, count, aggs))) return e;
    return ex.perm_finish((const Fp *)tn, (const Fp *)nullptr, count, out, out_stride, out_off);
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
    if ((e = scan_inclusive<AffineOp>(ex, mc, count, aggs))) return e;
    return ex.dil_finish((const Fp *)mc, count, out, out_stride, out_off);
}

}  // namespace ss
