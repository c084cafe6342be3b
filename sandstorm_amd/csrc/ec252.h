// ec252.h — the StarkWare curve y^2 = x^3 + x + beta over Fp in Jacobian coordinates, host+device
// (pedersen.hip; tests/cpp/ec_lazy_test.cpp checks the lazy forms against the plain ones on the host).
#pragma once
#include "fp252.h"
#include "fl252.h"

namespace ss {

struct Aff { Fp x, y; };
struct Jac { Fp x, y, z; };

// ---- plain 8 x 32 form (table construction on the host, reference for the lazy forms) ----
SS_HD Jac jac_double(const Jac &p) {
    Fp xx = fp_sqr(p.x), yy = fp_sqr(p.y), yyyy = fp_sqr(yy), zz = fp_sqr(p.z);
    Fp s = fp_mul(p.x, yy); s = fp_dbl(fp_dbl(s));
    Fp m = fp_add(fp_add(fp_dbl(xx), xx), fp_sqr(zz));
    Jac r;
    r.x = fp_sub(fp_sqr(m), fp_dbl(s));
    r.y = fp_sub(fp_mul(m, fp_sub(s, r.x)), fp_dbl(fp_dbl(fp_dbl(yyyy))));
    r.z = fp_dbl(fp_mul(p.y, p.z));
    return r;
}
// p + q, q affine.  The exceptional cases (p = +-q) are handled: doubling, or
// the point at infinity encoded as z = 0.
SS_HD Jac jac_add_aff(const Jac &p, const Aff &q) {
    if (fp_is_zero(p.z)) { Jac r; r.x = q.x; r.y = q.y; r.z = fp_one(); return r; }
    Fp zz = fp_sqr(p.z);
    Fp u2 = fp_mul(q.x, zz), s2 = fp_mul(q.y, fp_mul(zz, p.z));
    Fp h = fp_sub(u2, p.x), rr = fp_sub(s2, p.y);
    if (fp_is_zero(h)) {
        if (fp_is_zero(rr)) return jac_double(p);
        Jac o; o.x = fp_one(); o.y = fp_one(); o.z = fp_zero(); return o;
    }
    Fp hh = fp_sqr(h), hhh = fp_mul(hh, h), v = fp_mul(p.x, hh);
    Jac r;
    r.x = fp_sub(fp_sub(fp_sqr(rr), hhh), fp_dbl(v));
    r.y = fp_sub(fp_mul(rr, fp_sub(v, r.x)), fp_mul(p.y, hhh));
    r.z = fp_mul(p.z, h);
    return r;
}

// The lazy forms below are written once over the DOMAIN their values live in: EcR256 - Montgomery images x * 2^256, fl_mul's reduction
// (what the tables are BUILT in and the host tests check against the plain 8 x 32 form) - or EcR280 - x * 2^280, the ten-step
// reduction alone (fl252.h: 25 instructions fewer per product; the Pedersen kernels' accumulation, whose table points are stored
// in this form: round 6).  The formulas never mix a constant in: the domain only decides the product, the square and the one.
struct EcR256 {
    static SS_HD Fl mul(const Fl &a, const Fl &b) { return fl_mul(a, b); }
    static SS_HD Fl sqr(const Fl &a) { return fl_sqr(a); }
    static SS_HD Fl one() { return fl_one(); }
};
struct EcR280 {
    static SS_HD Fl mul(const Fl &a, const Fl &b) { return fl_mul_r280(a, b); }
    static SS_HD Fl sqr(const Fl &a) { return fl_sqr_r280(a); }
    static SS_HD Fl one() { return fl_one_r280(); }
};

// The same formulas in the lazy 9 x 28-bit form (fl252.h "safe" ops: every value stays
// normalised and < 2p).  Used by the device kernels; 1.6x the throughput of the 8 x 32 form.
struct JacL { Fl x, y, z; };
struct AffL { Fl x, y; };

template <class D> SS_HD JacL jacl_double_d(const JacL &p) {
    const Fl xx = D::sqr(p.x), yy = D::sqr(p.y), yyyy = D::sqr(yy), zz = D::sqr(p.z);
    const Fl s = fn_dbl(fn_dbl(D::mul(p.x, yy)));
    const Fl m = fn_add(fn_add(fn_dbl(xx), xx), D::sqr(zz));
    JacL r;
    r.x = fn_sub(D::sqr(m), fn_dbl(s));
    r.y = fn_sub(D::mul(m, fn_sub(s, r.x)), fn_dbl(fn_dbl(fn_dbl(yyyy))));
    r.z = fn_dbl(D::mul(p.y, p.z));
    return r;
}
// p + q, q affine (8M + 3S).  Lazy bounds: only the two coordinates that must be subtrahends again (x3, y3) are
// weakly reduced; h, r, the partial sums of x3 and (v - x3) go into the multiplier unreduced:
//   a - b + 2p (fl_sub_c<2,1>) of normalised a, b < 2p      : value < 4p, limbs < 2^29 + 2^25
//   products of two such values                              : < 16 p^2 / 2^256 + p < 1.51 p, normalised; the
//                                                              81 partial products sum below 9 * 2^58.3 < 2^62
//   x3 = rr^2 - hhh - 2v as (a - b + 2p) - (v + v) + 8p      : value < 12p, limbs < 2^30, then one weak reduction
// h = 0 (the exceptional cases p = +-q) is detected on z3 = z1 h, which is normalised anyway.
template <class D> SS_HD JacL jacl_add_aff_d(const JacL &p, const AffL &q) {
    if (fn_is_zero(p.z)) { JacL r; r.x = q.x; r.y = q.y; r.z = D::one(); return r; }
    const Fl zz = D::sqr(p.z);
    const Fl u2 = D::mul(q.x, zz), s2 = D::mul(q.y, D::mul(zz, p.z));
    const Fl h = fl_sub_c<2, 1>(u2, p.x), rr = fl_sub_c<2, 1>(s2, p.y);          // lazy: < 4p
    JacL r;
    r.z = D::mul(h, p.z);
    if (fn_is_zero(r.z)) {                                                        // z1 != 0, so h = 0 (mod p)
        if (fn_is_zero(fl_weak_reduce(rr))) return jacl_double_d<D>(p);
        JacL o; o.x = D::one(); o.y = D::one(); o.z = fl_zero(); return o;
    }
    // squares of the lazy h, rr (< 4p, limbs < 2^29 + 2^25): fl_sqr's doubled limbs stay below 2^31 and its 45 products sum
    // to what fl_mul's 81 would - the same column bounds with 36 multiplications fewer each
    const Fl hh = D::sqr(h), hhh = D::mul(h, hh), v = D::mul(p.x, hh);
    r.x = fl_weak_reduce(fl_sub_c<8, 2>(fl_sub_c<2, 1>(D::sqr(rr), hhh), fl_add(v, v)));
    r.y = fl_weak_reduce(fl_sub_c<2, 1>(D::mul(rr, fl_sub_c<2, 1>(v, r.x)), D::mul(p.y, hhh)));
    return r;
}

// XYZZ coordinates (x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2): the mixed addition is 8M + 2S - one squaring fewer than the Jacobian
// one above, which squares z1 first - and a hash only wants x = X / ZZ at its end, so nothing is owed for the fourth coordinate
// but nine registers (round 5; the accumulate kernels of pedersen.hip: one hash per lane, 20 - 22 additions in a row).
// Infinity is ZZ = 0.  The lazy discipline is jacl_add_aff's, value for value: u2, s2, v and the two new coordinates are plain
// products (normalised, < 2p); h, rr < 4p go into the multiplier unreduced; x3, y3 are weakly reduced.
struct XyzzL { Fl x, y, zz, zzz; };
template <class D> SS_HD XyzzL xyzzl_double_d(const XyzzL &p) {                 // dbl-2008-s-1 with a = 1
    const Fl u = fn_dbl(p.y), v = D::sqr(u), w = D::mul(u, v), s = D::mul(p.x, v);
    const Fl xx = D::sqr(p.x), m = fn_add(fn_add(fn_dbl(xx), xx), D::sqr(p.zz));
    XyzzL r;
    r.x = fn_sub(D::sqr(m), fn_dbl(s));
    r.y = fn_sub(D::mul(m, fn_sub(s, r.x)), D::mul(w, p.y));
    r.zz = D::mul(v, p.zz);
    r.zzz = D::mul(w, p.zzz);
    return r;
}
template <class D> SS_HD XyzzL xyzzl_add_aff_d(const XyzzL &p, const AffL &q) {  // madd-2008-s
    if (fn_is_zero(p.zz)) { XyzzL r; r.x = q.x; r.y = q.y; r.zz = D::one(); r.zzz = D::one(); return r; }
    const Fl u2 = D::mul(q.x, p.zz), s2 = D::mul(q.y, p.zzz);
    const Fl h = fl_sub_c<2, 1>(u2, p.x), rr = fl_sub_c<2, 1>(s2, p.y);          // lazy: < 4p
    const Fl hh = D::sqr(h);
    XyzzL r;
    r.zz = D::mul(p.zz, hh);
    if (fn_is_zero(r.zz)) {                                                       // zz1 != 0, so h = 0 (mod p)
        if (fn_is_zero(fl_weak_reduce(rr))) return xyzzl_double_d<D>(p);
        XyzzL o; o.x = D::one(); o.y = D::one(); o.zz = fl_zero(); o.zzz = fl_zero(); return o;
    }
    const Fl hhh = D::mul(h, hh), v = D::mul(p.x, hh);
    r.x = fl_weak_reduce(fl_sub_c<8, 2>(fl_sub_c<2, 1>(D::sqr(rr), hhh), fl_add(v, v)));
    r.y = fl_weak_reduce(fl_sub_c<2, 1>(D::mul(rr, fl_sub_c<2, 1>(v, r.x)), D::mul(p.y, hhh)));
    r.zzz = D::mul(p.zzz, hhh);
    return r;
}

// p + q, both affine and neither the point at infinity (mmadd-2008-s: 4M + 2S) -> XYZZ.  The first addition of a hash: its
// accumulator is still the affine shift point, and the generic form would multiply by ZZ = ZZZ = 1 four times.
template <class D> SS_HD XyzzL xyzzl_add_affs_d(const AffL &p, const AffL &q) {
    const Fl h = fl_sub_c<2, 1>(q.x, p.x), rr = fl_sub_c<2, 1>(q.y, p.y);        // lazy: < 4p
    const Fl hh = D::sqr(h);
    if (fn_is_zero(hh)) {                                                         // p = +-q: the generic form knows what to do
        XyzzL a; a.x = p.x; a.y = p.y; a.zz = D::one(); a.zzz = D::one();
        return xyzzl_add_aff_d<D>(a, q);
    }
    const Fl hhh = D::mul(h, hh), v = D::mul(p.x, hh);
    XyzzL r;
    r.x = fl_weak_reduce(fl_sub_c<8, 2>(fl_sub_c<2, 1>(D::sqr(rr), hhh), fl_add(v, v)));
    r.y = fl_weak_reduce(fl_sub_c<2, 1>(D::mul(rr, fl_sub_c<2, 1>(v, r.x)), D::mul(p.y, hhh)));
    r.zz = hh;
    r.zzz = hhh;
    return r;
}

// p + q, both Jacobian (12M + 4S); infinity is z = 0 on either side.  Used by the lane-split
// accumulation of small tree levels, where partial sums of one hash meet across lanes.
template <class D> SS_HD JacL jacl_add_d(const JacL &p, const JacL &q) {
    const bool pinf = fn_is_zero(p.z), qinf = fn_is_zero(q.z);
    const Fl z1z1 = D::sqr(p.z), z2z2 = D::sqr(q.z);
    const Fl u1 = D::mul(p.x, z2z2), u2 = D::mul(q.x, z1z1);
    const Fl s1 = D::mul(p.y, D::mul(q.z, z2z2)), s2 = D::mul(q.y, D::mul(p.z, z1z1));
    const Fl h = fn_sub(u2, u1), rr = fn_sub(s2, s1);
    if (pinf) return q;
    if (qinf) return p;
    if (fn_is_zero(h)) {
        if (fn_is_zero(rr)) return jacl_double_d<D>(p);
        JacL o; o.x = D::one(); o.y = D::one(); o.z = fl_zero(); return o;
    }
    const Fl hh = D::sqr(h), hhh = D::mul(hh, h), v = D::mul(u1, hh);
    JacL r;
    r.x = fn_sub(fn_sub(D::sqr(rr), hhh), fn_dbl(v));
    r.y = fn_sub(D::mul(rr, fn_sub(v, r.x)), D::mul(s1, hhh));
    r.z = D::mul(D::mul(p.z, q.z), h);
    return r;
}

// p + q, both affine (4M + 2S), either possibly the point at infinity (flagged: an affine point has no room for it) -> Jacobian.
// The first round of the lane-split accumulation: 32 table points meet pairwise.
template <class D> SS_HD JacL jacl_add_affs_d(const AffL &p, bool pinf, const AffL &q, bool qinf) {
    JacL r;
    if (pinf || qinf) {
        if (pinf && qinf) { r.x = D::one(); r.y = D::one(); r.z = fl_zero(); return r; }
        const AffL &o = pinf ? q : p;
        r.x = o.x; r.y = o.y; r.z = D::one();
        return r;
    }
    const Fl h = fn_sub(q.x, p.x), rr = fn_sub(q.y, p.y);
    if (fn_is_zero(h)) {
        if (fn_is_zero(rr)) { JacL d; d.x = p.x; d.y = p.y; d.z = D::one(); return jacl_double_d<D>(d); }
        r.x = D::one(); r.y = D::one(); r.z = fl_zero(); return r;
    }
    const Fl hh = D::sqr(h), hhh = D::mul(hh, h), v = D::mul(p.x, hh);
    r.x = fn_sub(fn_sub(D::sqr(rr), hhh), fn_dbl(v));
    r.y = fn_sub(D::mul(rr, fn_sub(v, r.x)), D::mul(p.y, hhh));
    r.z = h;
    return r;
}


// the R256 forms under their old names (table construction, tests/cpp/ec_lazy_test.cpp)
SS_HD JacL jacl_double(const JacL &p) { return jacl_double_d<EcR256>(p); }
SS_HD JacL jacl_add_aff(const JacL &p, const AffL &q) { return jacl_add_aff_d<EcR256>(p, q); }
SS_HD XyzzL xyzzl_double(const XyzzL &p) { return xyzzl_double_d<EcR256>(p); }
SS_HD XyzzL xyzzl_add_aff(const XyzzL &p, const AffL &q) { return xyzzl_add_aff_d<EcR256>(p, q); }
SS_HD JacL jacl_add(const JacL &p, const JacL &q) { return jacl_add_d<EcR256>(p, q); }
SS_HD JacL jacl_add_affs(const AffL &p, bool pinf, const AffL &q, bool qinf) { return jacl_add_affs_d<EcR256>(p, pinf, q, qinf); }

}  // namespace ss
