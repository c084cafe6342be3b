// Arithmetic of the 64-bit field p = 2^64 - 2^32 + 1 and of Fq3 = Fp[X] / (X^3 - 2) for csrc/goldilocks.hip (row X4).
//
// Host and device share every function here (tests/test_gl64_host.py compiles this header with g++ and holds it to 128-bit
// integer arithmetic), so the device's arithmetic is checked without a GPU.  Three forms of a value:
//   canonical   - the residue itself, < p: what is stored in HBM, hashed, compared;
//   lazy        - ANY 64-bit word congruent to the value (p < 2^64 < 2p: a residue has at most two such words).  The transforms'
//                 butterflies keep their sums lazy and canonicalise once per pass instead of once per addition;
//   wide        - unreduced 128-bit products added digit by digit into four 64-bit words (GlWide): a dot product pays ONE
//                 reduction for all its terms - an Fq3 product three instead of nine, a DEEP column sum three instead of nine per tap.
// 2^64 = 2^32 - 1 (EPS) and 2^96 = -1 modulo p are all the reductions need.
//
// On gfx950 a 64 x 64 -> 128 product is four v_mad_u64_u32 (quarter rate: the cost of sixteen plain instructions); writing the
// product out in 32-bit halves keeps the compiler from also computing the low word a second time with two v_mul_lo_u32
// (what `a * b` beside `__umul64hi(a, b)` compiles to: 28 issue slots instead of 16).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GL_HD __host__ __device__ __forceinline__
#else
#define GL_HD static inline
#endif

namespace ss {

static constexpr uint64_t GL_P = 0xFFFFFFFF00000001ull;
static constexpr uint64_t GL_EPS = 0xFFFFFFFFull;          // 2^64 mod p

// ---- canonical operands, canonical results
GL_HD uint64_t gl_canon(uint64_t r) { return r >= GL_P ? r - GL_P : r; }
GL_HD uint64_t gl_add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    if (s < a) s += GL_EPS;                                  // wrapped: + 2^64 = + EPS (cannot wrap again: a, b < p)
    return gl_canon(s);
}
GL_HD uint64_t gl_sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a + (GL_P - b); }

// ---- a lazy, b CANONICAL -> lazy
GL_HD uint64_t gl_add_lazy(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    if (s < a) s += GL_EPS;                                  // wrapped word <= p - 2, so + EPS stays below 2^64
    return s;
}
GL_HD uint64_t gl_sub_lazy(uint64_t a, uint64_t b) {
    uint64_t s = a - b;
    if (a < b) s -= GL_EPS;                                  // wrapped word >= 2^64 - (p - 1) = 2^32 > EPS
    return s;
}

// ---- products: any two 64-bit words in, the 128-bit product out
GL_HD uint64_t gl_mad32(uint32_t a, uint32_t b, uint64_t c) { return (uint64_t)a * b + c; }     // < 2^64 whenever c < 2^33
GL_HD void gl_mul_wide(uint64_t a, uint64_t b, uint64_t &lo, uint64_t &hi) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t p00 = gl_mad32(a0, b0, 0);
    const uint64_t p01 = gl_mad32(a0, b1, p00 >> 32);
    const uint64_t p10 = gl_mad32(a1, b0, (uint32_t)p01);
    hi = gl_mad32(a1, b1, p01 >> 32) + (p10 >> 32);
    lo = (p10 << 32) | (uint32_t)p00;
}
// lo + 2^64 hi = lo + EPS hi_lo - hi_hi, as a lazy word
GL_HD uint64_t gl_reduce128_lazy(uint64_t lo, uint64_t hi) {
    const uint64_t hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
    uint64_t t = lo - hi_hi;
    if (lo < hi_hi) t -= GL_EPS;                             // borrowed 2^64 = EPS; t >= 2^64 - 2^32 before, so no second borrow
    const uint64_t m = hi_lo * GL_EPS;                       // <= (2^32 - 1)^2
    uint64_t r = t + m;
    if (r < t) r += GL_EPS;                                  // wrapped word <= 2^64 - 2^33, so + EPS stays below 2^64
    return r;
}
GL_HD uint64_t gl_reduce128(uint64_t lo, uint64_t hi) { return gl_canon(gl_reduce128_lazy(lo, hi)); }
GL_HD uint64_t gl_mul_lazy(uint64_t a, uint64_t b) {         // lazy x lazy -> lazy
    uint64_t lo, hi;
    gl_mul_wide(a, b, lo, hi);
    return gl_reduce128_lazy(lo, hi);
}
GL_HD uint64_t gl_mul(uint64_t a, uint64_t b) { return gl_canon(gl_mul_lazy(a, b)); }            // lazy x lazy -> canonical

GL_HD uint64_t gl_pow(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    for (; e; e >>= 1) { if (e & 1) r = gl_mul(r, a); a = gl_mul(a, a); }
    return r;
}

// ---- wide accumulator: the sums of the products' four 32-bit digits, each in its own 64-bit word - no carry is ever looked at,
// and 2^32 products (or words) fit.  value = l0 + 2^32 l1 + 2^64 h0 + 2^96 h1 = l0 + 2^32 l1 + (2^32 - 1) h0 - h1 (mod p)
struct GlWide { uint64_t l0, l1, h0, h1; };
GL_HD GlWide glw_zero() { return GlWide{0, 0, 0, 0}; }
GL_HD void glw_mac(GlWide &w, uint64_t a, uint64_t b) {      // w += a b, any two words
    uint64_t lo, hi;
    gl_mul_wide(a, b, lo, hi);
    w.l0 += (uint32_t)lo; w.l1 += lo >> 32; w.h0 += (uint32_t)hi; w.h1 += hi >> 32;
}
GL_HD void glw_add(GlWide &w, uint64_t v) { w.l0 += (uint32_t)v; w.l1 += v >> 32; }
GL_HD void gl_add128(uint64_t &lo, uint64_t &hi, uint64_t alo, uint64_t ahi) {
    lo += alo;
    hi += ahi + (lo < alo);
}
GL_HD uint64_t glw_reduce(const GlWide &w) {                 // canonical
    uint64_t lo = w.l0, hi = 0;
    gl_add128(lo, hi, w.l1 << 32, w.l1 >> 32);
    gl_add128(lo, hi, w.h0 << 32, w.h0 >> 32);               // + 2^32 h0 ...
    hi -= lo < w.h0;                                         // ... - h0 (2^32 h0 >= h0: the difference is not negative)
    lo -= w.h0;
    return gl_sub(gl_reduce128(lo, hi), gl_canon(w.h1));     // < 3 2^96 before the reduction
}

// ---- Fq3: c0 + c1 X + c2 X^2, X^3 = 2
struct Gl3 { uint64_t c[3]; };
GL_HD Gl3 gl3_add(const Gl3 &a, const Gl3 &b) { return Gl3{{gl_add(a.c[0], b.c[0]), gl_add(a.c[1], b.c[1]), gl_add(a.c[2], b.c[2])}}; }
GL_HD Gl3 gl3_sub(const Gl3 &a, const Gl3 &b) { return Gl3{{gl_sub(a.c[0], b.c[0]), gl_sub(a.c[1], b.c[1]), gl_sub(a.c[2], b.c[2])}}; }
GL_HD Gl3 gl3_scale(const Gl3 &a, uint64_t s) { return Gl3{{gl_mul(a.c[0], s), gl_mul(a.c[1], s), gl_mul(a.c[2], s)}}; }
GL_HD Gl3 gl3_load(const uint64_t *p) { return Gl3{{p[0], p[1], p[2]}}; }

// three wide accumulators: sums of Fq3 products (and of Fq3 x Fp products) before any reduction
struct Gl3Wide { GlWide c[3]; };
GL_HD Gl3Wide gl3w_zero() { return Gl3Wide{{glw_zero(), glw_zero(), glw_zero()}}; }
// the factor whose upper coordinates meet X^3 = 2, doubled once: b -> (b0, b1, b2, 2 b1, 2 b2)
struct Gl3Rhs { uint64_t c[3], d1, d2; };
GL_HD Gl3Rhs gl3_rhs(const Gl3 &b) { return Gl3Rhs{{b.c[0], b.c[1], b.c[2]}, gl_add(b.c[1], b.c[1]), gl_add(b.c[2], b.c[2])}; }
GL_HD Gl3Rhs gl3_load_rhs(const uint64_t *p) { return Gl3Rhs{{p[0], p[1], p[2]}, p[3], p[4]}; }      // a factor stored in that form
GL_HD void gl3w_mac(Gl3Wide &w, const Gl3 &a, const Gl3Rhs &b) {         // w += a b
    glw_mac(w.c[0], a.c[0], b.c[0]); glw_mac(w.c[0], a.c[1], b.d2); glw_mac(w.c[0], a.c[2], b.d1);
    glw_mac(w.c[1], a.c[0], b.c[1]); glw_mac(w.c[1], a.c[1], b.c[0]); glw_mac(w.c[1], a.c[2], b.d2);
    glw_mac(w.c[2], a.c[0], b.c[2]); glw_mac(w.c[2], a.c[1], b.c[1]); glw_mac(w.c[2], a.c[2], b.c[0]);
}
GL_HD void gl3w_mac_base(Gl3Wide &w, const Gl3 &a, uint64_t s) {         // w += a s, s in Fp
    glw_mac(w.c[0], a.c[0], s); glw_mac(w.c[1], a.c[1], s); glw_mac(w.c[2], a.c[2], s);
}
GL_HD void gl3w_add(Gl3Wide &w, const Gl3 &a) { glw_add(w.c[0], a.c[0]); glw_add(w.c[1], a.c[1]); glw_add(w.c[2], a.c[2]); }
GL_HD void glw_addw(GlWide &w, const GlWide &v) { w.l0 += v.l0; w.l1 += v.l1; w.h0 += v.h0; w.h1 += v.h1; }  // the term counts add
GL_HD void gl3w_addw(Gl3Wide &w, const Gl3Wide &v) { glw_addw(w.c[0], v.c[0]); glw_addw(w.c[1], v.c[1]); glw_addw(w.c[2], v.c[2]); }
GL_HD Gl3 gl3w_reduce(const Gl3Wide &w) { return Gl3{{glw_reduce(w.c[0]), glw_reduce(w.c[1]), glw_reduce(w.c[2])}}; }
GL_HD Gl3 gl3_mul(const Gl3 &a, const Gl3 &b) {
    Gl3Wide w = gl3w_zero();
    gl3w_mac(w, a, gl3_rhs(b));
    return gl3w_reduce(w);
}

// a^-1 = adj(a) / N(a): adj = (a0^2 - 2 a1 a2, 2 a2^2 - a0 a1, a1^2 - a0 a2), N = a0 adj0 + 2 (a2 adj1 + a1 adj2) in Fp; 0 -> 0
GL_HD Gl3 gl3_inv(const Gl3 &a) {
    const uint64_t a12 = gl_mul(a.c[1], a.c[2]), a22 = gl_mul(a.c[2], a.c[2]);
    Gl3 adj;
    adj.c[0] = gl_sub(gl_mul(a.c[0], a.c[0]), gl_add(a12, a12));
    adj.c[1] = gl_sub(gl_add(a22, a22), gl_mul(a.c[0], a.c[1]));
    adj.c[2] = gl_sub(gl_mul(a.c[1], a.c[1]), gl_mul(a.c[0], a.c[2]));
    const uint64_t t1 = gl_mul(a.c[2], adj.c[1]), t2 = gl_mul(a.c[1], adj.c[2]);
    const uint64_t norm = gl_add(gl_mul(a.c[0], adj.c[0]), gl_add(gl_add(t1, t1), gl_add(t2, t2)));
    return gl3_scale(adj, gl_pow(norm, GL_P - 2));
}

}  // namespace ss
