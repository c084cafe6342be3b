// fl252.h — lazy 9 x 28-bit limb arithmetic for the StarkWare prime on gfx950.
//
// Why: measured on MI355X (profiles/r01_ubench_instruction_rates.txt) v_mad_u64_u32
// issues at ~half rate (5.3 cycles per wave-instruction) — but any VALU that
// consumes VCC needs 2 wait states after the VALU that produced it, and 64-bit adds
// are themselves half rate, so in the 8 x 32-bit form (fp252.h) the 64 multiplies of a
// product are outweighed by ~300 carry / move instructions.  With 28-bit limbs a
// column of the schoolbook product is a sum of <= 9 products < 2^57 (+ reduction
// terms): it fits a u64 with room to spare, so every partial product is ONE
// v_mad_u64_u32 accumulating in place — no carries, no moves — and additions /
// subtractions are 9 independent 32-bit adds with no carry chain at all.
//
// Representation (type Fl): value = sum l[i] * 2^(28 i); limbs are "lazy": l[i] may
// exceed 28 bits (bounds below).  Montgomery form with R = 2^256, exactly the
// memory format's R, so converting to/from the 8 x 32-bit image is pure bit
// repacking (+ a final reduction on the way out).
//
// Invariants ("value bound" V in units of p, "limb bound" L):
//   fl_mul inputs : limbs < 2^30 (a) and < 2^29 (b), so 9 products < 2^59 * 9 < 2^63
//   fl_mul output : normalised limbs < 2^28 (top limb < 2^29), value < a*b/2^256 + p
//                   -> < 1.25 p for a < 8p, b < p (twiddles);  < 3 p for a, b < 8 p
//   fl_add        : limb-wise, no carries; bounds add
//   fl_sub(a,b,k) : a - b + 2^k p computed limb-wise with the borrow-proof constant KP[k]
//   fl_weak_reduce: normalise limbs and bring the value below 2 p
#pragma once
#include "fp252.h"

namespace ss {

struct Fl {
    u32 l[9];
};

static constexpr u32 FL_MASK = (1u << 28) - 1u;

// p = 2^251 + 17*2^192 + 1 in 28-bit limbs: l0 = 1, l6 = 2^24 (bit 192), l7 = 1 (bit 196), l8 = 2^27 (bit 251)
#define FL_P0 1u
#define FL_P6 (1u << 24)
#define FL_P7 1u
#define FL_P8 (1u << 27)

// 8 x 32 (canonical Montgomery image) -> 9 x 28
SS_HD Fl fl_from_fp(const Fp &a) {
    Fl r;
    r.l[0] = a.v[0] & FL_MASK;
    r.l[1] = ((a.v[0] >> 28) | (a.v[1] << 4)) & FL_MASK;
    r.l[2] = ((a.v[1] >> 24) | (a.v[2] << 8)) & FL_MASK;
    r.l[3] = ((a.v[2] >> 20) | (a.v[3] << 12)) & FL_MASK;
    r.l[4] = ((a.v[3] >> 16) | (a.v[4] << 16)) & FL_MASK;
    r.l[5] = ((a.v[4] >> 12) | (a.v[5] << 20)) & FL_MASK;
    r.l[6] = ((a.v[5] >> 8) | (a.v[6] << 24)) & FL_MASK;
    r.l[7] = (a.v[6] >> 4) & FL_MASK;
    r.l[8] = a.v[7];                         // bits 224..255 (lazy top limb)
    return r;
}

// carry-propagate so that l[0..7] < 2^28; the top limb absorbs the rest
SS_HD Fl fl_normalize(const Fl &a) {
    Fl r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const u32 t = a.l[i] + c;
        r.l[i] = t & FL_MASK;
        c = t >> 28;
    }
    r.l[8] = a.l[8] + c;
    return r;
}

// normalised limbs, value < 2^256 -> 8 x 32 image (NOT reduced mod p)
SS_HD Fp fl_pack(const Fl &n) {
    Fp r;
    r.v[0] = n.l[0] | (n.l[1] << 28);
    r.v[1] = (n.l[1] >> 4) | (n.l[2] << 24);
    r.v[2] = (n.l[2] >> 8) | (n.l[3] << 20);
    r.v[3] = (n.l[3] >> 12) | (n.l[4] << 16);
    r.v[4] = (n.l[4] >> 16) | (n.l[5] << 12);
    r.v[5] = (n.l[5] >> 20) | (n.l[6] << 8);
    r.v[6] = (n.l[6] >> 24) | (n.l[7] << 4);
    r.v[7] = n.l[8];
    return r;
}

// value -> value - q*p with q = max(floor(value / 2^251) - 1, 0) read off the top limb
// (the "- 1" because p is slightly above 2^251: floor(value/2^251) * p may exceed the
// value).  Result: normalised, in [0, 2^252) i.e. < 2p, for any lazy value < 2^256.
// 32-bit signed limb arithmetic with arithmetic-shift carries: no VCC, no 64-bit ops.
SS_HD Fl fl_weak_reduce(const Fl &a) {
    const Fl n = fl_normalize(a);
    const int q0 = (int)(n.l[8] >> 27);               // <= 31
    const int q = q0 - (q0 > 0 ? 1 : 0);
    Fl r;
    int t = (int)n.l[0] - q;                          // q*p = q + q 2^192 + q 2^196 + q 2^251
    r.l[0] = (u32)t & FL_MASK;
    int c = t >> 28;
#pragma unroll
    for (int i = 1; i < 6; ++i) { t = (int)n.l[i] + c; r.l[i] = (u32)t & FL_MASK; c = t >> 28; }
    t = (int)n.l[6] - (q << 24) + c; r.l[6] = (u32)t & FL_MASK; c = t >> 28;
    t = (int)n.l[7] - q + c;         r.l[7] = (u32)t & FL_MASK; c = t >> 28;
    r.l[8] = n.l[8] - ((u32)q << 27) + (u32)c;           // unsigned: the top limb of a value near 2^256 is above 2^31 (wraps are meant)
    return r;
}

// The same in ONE carry chain, for a transform pass's store (ntt.hip): any lazy value < 2^256 whose limbs 0..7 are below
// 15 * 2^28 - 32 -> normalised limbs, value < 2^252 + 2^229 (top limb < 2^28 + 2^5).  The quotient is read off the
// UN-normalised top limb, q = max(floor(l8 / 2^27) - 1, 0) <= floor(value / p), and value - q p is computed as
// value - (q + 1) 2^251 + E with E = 2^251 - q (17 2^192 + 1) >= 0 written so that every limb is positive:
//   e0 = 2^28 - q, e1..e5 = 2^28 - 1, e6 = 2^28 - 1 - (q & 15) 2^24, e7 = 2^28 - 1 - q - (q >> 4), e8 = 2^27 - 1
// (q 2^192 = (q & 15) 2^24 2^168 + (q >> 4) 2^196), so the chain is unsigned adds, masks and shifts: no borrows, no
// compares, 37 instead of 54 vector instructions.  The top limb is exact modulo 2^32 and the true result fits it.
SS_HD Fl fl_weak_reduce1(const Fl &a) {
    const u32 q0 = a.l[8] >> 27;
    const u32 q = q0 ? q0 - 1u : 0u;
    Fl r;
    u32 t = a.l[0] + ((1u << 28) - q);
    r.l[0] = t & FL_MASK;
    u32 c = t >> 28;
#pragma unroll
    for (int i = 1; i < 6; ++i) { t = a.l[i] + FL_MASK + c; r.l[i] = t & FL_MASK; c = t >> 28; }
    t = a.l[6] + (FL_MASK - ((q & 15u) << 24)) + c; r.l[6] = t & FL_MASK; c = t >> 28;
    t = a.l[7] + (FL_MASK - q - (q >> 4)) + c;      r.l[7] = t & FL_MASK; c = t >> 28;
    r.l[8] = a.l[8] + ((1u << 27) - 1u) - ((q + 1u) << 27) + c;
    return r;
}

// fully reduced 8 x 32 image (< p) of a lazy value < 32 p
SS_HD Fp fl_to_fp(const Fl &a) {
    Fl w = fl_weak_reduce(a);                 // < 2p, normalised
    return fp_reduce_once(fl_pack(w));
}

SS_HD Fl fl_add(const Fl &a, const Fl &b) {
    Fl r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
}

// a - b + C*p, limb-wise and borrow-free: C*p is rewritten so that every limb is at
// least F*2^28 (each 2^28-unit borrowed at limb i is paid back as -F at limb i+1):
//   d0 = F 2^28 + C, d1..d5 = F 2^28 - F, d6 = F 2^28 - F + C 2^24, d7 = F 2^28 - F + C,
//   d8 = C 2^27 - F          (sum d_i 2^(28 i) = C p)
// valid when b's limbs 0..7 are <= F 2^28 - F and its top limb <= C 2^27 - F:
//   <C=2,  F=1>: b = a normalised product (fl_mul output, value < 2p)
//   <C=8,  F=2>: b's limbs < 2^29, value < 8p   (sum of two normalised values)
//   <C=16, F=4>: b's limbs < 2^30, value < 16p
template <u32 C, u32 F>
SS_HD Fl fl_sub_c(const Fl &a, const Fl &b) {
    constexpr u32 base = F << 28;
    Fl r;
    r.l[0] = a.l[0] + (base + C) - b.l[0];
#pragma unroll
    for (int i = 1; i < 6; ++i) r.l[i] = a.l[i] + (base - F) - b.l[i];
    r.l[6] = a.l[6] + (base - F + (C << 24)) - b.l[6];
    r.l[7] = a.l[7] + (base - F + C) - b.l[7];
    r.l[8] = a.l[8] + ((C << 27) - F) - b.l[8];
    return r;
}
SS_HD Fl fl_sub8p(const Fl &a, const Fl &b) { return fl_sub_c<8, 2>(a, b); }

// A value the optimiser cannot see through: keeps `m * 2^24` a v_mad_u64_u32 that
// accumulates in place instead of a 64-bit shift + 64-bit add pair.
SS_HD u32 fl_opaque(u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#endif
    return x;
}

// Montgomery reduction of 17 lazy 64-bit columns (weights 2^(28 k)) by R = 2^256 =
// 2^(9*28) * 2^4; output normalised (limbs < 2^28, top limb small).
SS_HD Fl fl_mont_reduce(u64 (&c)[18]) {
    const u32 k17 = fl_opaque(17u << 24), k27 = fl_opaque(1u << 27), k1 = fl_opaque(1u);
    // nine 28-bit Montgomery steps: p = 1 (mod 2^28) so m = -c[i] mod 2^28.  The columns are lazy 64-bit sums, so m * 17 * 2^192 goes
    // into ONE of them as m * (17 << 24) < 2^57 (limbs 6 and 7 of p at once: a multiply-add less per step; round 6)
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const u32 m = (0u - (u32)c[i]) & FL_MASK;
        c[i + 1] += ((u64)m * k1 + c[i]) >> 28;       // limb i becomes 0; its carry moves up
        c[i + 6] += (u64)m * k17;
        c[i + 8] += (u64)m * k27;
    }
    // one 4-bit step to complete R = 2^256
    {
        const u32 m = (0u - (u32)c[9]) & 15u;
        c[9] += m;
        c[15] += (u64)m * k17;
        c[17] += (u64)m * k27;
    }
    // shift right by 4 across columns and normalise
    Fl r;
    u64 carry = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const u64 t = (c[9 + j] >> 4) + ((c[10 + j] & 15ull) << 24) + carry;
        r.l[j] = (u32)t & FL_MASK;
        carry = t >> 28;
    }
    r.l[8] = (u32)((c[17] >> 4) + carry);
    return r;
}

// Montgomery product a*b*2^-256 mod p: 81 in-place multiply-adds, then the reduction.
// a's limbs may be any u32 (value < 2^256); b's limbs must be < 2^28 (top limb < 2^29).
SS_HD Fl fl_mul(const Fl &a, const Fl &b) {
    u64 c[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) c[i + j] += (u64)a.l[i] * b.l[j];
    return fl_mont_reduce(c);
}

// a * t * 2^-280 mod p for a multiplier t the CALLER controls (twiddles, coefficients, tables):
// such operands are kept as t * 2^280 mod p in limb form ("R280 form": fl_to_r280), so that
// a (in the interchange domain, x * 2^256) times t comes out in the interchange domain again.
// 2^280 = 2^(10 * 28): the reduction is exactly ten 28-bit steps and the result is columns
// 10..18 as they stand - no 4-bit tail step and no 4-bit shift across columns (that tail is
// ~50 of fl_mul's 223 VALU instructions), and the operand needs no re-limbing.  Same
// preconditions as fl_mul (a any u32 limbs with value < 2^256, t normalised and canonical) and a
// stronger contraction: result normalised and < a t / 2^280 + p < 1.01 p.
SS_HD Fl fl_mul_r280(const Fl &a, const Fl &t) {
    u64 c[19];
#pragma unroll
    for (int k = 0; k < 19; ++k) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) c[i + j] += (u64)a.l[i] * t.l[j];
    const u32 k17 = fl_opaque(17u << 24), k27 = fl_opaque(1u << 27), k1 = fl_opaque(1u);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const u32 m = (0u - (u32)c[i]) & FL_MASK;
        c[i + 1] += ((u64)m * k1 + c[i]) >> 28;       // limb i becomes 0; its carry moves up
        c[i + 6] += (u64)m * k17;                     // m * 17 * 2^192: limbs 6 and 7 of p into one lazy column (fl_mont_reduce)
        c[i + 8] += (u64)m * k27;
    }
    Fl r;
    u64 carry = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const u64 v = c[10 + j] + carry;
        r.l[j] = (u32)v & FL_MASK;
        carry = v >> 28;
    }
    r.l[8] = (u32)(c[18] + carry);
    return r;
}
// ---- fused dot products: sum_k a_k * t_k (t_k in R280 form) with ONE Montgomery reduction.  The 81 partial products of
// every term go straight into 19 64-bit columns; with normalised 28-bit limbs on both sides a column grows by < 9 * 2^56 per
// term, so FL_WIDE_MAX_TERMS terms stay below 2^64 with room for the reduction's own terms.  fl_wide_reduce is the ten-step
// R280 reduction of fl_mul_r280: result normalised, < terms * 2p * p / 2^280 + p < 1.01 p.  A term costs its 81
// v_mad_u64_u32 and nothing else - the reduction (more than half of a product's instructions) is paid once per sum.
struct FlWide { u64 c[19]; };
static constexpr int FL_WIDE_MAX_TERMS = 16;
SS_HD void fl_wide_zero(FlWide &w) {
#pragma unroll
    for (int k = 0; k < 19; ++k) w.c[k] = 0;
}
SS_HD void fl_wide_mad(FlWide &w, const Fl &a, const Fl &t) {
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) w.c[i + j] += (u64)a.l[i] * t.l[j];
}
SS_HD Fl fl_wide_reduce(FlWide &w) {
    const u32 k17 = fl_opaque(17u << 24), k27 = fl_opaque(1u << 27), k1 = fl_opaque(1u);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const u32 m = (0u - (u32)w.c[i]) & FL_MASK;
        w.c[i + 1] += ((u64)m * k1 + w.c[i]) >> 28;
        w.c[i + 6] += (u64)m * k17;                   // (a column holds <= 144 products < 2^56 and this < 2^57: below 2^64)
        w.c[i + 8] += (u64)m * k27;
    }
    Fl r;
    u64 carry = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const u64 v = w.c[10 + j] + carry;
        r.l[j] = (u32)v & FL_MASK;
        carry = v >> 28;
    }
    r.l[8] = (u32)(w.c[18] + carry);
    return r;
}

// interchange-domain image (x * 2^256, canonical) -> R280 form of x (x * 2^280 mod p, canonical, limbs)
SS_HD Fl fl_to_r280(const Fp &mont256) {
    Fp two24 = fp_zero(); two24.v[0] = 1u << 24;
    return fl_from_fp(fp_mul(mont256, fp_to_mont(two24)));
}

// Montgomery square of a NORMALISED value: 9 squares + 36 doubled cross products.
SS_HD Fl fl_sqr(const Fl &a) {
    u64 c[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) c[k] = 0;
    u32 d[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] = a.l[i] << 1;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        c[2 * i] += (u64)a.l[i] * a.l[i];
#pragma unroll
        for (int j = i + 1; j < 9; ++j) c[i + j] += (u64)a.l[i] * d[j];
    }
    return fl_mont_reduce(c);
}

// ---- the R280 DOMAIN: a value x kept as x * 2^280 mod p.  fl_mul_r280 of two such values is x y * 2^280 again, so a computation that
// never leaves the hot-loop form can run on the ten-step reduction throughout - no 4-bit tail, no shift across columns (25 of a
// product's ~200 instructions).  The Pedersen kernels do (ec252.h EcR280; their tables hold the points in this form): round 6.
// Either operand may be lazy as fl_mul's (limbs < 2^30 / < 2^29 + 2^25): the columns stay below 2^63, the result is normalised and
// < a b / 2^280 + p < 1.0000001 p for a, b < 4p.
SS_HD Fl fl_sqr_r280(const Fl &a) {
    u64 c[19];
#pragma unroll
    for (int k = 0; k < 19; ++k) c[k] = 0;
    u32 d[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] = a.l[i] << 1;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        c[2 * i] += (u64)a.l[i] * a.l[i];
#pragma unroll
        for (int j = i + 1; j < 9; ++j) c[i + j] += (u64)a.l[i] * d[j];
    }
    const u32 k17 = fl_opaque(17u << 24), k27 = fl_opaque(1u << 27), k1 = fl_opaque(1u);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const u32 m = (0u - (u32)c[i]) & FL_MASK;
        c[i + 1] += ((u64)m * k1 + c[i]) >> 28;
        c[i + 6] += (u64)m * k17;
        c[i + 8] += (u64)m * k27;
    }
    Fl r;
    u64 carry = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const u64 v = c[10 + j] + carry;
        r.l[j] = (u32)v & FL_MASK;
        carry = v >> 28;
    }
    r.l[8] = (u32)(c[18] + carry);
    return r;
}
// 1 in the R280 domain: 2^280 mod p
SS_HD Fl fl_one_r280() {
    Fl r;
    r.l[0] = 0x1u; r.l[1] = 0xffffffeu; r.l[2] = 0xfffffffu; r.l[3] = 0xfffffffu; r.l[4] = 0xfffffffu; r.l[5] = 0xfffffffu;
    r.l[6] = 0xffffffu; r.l[7] = 0xe000001u; r.l[8] = 0x7fffffdu;
    return r;
}

SS_HD Fl fl_zero() {
    Fl r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = 0;
    return r;
}

// ---- "safe" forms: inputs and outputs normalised (limbs < 2^28) with value < 2p.  Any
//      sequence of these is closed, at the price of one weak reduction per add/sub.
SS_HD Fl fn_add(const Fl &a, const Fl &b) { return fl_weak_reduce(fl_add(a, b)); }
SS_HD Fl fn_sub(const Fl &a, const Fl &b) { return fl_weak_reduce(fl_sub_c<2, 1>(a, b)); }
SS_HD Fl fn_mul(const Fl &a, const Fl &b) { return fl_mul(a, b); }       // < 4p^2/2^256 + p < 1.13 p
SS_HD Fl fn_sqr(const Fl &a) { return fl_sqr(a); }
SS_HD Fl fn_dbl(const Fl &a) { return fl_weak_reduce(fl_add(a, a)); }
// zero test for a normalised value < 2p: the only representatives of 0 are 0 and p
SS_HD bool fn_is_zero(const Fl &a) {
    u32 z = 0, e = (a.l[0] ^ FL_P0) | (a.l[6] ^ FL_P6) | (a.l[7] ^ FL_P7) | (a.l[8] ^ FL_P8);
#pragma unroll
    for (int i = 0; i < 9; ++i) z |= a.l[i];
#pragma unroll
    for (int i = 1; i < 6; ++i) e |= a.l[i];
    return z == 0 || e == 0;
}
SS_HD Fl fl_one() { return fl_from_fp(fp_one()); }
// a^(p-2), p - 2 = 2^251 + 2^196 + (2^192 - 1); input/output normalised.
// Top bits by square-and-multiply (bit 251, bit 196), the 192 trailing ones in 24 windows
// of 8 with a^(2^8 - 1): 258 squarings + 28 multiplications instead of 251 + 193.
SS_HD Fl fn_inv(const Fl &a) {
    const Fl x2 = fl_mul(fl_sqr(a), a);                       // a^3
    Fl t = fl_sqr(fl_sqr(x2));
    const Fl x4 = fl_mul(t, x2);                              // a^15
    t = x4;
    for (int i = 0; i < 4; ++i) t = fl_sqr(t);
    const Fl x8 = fl_mul(t, x4);                              // a^255
    Fl r = a;                                                 // bit 251
    for (int i = 0; i < 55; ++i) r = fl_sqr(r);
    r = fl_mul(r, a);                                         // bit 196
    for (int i = 0; i < 4; ++i) r = fl_sqr(r);                // down to bit 192
    for (int w = 0; w < 24; ++w) {
        for (int i = 0; i < 8; ++i) r = fl_sqr(r);
        r = fl_mul(r, x8);
    }
    return r;
}

}  // namespace ss
