// inv252.h — modular inversion mod p = 2^251 + 17*2^192 + 1 by Bernstein-Yang "safegcd"
// division steps (the variable-time, 30-bits-at-a-time formulation popularised by
// libsecp256k1's modinv32), host+device.
//
// Why: x^(p-2) is 251 dependent squarings — ~0.11 ms of pure latency for a wave on gfx950, and
// the top levels of every Pedersen Merkle tree (a handful of hashes each, one level after the
// other) pay it once per level.  safegcd needs ~18 outer iterations of
//   30 branch-free division steps on the low words  (~20 integer instructions each)
//   one 2x2-matrix update of (d, e) mod p and of (f, g)   (90 signed 32x32+64 multiply-adds)
// i.e. ~15 k instructions instead of ~45 k (+21 k half-rate slots) for the power.
//
// Representation: signed 30-bit limbs, 9 of them (the top limb carries the sign).
// Invariants: f, g integers with gcd(f, g) = gcd(p, x); d*x = f, e*x = g (mod p);
// d, e in (-2p, p).  When g = 0, f = +-gcd = +-1 and x^-1 = d * f.
#pragma once
#include "fp252.h"

namespace ss {

struct S30 { int32_t v[9]; };

static constexpr int32_t SG_M30 = (1 << 30) - 1;
// p in 30-bit limbs: 1, 0, 0, 0, 0, 0, 0x11000 (17 * 2^192 = 0x11000 * 2^180), 0, 0x800 (2^251 = 0x800 * 2^240)
SS_HD int32_t sg_modulus(int i) { return i == 0 ? 1 : i == 6 ? 0x11000 : i == 8 ? 0x800 : 0; }
// p^-1 mod 2^30 = 1 (p = 1 mod 2^192)

// 30 division steps on the low 32 bits of (f, g); returns the new eta and the transition matrix
// t = (u, v; q, r) scaled by 2^30:  (f, g) <- t (f, g) / 2^30.
SS_HD int32_t sg_divsteps_30(int32_t eta, uint32_t f, uint32_t g, int32_t t[4]) {
    uint32_t u = 1, v = 0, q = 0, r = 1;          // as uint32 with wrap-around = int32 arithmetic
#pragma unroll 1
    for (int i = 0; i < 30; ++i) {
        uint32_t c1 = (uint32_t)(eta >> 31);       // eta < 0 ? ~0 : 0
        const uint32_t c2 = 0u - (g & 1u);         // g odd ? ~0 : 0
        const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;     // conditionally negated f, u, v
        g += x & c2; q += y & c2; r += z & c2;
        c1 &= c2;                                   // eta < 0 and g odd: swap roles
        eta = (int32_t)(((uint32_t)eta ^ c1) - 1u);
        f += g & c1; u += q & c1; v += r & c1;
        g >>= 1; u <<= 1; v <<= 1;
    }
    t[0] = (int32_t)u; t[1] = (int32_t)v; t[2] = (int32_t)q; t[3] = (int32_t)r;
    return eta;
}

// The same 30 division steps, variable time: runs of even g are shifted out at once (count trailing zeros) and, while eta >= 0, up to
// six low bits of g are cancelled with one multiple of f (w = -g / f mod 2^k; for odd f, f (f^2 - 2) = -1/f mod 64).  The same
// steps in the same order, so the same matrix and eta as sg_divsteps_30 (tests/test_inv_safegcd.py holds one to the other) in a
// third of the instructions - for the ONE lane per hash that inverts at the top of a tree, where latency is all there is; lanes of
// a wave that run it diverge, so the batched inversions of large levels keep the branch-free form.
SS_HD int32_t sg_divsteps_30_var(int32_t eta, uint32_t f, uint32_t g, int32_t t[4]) {
    uint32_t u = 1, v = 0, q = 0, r = 1;
    int i = 30;
    for (;;) {
        const int zeros = __builtin_ctz(g | (0xffffffffu << i));      // at most i
        g >>= zeros; u <<= zeros; v <<= zeros;
        eta -= zeros; i -= zeros;
        if (i == 0) break;
        if (eta < 0) {                                                // g odd, eta < 0: (f, g) <- (g, -f); with the step's own
            uint32_t tmp;                                             // decrement (counted with the zeros) eta -> -eta - 2 as above
            eta = -eta - 1;
            tmp = f; f = g; g = 0u - tmp;
            tmp = u; u = q; q = 0u - tmp;
            tmp = v; v = r; r = 0u - tmp;
        }
        // eta >= 0: cancel low bits of g - no more than i (done by then) and no more than eta + 1 (the sign flips there)
        const int limit = (eta + 1) > i ? i : (eta + 1);
        const uint32_t m = (0xffffffffu >> (32 - limit)) & 63u;
        const uint32_t w = (g * f * (f * f - 2u)) & m;
        g += f * w; q += u * w; r += v * w;
    }
    t[0] = (int32_t)u; t[1] = (int32_t)v; t[2] = (int32_t)q; t[3] = (int32_t)r;
    return eta;
}

// (d, e) <- t (d, e) / 2^30 mod p, keeping d, e in (-2p, p)
SS_HD void sg_update_de(S30 &d, S30 &e, const int32_t t[4]) {
    const int64_t u = t[0], v = t[1], q = t[2], r = t[3];
    const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
    int32_t md = (t[0] & sd) + (t[1] & se), me = (t[2] & sd) + (t[3] & se);
    int64_t cd = u * d.v[0] + v * e.v[0], ce = q * d.v[0] + r * e.v[0];
    // multiples of p that clear the low 30 bits (p^-1 mod 2^30 = 1)
    md -= (int32_t)(((uint32_t)cd + (uint32_t)md) & (uint32_t)SG_M30);
    me -= (int32_t)(((uint32_t)ce + (uint32_t)me) & (uint32_t)SG_M30);
    cd += (int64_t)sg_modulus(0) * md; ce += (int64_t)sg_modulus(0) * me;
    cd >>= 30; ce >>= 30;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        cd += u * d.v[i] + v * e.v[i]; ce += q * d.v[i] + r * e.v[i];
        if (sg_modulus(i)) { cd += (int64_t)sg_modulus(i) * md; ce += (int64_t)sg_modulus(i) * me; }
        d.v[i - 1] = (int32_t)cd & SG_M30; cd >>= 30;
        e.v[i - 1] = (int32_t)ce & SG_M30; ce >>= 30;
    }
    d.v[8] = (int32_t)cd; e.v[8] = (int32_t)ce;
}

// (f, g) <- t (f, g) / 2^30 (exact)
SS_HD void sg_update_fg(S30 &f, S30 &g, const int32_t t[4]) {
    const int64_t u = t[0], v = t[1], q = t[2], r = t[3];
    int64_t cf = u * f.v[0] + v * g.v[0], cg = q * f.v[0] + r * g.v[0];
    cf >>= 30; cg >>= 30;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        cf += u * f.v[i] + v * g.v[i]; cg += q * f.v[i] + r * g.v[i];
        f.v[i - 1] = (int32_t)cf & SG_M30; cf >>= 30;
        g.v[i - 1] = (int32_t)cg & SG_M30; cg >>= 30;
    }
    f.v[8] = (int32_t)cf; g.v[8] = (int32_t)cg;
}

// the end of an inversion: d in (-2p, p) as signed limbs, f = +-1 (fneg: the inverse is -d) -> the canonical 8 x u32
SS_HD Fp sg_finish(const S30 &d, bool fneg) {
    // d in (-2p, p) as signed limbs -> 9 x 32-bit two's complement
    uint32_t w[9];
    {
        int64_t acc = 0;
        // gather 30-bit limbs into 32-bit words: word k holds bits [32k, 32k+32)
        // value = sum d_i 2^(30 i); process by accumulating a running 64-bit window
        int bitpos = 0;     // number of valid bits in acc beyond those already emitted
        int wi = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            acc += (int64_t)d.v[i] * ((int64_t)1 << bitpos);   // limbs 0..7 are in [0, 2^30); limb 8 is signed (a product: shifting a negative value left is undefined before C++20)
            bitpos += 30;
            if (bitpos >= 32 && wi < 9) {
                w[wi++] = (uint32_t)acc;
                acc >>= 32;                          // arithmetic: keeps the sign of the top limb
                bitpos -= 32;
            }
        }
        // remaining words: sign extension
        for (; wi < 9; ++wi) { w[wi] = (uint32_t)acc; acc >>= 32; }
    }
    if (fneg) {                                         // two's complement negate (288 bits)
        uint64_t c = 1;
#pragma unroll
        for (int i = 0; i < 9; ++i) { c += (uint32_t)~w[i]; w[i] = (uint32_t)c; c >>= 32; }
    }
    // now in (-2p, 2p): add p while negative (twice), subtract p while >= p (twice)
    const uint32_t pw[9] = {SS_P0, 0, 0, 0, 0, 0, SS_P6, SS_P7, 0};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint32_t neg = 0u - (w[8] >> 31);
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) { c += (uint64_t)w[i] + (pw[i] & neg); w[i] = (uint32_t)c; c >>= 32; }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        uint32_t tmp[9];
        int64_t b = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) { b += (int64_t)w[i] - (int64_t)pw[i]; tmp[i] = (uint32_t)b; b >>= 32; }
        const bool ge = (tmp[8] >> 31) == 0;            // w - p >= 0
        if (ge) {
#pragma unroll
            for (int i = 0; i < 9; ++i) w[i] = tmp[i];
        }
    }
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = w[i];
    return r;
}

// x^-1 mod p for a canonical integer x < p given as 8 x u32 (NOT Montgomery); 0 -> 0.  VAR: the variable-time division steps.
template <bool VAR = false>
SS_HD Fp sg_inverse_canonical(const Fp &x) {
    S30 d, e, f, g;
#pragma unroll
    for (int i = 0; i < 9; ++i) { d.v[i] = 0; e.v[i] = 0; f.v[i] = sg_modulus(i); }
    e.v[0] = 1;
    // 8 x 32 -> 9 x 30
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bit = 30 * i, w = bit >> 5, s = bit & 31;
        uint64_t two = x.v[w];
        if (w + 1 < 8) two |= (uint64_t)x.v[w + 1] << 32;
        g.v[i] = (int32_t)((uint32_t)(two >> s) & (uint32_t)SG_M30);
    }
    int32_t eta = -1;
#pragma unroll 1
    for (int it = 0; it < 32; ++it) {                   // 741 division steps suffice for 256-bit inputs: 25 rounds
        int32_t t[4];
        const uint32_t f0 = (uint32_t)f.v[0] | ((uint32_t)f.v[1] << 30), g0 = (uint32_t)g.v[0] | ((uint32_t)g.v[1] << 30);
        eta = VAR ? sg_divsteps_30_var(eta, f0, g0, t) : sg_divsteps_30(eta, f0, g0, t);
        sg_update_de(d, e, t);
        sg_update_fg(f, g, t);
        int32_t nz = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) nz |= g.v[i];
        if (nz == 0) break;
    }
    return sg_finish(d, f.v[8] < 0);                   // f = -1: the inverse is -d
}

// Montgomery in, Montgomery out: a = x R  ->  x^-1 R   (R = 2^256).  sg gives (x R)^-1 = x^-1 R^-1; times R^2 by
// one Montgomery multiplication with R^3.
template <bool VAR = false>
SS_HD Fp fp_inv_safegcd(const Fp &a_mont) {
    Fp r3;
    r3.v[0] = 0x406df18eu; r3.v[1] = 0xcc7177d1u; r3.v[2] = 0x77ffcc06u; r3.v[3] = 0x75457066u;
    r3.v[4] = 0x36300018u; r3.v[5] = 0xf47d84f8u; r3.v[6] = 0x873c0a6du; r3.v[7] = 0x038e5f79u;
    return fp_mul(sg_inverse_canonical<VAR>(a_mont), r3);
}

}  // namespace ss
