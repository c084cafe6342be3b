// fp252.h — 252-bit StarkWare prime field for CDNA4 (gfx950), host+device.
//
// p = 2^251 + 17*2^192 + 1 (reference cli/src/main.rs:25-26).  Elements are
// kept in Montgomery form, R = 2^256, as 8 x u32 little-endian limbs — the same
// 32-byte image as the reference's 4 x u64 `Fp` (crypto/src/utils.rs:8-22), so
// a global load of two dwordx4 IS the conversion.
//
// Design notes (CDNA4):
//  * There is no 64x64 multiplier; the product is 64 v_mad_u64_u32
//    (32x32+64 -> 64).  gfx950 needs 2 wait states between a VALU that writes
//    VCC and a VALU that reads it as carry-in, so long v_addc chains are slow;
//    the row-wise form `t + a*b + carry` (never overflows 64 bits) lets hipcc
//    use v_lshl_add_u64 64-bit adds instead of carry chains.
//  * p = 1 (mod 2^64), so -p^-1 = -1 and every Montgomery reduction step is
//    m = -t0; t += m + m*(2^59+17)*2^192: shifts and adds, no multiplies.
//  * Values are kept fully reduced (< p) between operations.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)          /* hipcc: device + host; g++ (tests/cpp, the host library): host only */
#define SS_HD __host__ __device__ __forceinline__
#else
#define SS_HD inline
#endif

namespace ss {

typedef uint32_t u32;
typedef uint64_t u64;

struct Fp {
    u32 v[8];
};

// p, 32-bit limbs
#define SS_P0 0x00000001u
#define SS_P6 0x00000011u
#define SS_P7 0x08000000u

SS_HD Fp fp_zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = 0;
    return r;
}
// R mod p = 2^256 mod p = -(32*17*2^192 + 32) mod p  (Montgomery one)
SS_HD Fp fp_one() {
    Fp r;
    r.v[0] = 0xffffffe1u; r.v[1] = 0xffffffffu; r.v[2] = 0xffffffffu; r.v[3] = 0xffffffffu;
    r.v[4] = 0xffffffffu; r.v[5] = 0xffffffffu; r.v[6] = 0xfffffdf0u; r.v[7] = 0x07ffffffu;
    return r;
}
SS_HD bool fp_is_zero(const Fp &a) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= a.v[i];
    return o == 0;
}
SS_HD bool fp_eq(const Fp &a, const Fp &b) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= a.v[i] ^ b.v[i];
    return o == 0;
}

// r = a - p if a >= p else a, for a < 2p (a may use all 256 bits)
SS_HD Fp fp_reduce_once(const Fp &a) {
    // d = a - p, borrow out tells a < p
    u32 d[8];
    u64 br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32 pi = i == 0 ? SS_P0 : i == 6 ? SS_P6 : i == 7 ? SS_P7 : 0u;
        u64 x = (u64)a.v[i] - pi - br;
        d[i] = (u32)x;
        br = (x >> 32) & 1;
    }
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = br ? a.v[i] : d[i];
    return r;
}

SS_HD Fp fp_add(const Fp &a, const Fp &b) {
    Fp s;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        c += (u64)a.v[i] + b.v[i];
        s.v[i] = (u32)c;
        c >>= 32;
    }
    return fp_reduce_once(s);  // a + b < 2p < 2^253: no carry out
}

SS_HD Fp fp_sub(const Fp &a, const Fp &b) {
    u32 d[8];
    u64 br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u64 x = (u64)a.v[i] - b.v[i] - br;
        d[i] = (u32)x;
        br = (x >> 32) & 1;
    }
    // add p back when a < b
    Fp r;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32 pi = i == 0 ? SS_P0 : i == 6 ? SS_P6 : i == 7 ? SS_P7 : 0u;
        c += (u64)d[i] + (br ? pi : 0u);
        r.v[i] = (u32)c;
        c >>= 32;
    }
    return r;
}

SS_HD Fp fp_neg(const Fp &a) { return fp_sub(fp_zero(), a); }

SS_HD Fp fp_dbl(const Fp &a) { return fp_add(a, a); }

// 512-bit product, row-wise: t[i+j] + a[j]*b[i] + carry never overflows u64.
SS_HD void fp_mul_wide(const Fp &a, const Fp &b, u32 t[16]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32 carry = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            u64 p = (u64)a.v[j] * b.v[i] + t[i + j] + carry;
            t[i + j] = (u32)p;
            carry = (u32)(p >> 32);
        }
        t[i + 8] = carry;
    }
}

// Montgomery reduction of a 512-bit t < p * 2^256: returns t * 2^-256 mod p.
// Four 64-bit steps; with p = 1 + K*2^192, K = 2^59 + 17 and -p^-1 = -1:
//   m = -w[i];  w[i] + m = 2^64 * (w[i] != 0);  m*K added at word i+3.
SS_HD Fp fp_mont_reduce(const u32 t[16]) {
    u64 w[9];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = (u64)t[2 * i] | ((u64)t[2 * i + 1] << 32);
    w[8] = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u64 m = 0 - w[i];
        u64 carry = w[i] != 0;  // from w[i] + m
        // m * K = (m << 59) + 17 m, a 124-bit value (hi:lo)
        u64 k_lo = m << 59, k_hi = m >> 5;
        u64 s_lo = (m << 4) + m, s_hi = (m >> 60) + (s_lo < m);
        u64 lo = k_lo + s_lo;
        u64 hi = k_hi + s_hi + (lo < k_lo);
        // propagate `carry` through words i+1, i+2, then add (hi:lo) at i+3, i+4
        u64 x = w[i + 1] + carry; carry = x < carry; w[i + 1] = x;
        x = w[i + 2] + carry; carry = x < carry; w[i + 2] = x;
        x = w[i + 3] + carry; u64 c2 = x < carry; x += lo; c2 += x < lo; w[i + 3] = x;
        x = w[i + 4] + c2; u64 c3 = x < c2; x += hi; c3 += x < hi; w[i + 4] = x;
#pragma unroll
        for (int k = i + 5; k < 9; ++k) { x = w[k] + c3; c3 = x < c3; w[k] = x; }
    }
    Fp r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.v[2 * i] = (u32)w[4 + i]; r.v[2 * i + 1] = (u32)(w[4 + i] >> 32); }
    // t < p*2^256 => result < 2p; w[8] is zero then
    return fp_reduce_once(r);
}

SS_HD Fp fp_mul(const Fp &a, const Fp &b) {
    u32 t[16];
    fp_mul_wide(a, b, t);
    return fp_mont_reduce(t);
}
SS_HD Fp fp_sqr(const Fp &a) { return fp_mul(a, a); }

// x * 2^-k mod p for 0 < k <= 32 (a partial Montgomery step: p = 1 mod 2^32,
// so m = -x mod 2^k and (x + m p) >> k is exact).  Used for the 1/n of inverse NTTs.
SS_HD Fp fp_div_pow2(const Fp &a, unsigned k) {
    u32 mask = k == 32 ? 0xffffffffu : ((1u << k) - 1u);
    u32 m = (0u - a.v[0]) & mask;
    // x + m*p, p = 1 + 17*2^192 + 2^251: 257-bit worst case -> 9 limbs
    u32 s[9];
    u64 c = (u64)a.v[0] + m;
    s[0] = (u32)c; c >>= 32;
#pragma unroll
    for (int i = 1; i < 6; ++i) { c += a.v[i]; s[i] = (u32)c; c >>= 32; }
    c += (u64)a.v[6] + (u64)m * 17; s[6] = (u32)c; c >>= 32;
    c += (u64)a.v[7] + ((u64)m << 27); s[7] = (u32)c; c >>= 32;
    s[8] = (u32)c;
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = k == 32 ? s[i + 1] : ((s[i] >> k) | (s[i + 1] << (32 - k)));
    return r;  // < p when a < p
}

SS_HD Fp fp_pow_u64(Fp a, u64 e) {
    Fp r = fp_one();
    while (e) {
        if (e & 1) r = fp_mul(r, a);
        a = fp_sqr(a);
        e >>= 1;
    }
    return r;
}

// a^(p-2); p-2 = 2^251 + 17*2^192 - 1
SS_HD Fp fp_inv(const Fp &a) {
    // exponent bits: bit 251; bits 196,192 from 17*2^192 minus 1 -> (17*2^192 - 1) = 2^196 + 2^192 - 1
    //  = 2^196 + (2^192 - 1): bit 196 set, bits 0..191 all set.
    Fp r = a;  // bit 251
    for (int i = 250; i >= 0; --i) {
        r = fp_sqr(r);
        bool bit = (i == 196) || (i < 192);
        if (bit) r = fp_mul(r, a);
    }
    return r;
}

// canonical integer (< p) <-> Montgomery
SS_HD Fp fp_from_mont(const Fp &a) {
    u32 t[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) { t[i] = a.v[i]; t[i + 8] = 0; }
    return fp_mont_reduce(t);
}
SS_HD Fp fp_r2() {
    // R^2 mod p = 2^512 mod p
    Fp r;
    r.v[0] = 0x7e000401u; r.v[1] = 0xfffffd73u; r.v[2] = 0x330fffffu; r.v[3] = 0x00000001u;
    r.v[4] = 0xff6f8000u; r.v[5] = 0xffffffffu; r.v[6] = 0x5e008810u; r.v[7] = 0x07ffd4abu;
    return r;
}
SS_HD Fp fp_to_mont(const Fp &a) { return fp_mul(a, fp_r2()); }
SS_HD Fp fp_from_u64(u64 x) {
    Fp a = fp_zero();
    a.v[0] = (u32)x; a.v[1] = (u32)(x >> 32);
    return fp_to_mont(a);
}

}  // namespace ss
