// fp252_host.h — Fp252 products for HOST code that runs per proof (the coin's Pedersen chain, the out-of-domain step's level
// multipliers).  fp252.h's 8 x 32-bit limbs are shaped for the GPU's 32-bit multiplier and run at ~45 ns per product on a CPU
// core; the same Montgomery product (R = 2^256: the SAME images, limb pairs read as one word) in 4 x 64-bit limbs with 128-bit
// partial products takes ~5 ns.  Host only (unsigned __int128); fp252.h stays the arithmetic the device code and its host
// checks (tests/cpp, tests/hipemu) share.
#pragma once
#include "fp252.h"

namespace ss {

struct H4 { uint64_t v[4]; };
typedef unsigned __int128 u128;
constexpr uint64_t H4_P3 = 0x0800000000000011ull;                    // p = 2^251 + 17 * 2^192 + 1 = {1, 0, 0, H4_P3}
inline H4 h4_from_fp(const Fp &a) { H4 r; for (int i = 0; i < 4; ++i) r.v[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32); return r; }
inline Fp h4_to_fp(const H4 &a) { Fp r; for (int i = 0; i < 4; ++i) { r.v[2 * i] = (u32)a.v[i]; r.v[2 * i + 1] = (u32)(a.v[i] >> 32); } return r; }
inline bool h4_is_zero(const H4 &a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
inline bool h4_geq_p(const H4 &a) {
    if (a.v[3] != H4_P3) return a.v[3] > H4_P3;
    return (a.v[2] | a.v[1]) != 0 || a.v[0] >= 1;
}
inline H4 h4_sub_p(const H4 &a) {
    H4 r; u128 d = (u128)a.v[0] - 1; r.v[0] = (uint64_t)d; uint64_t br = (uint64_t)(d >> 64) & 1;
    d = (u128)a.v[1] - br; r.v[1] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1;
    d = (u128)a.v[2] - br; r.v[2] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1;
    r.v[3] = a.v[3] - H4_P3 - br;
    return r;
}
inline H4 h4_add(const H4 &a, const H4 &b) {                          // canonical in, canonical out (a + b < 2p < 2^256)
    H4 r; u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a.v[i] + b.v[i]; r.v[i] = (uint64_t)c; c >>= 64; }
    return h4_geq_p(r) ? h4_sub_p(r) : r;
}
inline H4 h4_sub(const H4 &a, const H4 &b) {
    H4 r; uint64_t br = 0;
    for (int i = 0; i < 4; ++i) { const u128 d = (u128)a.v[i] - b.v[i] - br; r.v[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
    if (br) {                                                         // + p
        u128 c = (u128)r.v[0] + 1; r.v[0] = (uint64_t)c; c >>= 64;
        c += r.v[1]; r.v[1] = (uint64_t)c; c >>= 64;
        c += r.v[2]; r.v[2] = (uint64_t)c; c >>= 64;
        r.v[3] += H4_P3 + (uint64_t)c;
    }
    return r;
}
// Montgomery product, operand scanning with the reduction interleaved.  p = 1 mod 2^64, so the quotient digit is m = -t0 and
// t + m p = t + m + m H4_P3 2^192
inline H4 h4_mul(const H4 &a, const H4 &b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)a.v[i] * b.v[j] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        const uint64_t m = 0 - t[0];
        c = (u128)t[0] + m;                                           // low word becomes zero
        c >>= 64;
        c += t[1]; t[0] = (uint64_t)c; c >>= 64;
        c += t[2]; t[1] = (uint64_t)c; c >>= 64;
        c += (u128)m * H4_P3 + t[3]; t[2] = (uint64_t)c; c >>= 64;
        c += t[4]; t[3] = (uint64_t)c; c >>= 64;
        t[4] = t[5] + (uint64_t)c;
    }
    H4 r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || h4_geq_p(r)) r = h4_sub_p(r);
    return r;
}
inline H4 h4_sqr(const H4 &a) { return h4_mul(a, a); }

// the same on Fp images
inline Fp fph_mul(const Fp &a, const Fp &b) { return h4_to_fp(h4_mul(h4_from_fp(a), h4_from_fp(b))); }
inline Fp fph_sqr(const Fp &a) { const H4 x = h4_from_fp(a); return h4_to_fp(h4_mul(x, x)); }
inline Fp fph_pow_u64(const Fp &a, uint64_t e) {
    H4 x = h4_from_fp(a), r = h4_from_fp(fp_one());
    while (e) { if (e & 1) r = h4_mul(r, x); x = h4_mul(x, x); e >>= 1; }
    return h4_to_fp(r);
}

}  // namespace ss
