// Host check of csrc/fp252_host.h (the 4 x 64-bit Montgomery arithmetic the host's per-proof work runs in: the coin's Pedersen chain,
// DEEP's polynomials) against csrc/fp252.h (the 8 x 32-bit arithmetic everything else is held to): the same images, bit for bit.
#include <cstdint>
#include <cstdio>
#include <random>

#include "../../sandstorm_amd/csrc/fp252_host.h"

using namespace ss;

static Fp canonical(std::mt19937_64 &rng, int kind) {
    Fp a;
    for (int i = 0; i < 8; ++i) a.v[i] = (u32)rng();
    a.v[7] &= 0x07ffffffu;                                   // < 2^251 < p
    if (kind == 1) a = fp_zero();
    if (kind == 2) a = fp_one();
    if (kind == 3) { a = fp_zero(); a.v[6] = SS_P6; a.v[7] = SS_P7; }                       // p - 1
    if (kind == 4) { for (int i = 0; i < 6; ++i) a.v[i] = 0xffffffffu; a.v[6] = SS_P6 - 1; a.v[7] = SS_P7; }   // just below 17 * 2^192 + 2^251
    if (kind == 5) { a = fp_zero(); a.v[0] = 1; }
    return a;
}

int main() {
    std::mt19937_64 rng(2026);
    int bad = 0;
    for (int it = 0; it < 200000; ++it) {
        const Fp a = canonical(rng, it < 36 ? it % 6 : 0), b = canonical(rng, it < 36 ? it / 6 : 0);
        const H4 ha = h4_from_fp(a), hb = h4_from_fp(b);
        if (!fp_eq(h4_to_fp(ha), a)) ++bad;
        if (!fp_eq(h4_to_fp(h4_mul(ha, hb)), fp_mul(a, b))) { if (bad < 5) printf("mul differs at %d\n", it); ++bad; }
        if (!fp_eq(h4_to_fp(h4_add(ha, hb)), fp_add(a, b))) { if (bad < 5) printf("add differs at %d\n", it); ++bad; }
        if (!fp_eq(h4_to_fp(h4_sub(ha, hb)), fp_sub(a, b))) { if (bad < 5) printf("sub differs at %d\n", it); ++bad; }
        if (!fp_eq(fph_mul(a, b), fp_mul(a, b)) || !fp_eq(fph_sqr(a), fp_sqr(a))) ++bad;
        if (it % 64 == 0) {
            const uint64_t e = it < 640 ? (uint64_t)it / 64 : rng();
            if (!fp_eq(fph_pow_u64(a, e), fp_pow_u64(a, e))) { if (bad < 5) printf("pow differs at %d\n", it); ++bad; }
        }
    }
    printf("bad = %d\n", bad);
    return bad != 0;
}
