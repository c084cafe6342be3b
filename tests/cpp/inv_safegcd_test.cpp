#include <cstdio>
#include <cstdlib>
#include <random>
#include "inv252.h"
using namespace ss;
int main() {
    std::mt19937_64 rng(42);
    int bad = 0;
    for (int it = 0; it < 20000; ++it) {
        Fp a;
        for (int i = 0; i < 8; ++i) a.v[i] = (u32)rng();
        a.v[7] &= 0x07ffffffu;                      // < 2^251 < p
        if (it == 0) a = fp_zero();
        if (it == 1) a = fp_one();
        if (it == 2) { a = fp_zero(); a.v[0] = 1; }
        if (it == 3) { a = fp_zero(); a.v[0] = 0; a.v[6] = SS_P6; a.v[7] = SS_P7; }   // p - 1
        Fp r1 = fp_inv(a), r2 = fp_inv_safegcd(a), r3 = fp_inv_safegcd<true>(a);
        if (!fp_eq(r2, r3)) { if (bad < 5) printf("variable-time inversion differs at %d\n", it); ++bad; }
        if (!fp_eq(r1, r2)) { if (bad < 5) { printf("mismatch at %d\n", it); for (int i=0;i<8;++i) printf("%08x %08x %08x\n", a.v[i], r1.v[i], r2.v[i]); } ++bad; }
    }
    // the variable-time division steps are the same steps: same matrix, same eta, for any (eta, f odd, g)
    for (int it = 0; it < 2000000; ++it) {
        const uint32_t f = (uint32_t)rng() | 1u, g = it < 64 ? (uint32_t)it : it < 96 ? (1u << (it - 64)) : (uint32_t)rng();
        const int32_t eta = it % 7 == 0 ? -1 : (int32_t)(rng() % 1200) - 600;
        int32_t ta[4], tb[4];
        const int32_t ea = sg_divsteps_30(eta, f, g, ta), eb = sg_divsteps_30_var(eta, f, g, tb);
        if (ea != eb || ta[0] != tb[0] || ta[1] != tb[1] || ta[2] != tb[2] || ta[3] != tb[3]) { if (bad < 5) printf("division steps differ at %d\n", it); ++bad; }
    }
    printf("bad = %d\n", bad);
    return bad != 0;
}
