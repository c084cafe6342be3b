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
        Fp r1 = fp_inv(a), r2 = fp_inv_safegcd(a);
        if (!fp_eq(r1, r2)) { if (bad < 5) { printf("mismatch at %d\n", it); for (int i=0;i<8;++i) printf("%08x %08x %08x\n", a.v[i], r1.v[i], r2.v[i]); } ++bad; }
    }
    printf("bad = %d\n", bad);
    return bad != 0;
}
