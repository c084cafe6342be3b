// Host check of the fused dot product (sandstorm_amd/csrc/fl252.h FlWide: what the DEEP kernel and the generated
// constraint kernels accumulate with) against sums of single Montgomery products, at the limb bounds it is specified
// for: FL_WIDE_MAX_TERMS terms, multiplicands weakly reduced (< 2p, limbs < 2^28) or at the largest normalised value,
// multipliers in R280 form.
#include <cstdint>
#include <cstdio>
#include "../../sandstorm_amd/csrc/fl252.h"

using namespace ss;

static uint64_t st = 0x9E3779B97F4A7C15ull;
static uint64_t next() { uint64_t z = (st += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static Fp rand_fp() {
    Fp a;
    for (int i = 0; i < 8; ++i) a.v[i] = (u32)next();
    a.v[7] &= 0x07ffffffu;                    // < 2^251 < p
    return a;
}

int main() {
    int bad = 0;
    for (int round = 0; round < 2000; ++round) {
        const int terms = 1 + (int)(next() % FL_WIDE_MAX_TERMS);
        FlWide w;
        fl_wide_zero(w);
        Fp want = fp_zero();
        for (int k = 0; k < (round % 7 == 0 ? FL_WIDE_MAX_TERMS : terms); ++k) {
            const Fp a = rand_fp(), t = rand_fp();
            Fl al = fl_from_fp(a);
            if (round % 3 == 1) al = fl_weak_reduce(fl_add(al, fl_from_fp(rand_fp())));     // a weakly reduced sum: < 2p
            if (round % 5 == 2) for (int i = 0; i < 9; ++i) al.l[i] = FL_MASK;               // the largest normalised limbs
            fl_wide_mad(w, al, fl_to_r280(t));
            want = fp_add(want, fp_mul(fl_to_fp(al), t));
        }
        const Fl r = fl_wide_reduce(w);
        for (int i = 0; i < 8; ++i) if (r.l[i] > FL_MASK) { ++bad; break; }                  // normalised
        if (!fp_eq(fl_to_fp(r), want)) ++bad;
    }
    printf(bad ? "FL_WIDE_FAIL %d\n" : "FL_WIDE_OK\n", bad);
    return bad != 0;
}
