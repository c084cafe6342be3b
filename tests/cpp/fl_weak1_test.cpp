// Host check of fl_weak_reduce1 (sandstorm_amd/csrc/fl252.h: the one-chain weak reduction a transform pass stores through)
// at the bounds it is specified for: limbs 0..7 up to 15 * 2^28 - 33, any top limb (value < 2^256): the result is the same
// residue, normalised, below 2^252 + 2^229.
#include <cstdint>
#include <cstdio>
#include "../../sandstorm_amd/csrc/fl252.h"

using namespace ss;

static uint64_t st = 0x9E3779B97F4A7C15ull;
static uint64_t next() { uint64_t z = (st += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

int main() {
    int bad = 0;
    const u32 LMAX = 15u * (1u << 28) - 33u;
    for (int round = 0; round < 200000; ++round) {
        Fl a;
        for (int i = 0; i < 8; ++i) {
            const int kind = (int)(next() % 6);
            a.l[i] = kind == 0 ? LMAX : kind == 1 ? 0u : kind == 2 ? (u32)(next() % 3) : kind == 3 ? LMAX - (u32)(next() % 3) : (u32)(next() % ((uint64_t)LMAX + 1));
        }
        const int kt = (int)(next() % 6);
        // the value must stay below 2^256: top limb + the carries of the lower limbs (< 16) below 2^32
        a.l[8] = kt == 0 ? 0xffffffe0u : kt == 1 ? 0u : kt == 2 ? (u32)(next() % (1u << 27)) : kt == 3 ? (1u << 27) + (u32)(next() % 3) : kt == 4 ? ((u32)(next() % 31) << 27) + (u32)(next() % 5) - 2u * (next() % 2) : (u32)(next() % 0xffffffe0ull);
        if (a.l[8] > 0xffffffe0u) a.l[8] = 0xffffffe0u;
        const Fl r = fl_weak_reduce1(a);
        for (int i = 0; i < 8; ++i) if (r.l[i] > FL_MASK) { ++bad; break; }
        if (r.l[8] >= (1u << 28) + (1u << 5)) ++bad;                                  // < 2^252 + 2^229
        if (!fp_eq(fl_to_fp(r), fl_to_fp(a))) ++bad;                                  // same residue (fl_to_fp: any value < 32 p)
    }
    printf(bad ? "FL_WEAK1_FAIL %d\n" : "FL_WEAK1_OK\n", bad);
    return bad != 0;
}
