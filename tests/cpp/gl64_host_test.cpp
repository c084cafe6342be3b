// Host check of sandstorm_amd/csrc/gl64.h (tests/test_gl64_host.py): the device's field arithmetic for p = 2^64 - 2^32 + 1 - products in
// 32-bit halves, lazy sums, wide accumulators, Fq3 - against 128-bit integer arithmetic, on edge words and random ones; and a
// decimation-in-time network run with the lazy butterflies of csrc/goldilocks.hip against the same network on canonical values.
#include "gl64.h"
#include <cstdio>
#include <cstdlib>
#include <random>
using namespace ss;
typedef unsigned __int128 u128;
static uint64_t modp(u128 v) { return (uint64_t)(v % GL_P); }
static uint64_t mulp(uint64_t a, uint64_t b) { return modp((u128)(a % GL_P) * (b % GL_P)); }
int main() {
    std::mt19937_64 rng(12345);
    const uint64_t edge[] = {0, 1, 2, 0xFFFFFFFFull, 0x100000000ull, 0x100000001ull, GL_P - 2, GL_P - 1, GL_P, GL_P + 1, 0xFFFFFFFFFFFFFFFEull, 0xFFFFFFFFFFFFFFFFull,
                             0xFFFFFFFF00000000ull, 0x8000000000000000ull, 0x7FFFFFFFFFFFFFFFull, 0xFFFFFFFEFFFFFFFFull};
    const int NE = sizeof(edge) / sizeof(edge[0]);
    auto pick = [&](int i) -> uint64_t { return i < NE ? edge[i] : ((i & 7) == 0 ? (rng() | 0xFFFFFFFF00000000ull) : rng()); };
    long bad = 0;
    for (int i = 0; i < NE + 3000; ++i) for (int j = 0; j < NE + 300; ++j) {
        const uint64_t a = pick(i), b = pick(j);
        uint64_t lo, hi; gl_mul_wide(a, b, lo, hi);
        const u128 pr = (u128)a * b;
        if (lo != (uint64_t)pr || hi != (uint64_t)(pr >> 64)) { ++bad; }
        if (gl_mul(a, b) != mulp(a, b)) ++bad;
        if (gl_mul_lazy(a, b) % GL_P != mulp(a, b)) ++bad;
        if (gl_reduce128(a, b) != modp(((u128)b << 64) | a)) ++bad;
        const uint64_t bc = b % GL_P, ac = a % GL_P;
        if (gl_add_lazy(a, bc) % GL_P != modp((u128)ac + bc)) ++bad;
        if (gl_sub_lazy(a, bc) % GL_P != modp((u128)ac + GL_P - bc)) ++bad;
        if (gl_add(ac, bc) != modp((u128)ac + bc)) ++bad;
        if (gl_sub(ac, bc) != modp((u128)ac + GL_P - bc)) ++bad;
    }
    // wide sums: long runs of extreme and random words
    for (int run = 0; run < 400; ++run) {
        GlWide w = glw_zero(); uint64_t ref = 0;
        const int n = run < 8 ? 100000 : 1 + (int)(rng() % 600);
        for (int k = 0; k < n; ++k) {
            uint64_t a = run < 4 ? 0xFFFFFFFFFFFFFFFFull : pick((int)(rng() % (NE + 40))), b = run < 4 ? 0xFFFFFFFFFFFFFFFFull - (run & 1) : pick((int)(rng() % (NE + 40)));
            if (rng() % 5 == 0) { glw_add(w, a); ref = modp((u128)ref + a % GL_P); }
            else { glw_mac(w, a, b); ref = modp((u128)ref + mulp(a, b)); }
        }
        if (glw_reduce(w) != ref) ++bad;
    }
    // Fq3
    for (int t = 0; t < 20000; ++t) {
        Gl3 a, b; for (int c = 0; c < 3; ++c) { a.c[c] = pick((int)(rng() % (NE + 20))) % GL_P; b.c[c] = pick((int)(rng() % (NE + 20))) % GL_P; }
        uint64_t d[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) d[i + j] = modp((u128)d[i + j] + mulp(a.c[i], b.c[j]));
        const uint64_t e0 = modp((u128)d[0] + 2 * (u128)d[3]), e1 = modp((u128)d[1] + 2 * (u128)d[4]), e2 = d[2];
        const Gl3 r = gl3_mul(a, b);
        if (r.c[0] != e0 || r.c[1] != e1 || r.c[2] != e2) ++bad;
        Gl3Wide w = gl3w_zero(); gl3w_mac(w, a, gl3_rhs(b)); gl3w_mac_base(w, b, a.c[0]); gl3w_add(w, a);
        const Gl3 s = gl3w_reduce(w);
        if (s.c[0] != modp((u128)e0 + mulp(b.c[0], a.c[0]) + a.c[0]) || s.c[1] != modp((u128)e1 + mulp(b.c[1], a.c[0]) + a.c[1]) || s.c[2] != modp((u128)e2 + mulp(b.c[2], a.c[0]) + a.c[2])) ++bad;
    }
    // the transform's butterflies: lazy sums and differences, canonical once at the end (goldilocks.hip gl_group, DIT) and the
    // decimation-in-frequency form (lazy difference into the product); any network of butterflies does for the comparison
    for (int run = 0; run < 40; ++run) {
        const int LOG = 10, N = 1 << LOG;
        static uint64_t x[1 << 10], y[1 << 10], u[1 << 10];
        for (int i = 0; i < N; ++i) x[i] = y[i] = u[i] = run == 0 ? GL_P - 1 : run == 1 ? (i & 1 ? GL_P - 1 : 0) : run == 2 ? (i & 1 ? 1 : GL_P - 1) : pick((int)(rng() % (NE + 60))) % GL_P;
        uint64_t wroot = 1; { uint64_t b = 7, e = (GL_P - 1) >> LOG; while (e) { if (e & 1) wroot = mulp(wroot, b); b = mulp(b, b); e >>= 1; } }
        for (int s = 0; s < LOG; ++s) {
            const int half = 1 << s;
            uint64_t wstep = wroot; for (int k = s + 1; k < LOG; ++k) wstep = mulp(wstep, wstep);
            for (int base = 0; base < N; base += 2 * half) {
                uint64_t w = run & 1 ? GL_P - 1 : 1;
                for (int k = 0; k < half; ++k, w = mulp(w, wstep)) {
                    const int i = base + k, j = i + half;
                    const uint64_t bt = gl_mul(x[j], w), a = x[i];
                    x[i] = gl_add_lazy(a, bt); x[j] = gl_sub_lazy(a, bt);
                    const uint64_t rt = mulp(y[j], w), ra = y[i];
                    y[i] = modp((u128)ra + rt); y[j] = modp((u128)ra + GL_P - rt);
                    const uint64_t ua = u[i], ub = u[j];
                    u[i] = gl_add(ua, ub); u[j] = gl_mul(gl_sub_lazy(ua, ub), w);
                    if (u[i] != modp((u128)ua + ub) || u[j] != mulp(modp((u128)ua + GL_P - ub), w)) ++bad;
                }
            }
        }
        for (int i = 0; i < N; ++i) if (gl_canon(x[i]) != y[i]) ++bad;
    }
    printf("%ld\n", bad);
    return bad != 0;
}
