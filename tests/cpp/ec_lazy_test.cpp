// Host check of the lazy Jacobian formulas (sandstorm_amd/csrc/ec252.h, the code the Pedersen kernels run) against
// the plain 8 x 32 ones on chains of mixed additions over multiples of the Pedersen base point P1
// (builtins/src/pedersen/constants.rs:5-30; tests/golden/pedersen.json), exceptional cases included.
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../../sandstorm_amd/csrc/ec252.h"

using namespace ss;

static uint64_t rng_state = 0x243F6A8885A308D3ull;
static uint64_t splitmix() {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static Fp from_canon64(const uint64_t c[4]) {
    Fp a;
    for (int i = 0; i < 4; ++i) { a.v[2 * i] = (u32)c[i]; a.v[2 * i + 1] = (u32)(c[i] >> 32); }
    return fp_to_mont(a);
}
static Aff to_affine(const Jac &p) {
    const Fp zi = fp_inv(p.z), zi2 = fp_sqr(zi);
    return Aff{fp_mul(p.x, zi2), fp_mul(p.y, fp_mul(zi2, zi))};
}
static Jac lift(const Aff &a) { return Jac{a.x, a.y, fp_one()}; }
static AffL limb(const Aff &a) { return AffL{fl_from_fp(a.x), fl_from_fp(a.y)}; }
static bool same(const JacL &l, const Jac &j) {
    return fp_eq(fl_to_fp(l.x), j.x) && fp_eq(fl_to_fp(l.y), j.y) && fp_eq(fl_to_fp(l.z), j.z);
}

int main() {
    const uint64_t P1X[4] = {0x1080d17957ebe47bull, 0x8fa8120b6d56eb0cull, 0x969c748655fca9e5ull, 0x0234287dcbaffe7full};
    const uint64_t P1Y[4] = {0x6ed0268ee89e5615ull, 0x940135dd7a6c94ccull, 0x1e889527d41f4e39ull, 0x03b056f100f96fb2ull};
    const Aff g{from_canon64(P1X), from_canon64(P1Y)};
    // a pool of affine multiples of g
    std::vector<Aff> pool;
    Jac run = lift(g);
    for (int i = 0; i < 300; ++i) {
        const int steps = 1 + (int)(splitmix() % 5);
        for (int k = 0; k < steps; ++k) run = (splitmix() & 1) ? jac_double(run) : jac_add_aff(run, g);
        pool.push_back(to_affine(run));
    }
    int checked = 0;
    for (int chain = 0; chain < 200; ++chain) {
        Jac acc = lift(pool[splitmix() % pool.size()]);
        JacL accl{fl_from_fp(acc.x), fl_from_fp(acc.y), fl_from_fp(acc.z)};
        for (int i = 0; i < 34; ++i) {                      // a hash is 32 mixed additions
            const Aff &q = pool[splitmix() % pool.size()];
            acc = jac_add_aff(acc, q);
            accl = jacl_add_aff(accl, limb(q));
            if (!same(accl, acc)) { printf("mismatch: chain %d step %d\n", chain, i); return 1; }
            ++checked;
        }
    }
    // exceptional cases: p = q (doubling), p = -q (infinity), p = infinity
    for (int i = 0; i < 50; ++i) {
        const Aff q = pool[splitmix() % pool.size()];
        const Aff t = pool[splitmix() % pool.size()];
        // a Jacobian representative of q with z != 1: (q + t) - t
        Jac p = jac_add_aff(lift(q), t);
        p = jac_add_aff(p, Aff{t.x, fp_neg(t.y)});
        JacL pl{fl_from_fp(p.x), fl_from_fp(p.y), fl_from_fp(p.z)};
        if (!same(jacl_add_aff(pl, limb(q)), jac_add_aff(p, q))) { printf("doubling case mismatch %d\n", i); return 1; }
        const Aff nq{q.x, fp_neg(q.y)};
        const JacL inf = jacl_add_aff(pl, limb(nq));
        if (!fp_is_zero(fl_to_fp(inf.z)) || !fp_is_zero(jac_add_aff(p, nq).z)) { printf("infinity case mismatch %d\n", i); return 1; }
        const JacL back = jacl_add_aff(inf, limb(t));       // infinity + t = t
        if (!fp_eq(fl_to_fp(back.x), t.x) || !fp_eq(fl_to_fp(back.y), t.y)) { printf("from-infinity mismatch %d\n", i); return 1; }
        checked += 3;
    }
    // XYZZ (the accumulate kernels, round 5): chains of mixed additions against the plain Jacobian ones - x = X / ZZ, y = Y / ZZZ -,
    // the doubling and infinity cases, and a start from infinity
    {
        auto affine_of = [](const XyzzL &l) {
            const Fp zz = fl_to_fp(l.zz), zzz = fl_to_fp(l.zzz);
            return Aff{fp_mul(fl_to_fp(l.x), fp_inv(zz)), fp_mul(fl_to_fp(l.y), fp_inv(zzz))};
        };
        for (int chain = 0; chain < 100; ++chain) {
            const Aff start = pool[splitmix() % pool.size()];
            Jac acc = lift(start);
            XyzzL accx{fl_from_fp(start.x), fl_from_fp(start.y), fl_one(), fl_one()};
            for (int i = 0; i < 24; ++i) {
                const Aff &q = pool[splitmix() % pool.size()];
                acc = jac_add_aff(acc, q);
                accx = xyzzl_add_aff(accx, limb(q));
                ++checked;
            }
            if (fp_is_zero(acc.z) != fp_is_zero(fl_to_fp(accx.zz))) { printf("xyzz: infinity disagreement, chain %d\n", chain); return 1; }
            if (fp_is_zero(acc.z)) continue;
            const Aff want = to_affine(acc), got = affine_of(accx);
            if (!fp_eq(got.x, want.x) || !fp_eq(got.y, want.y)) { printf("xyzz chain mismatch %d\n", chain); return 1; }
            // ZZ^3 = ZZZ^2 (the representation's invariant)
            const Fp zz = fl_to_fp(accx.zz), zzz = fl_to_fp(accx.zzz);
            if (!fp_eq(fp_mul(fp_sqr(zz), zz), fp_sqr(zzz))) { printf("xyzz invariant broken %d\n", chain); return 1; }
        }
        for (int i = 0; i < 50; ++i) {
            const Aff q = pool[splitmix() % pool.size()], t = pool[splitmix() % pool.size()];
            XyzzL pl{fl_from_fp(q.x), fl_from_fp(q.y), fl_one(), fl_one()};
            pl = xyzzl_add_aff(pl, limb(t));
            pl = xyzzl_add_aff(pl, limb(Aff{t.x, fp_neg(t.y)}));        // q again, with ZZ != 1
            const Aff twice = to_affine(jac_double(lift(q))), got = affine_of(xyzzl_add_aff(pl, limb(q)));
            if (!fp_eq(got.x, twice.x) || !fp_eq(got.y, twice.y)) { printf("xyzz doubling case mismatch %d\n", i); return 1; }
            const XyzzL inf = xyzzl_add_aff(pl, limb(Aff{q.x, fp_neg(q.y)}));
            if (!fp_is_zero(fl_to_fp(inf.zz))) { printf("xyzz infinity case mismatch %d\n", i); return 1; }
            const XyzzL back = xyzzl_add_aff(inf, limb(t));
            if (!fp_eq(fl_to_fp(back.x), t.x) || !fp_eq(fl_to_fp(back.y), t.y) || !fp_eq(fl_to_fp(back.zz), fp_one())) { printf("xyzz from-infinity mismatch %d\n", i); return 1; }
            checked += 3;
        }
    }
    // The same in the R280 DOMAIN (values x * 2^280, the ten-step reduction alone: what the Pedersen kernels run from round 6 on, their
    // table points stored in that form): XYZZ chains with doublings and infinities, affine + affine, the Jacobian + Jacobian of the
    // small levels - coordinates are ratios (x = X / ZZ, y = Y / ZZZ; X / Z^2, Y / Z^3 after rescaling Z), so the plain inverse reads them
    {
        Fp two24 = fp_zero(); two24.v[0] = 1u << 24;
        const Fp k24 = fp_to_mont(two24), k24inv = fp_inv(k24);
        auto limb280 = [&](const Aff &a) { return AffL{fl_from_fp(fp_mul(a.x, k24)), fl_from_fp(fp_mul(a.y, k24))}; };
        auto affine_of = [](const XyzzL &l) {
            const Fp zz = fl_to_fp(l.zz), zzz = fl_to_fp(l.zzz);
            return Aff{fp_mul(fl_to_fp(l.x), fp_inv(zz)), fp_mul(fl_to_fp(l.y), fp_inv(zzz))};
        };
        auto affine_of_jac = [&](const JacL &l) {          // stored X 2^280, Y 2^280, Z 2^280 -> the R256 images, then X / Z^2, Y / Z^3
            const Jac j{fp_mul(fl_to_fp(l.x), k24inv), fp_mul(fl_to_fp(l.y), k24inv), fp_mul(fl_to_fp(l.z), k24inv)};
            return to_affine(j);
        };
        if (!fp_eq(fl_to_fp(fl_one_r280()), k24)) { printf("fl_one_r280 is not 2^280\n"); return 1; }
        for (int chain = 0; chain < 100; ++chain) {
            const Aff start = pool[splitmix() % pool.size()];
            Jac acc = lift(start);
            const AffL s280 = limb280(start);
            XyzzL accx{s280.x, s280.y, fl_one_r280(), fl_one_r280()};
            for (int i = 0; i < 24; ++i) {
                // every eighth step repeats or negates what the chain holds: the doubling and infinity branches inside a chain
                Aff q = pool[splitmix() % pool.size()];
                if (i % 8 == 5 && !fp_is_zero(acc.z)) { q = to_affine(acc); if (chain & 1) q.y = fp_neg(q.y); }
                acc = jac_add_aff(acc, q);
                accx = xyzzl_add_aff_d<EcR280>(accx, limb280(q));
                ++checked;
                if (fp_is_zero(acc.z) != fp_is_zero(fl_to_fp(accx.zz))) { printf("r280 xyzz: infinity disagreement, chain %d step %d\n", chain, i); return 1; }
                if (fp_is_zero(acc.z)) continue;
                const Aff want = to_affine(acc), got = affine_of(accx);
                if (!fp_eq(got.x, want.x) || !fp_eq(got.y, want.y)) { printf("r280 xyzz chain mismatch %d step %d\n", chain, i); return 1; }
            }
        }
        // affine + affine -> XYZZ (a hash's first addition), generic pairs, q + q and q - q, both domains
        for (int i = 0; i < 300; ++i) {
            const Aff a = pool[splitmix() % pool.size()], b = i % 8 == 0 ? a : i % 8 == 1 ? Aff{a.x, fp_neg(a.y)} : pool[splitmix() % pool.size()];
            const Jac want = jac_add_aff(lift(a), b);
            const XyzzL g280 = xyzzl_add_affs_d<EcR280>(limb280(a), limb280(b)), g256 = xyzzl_add_affs_d<EcR256>(limb(a), limb(b));
            if (fp_is_zero(want.z) != fp_is_zero(fl_to_fp(g280.zz)) || fp_is_zero(want.z) != fp_is_zero(fl_to_fp(g256.zz))) { printf("xyzz affine + affine: infinity disagreement %d\n", i); return 1; }
            if (!fp_is_zero(want.z)) {
                const Aff w = to_affine(want), x280 = affine_of(g280), x256 = affine_of(g256);
                if (!fp_eq(x280.x, w.x) || !fp_eq(x280.y, w.y) || !fp_eq(x256.x, w.x) || !fp_eq(x256.y, w.y)) { printf("xyzz affine + affine mismatch %d\n", i); return 1; }
                // and the chain goes on from it
                const Aff c = pool[splitmix() % pool.size()];
                const Jac w2 = jac_add_aff(want, c);
                if (!fp_is_zero(w2.z)) {
                    const Aff ww = to_affine(w2), gg = affine_of(xyzzl_add_aff_d<EcR280>(g280, limb280(c)));
                    if (!fp_eq(gg.x, ww.x) || !fp_eq(gg.y, ww.y)) { printf("xyzz affine + affine, then + c: mismatch %d\n", i); return 1; }
                }
            }
            checked += 2;
        }
        for (int i = 0; i < 400; ++i) {
            const Aff a = pool[splitmix() % pool.size()], b = i % 8 == 0 ? a : i % 8 == 1 ? Aff{a.x, fp_neg(a.y)} : pool[splitmix() % pool.size()];
            const JacL got = jacl_add_affs_d<EcR280>(limb280(a), false, limb280(b), false);
            const Jac want = jac_add_aff(lift(a), b);
            if (fp_is_zero(want.z) != fp_is_zero(fl_to_fp(got.z))) { printf("r280 affine + affine: infinity disagreement %d\n", i); return 1; }
            if (!fp_is_zero(want.z)) {
                const Aff w = to_affine(want), g2 = affine_of_jac(got);
                if (!fp_eq(g2.x, w.x) || !fp_eq(g2.y, w.y)) { printf("r280 affine + affine mismatch %d\n", i); return 1; }
                // Jacobian + Jacobian and Jacobian + affine on top of it
                const Aff c = pool[splitmix() % pool.size()], d = pool[splitmix() % pool.size()];
                const JacL cd = jacl_add_affs_d<EcR280>(limb280(c), false, limb280(d), false);
                const Jac wcd = jac_add_aff(lift(c), d);
                if (!fp_is_zero(wcd.z)) {
                    const JacL sum = jacl_add_d<EcR280>(got, cd);
                    Jac wsum = jac_add_aff(want, c);
                    wsum = jac_add_aff(wsum, d);
                    if (fp_is_zero(wsum.z) != fp_is_zero(fl_to_fp(sum.z))) { printf("r280 jac + jac: infinity disagreement %d\n", i); return 1; }
                    if (!fp_is_zero(wsum.z)) {
                        const Aff ws = to_affine(wsum), gs = affine_of_jac(sum);
                        if (!fp_eq(gs.x, ws.x) || !fp_eq(gs.y, ws.y)) { printf("r280 jac + jac mismatch %d\n", i); return 1; }
                        const Aff e = pool[splitmix() % pool.size()];
                        const JacL se = jacl_add_aff_d<EcR280>(sum, limb280(e));
                        const Jac wse = jac_add_aff(wsum, e);
                        if (!fp_is_zero(wse.z)) {
                            const Aff w2 = to_affine(wse), g3 = affine_of_jac(se);
                            if (!fp_eq(g3.x, w2.x) || !fp_eq(g3.y, w2.y)) { printf("r280 jac + affine mismatch %d\n", i); return 1; }
                        }
                    }
                }
            }
            checked += 3;
        }
        // the R280 product and square at the lazy bounds
        for (int it = 0; it < 2000; ++it) {
            Fl a, b;
            for (int i = 0; i < 9; ++i) {
                const u32 amax = (1u << 30) + (1u << 27), bmax = (1u << 29) + (1u << 25);
                a.l[i] = it == 0 ? amax : amax - (u32)(splitmix() % (it & 1 ? 16u : amax));
                b.l[i] = it == 0 ? bmax : bmax - (u32)(splitmix() % (it & 2 ? 16u : bmax));
            }
            a.l[8] &= (1u << 29) - 1; b.l[8] &= (1u << 29) - 1;
            // a b 2^-280 = (a b 2^-256) 2^-24
            const Fp want = fp_mul(fp_mul(fl_to_fp(a), fl_to_fp(b)), k24inv), wantsq = fp_mul(fp_sqr(fl_to_fp(b)), k24inv);
            if (!fp_eq(fl_to_fp(fl_mul_r280(a, b)), want)) { printf("fl_mul_r280 at the lazy bounds: mismatch %d\n", it); return 1; }
            if (!fp_eq(fl_to_fp(fl_sqr_r280(b)), wantsq)) { printf("fl_sqr_r280 at the lazy bounds: mismatch %d\n", it); return 1; }
        }
    }
    // affine + affine (the first round of the lane-split accumulation) against the plain mixed addition: generic pairs, q + q, q - q,
    // and the point at infinity on either or both sides
    for (int i = 0; i < 400; ++i) {
        const Aff a = pool[splitmix() % pool.size()], b = i % 8 == 0 ? a : i % 8 == 1 ? Aff{a.x, fp_neg(a.y)} : pool[splitmix() % pool.size()];
        const JacL got = jacl_add_affs(limb(a), false, limb(b), false);
        const Jac want = jac_add_aff(lift(a), b);
        if (fp_is_zero(want.z) ? !fp_is_zero(fl_to_fp(got.z)) : !same(got, want)) { printf("affine + affine mismatch %d\n", i); return 1; }
        const JacL l = jacl_add_affs(limb(a), true, limb(b), false), r = jacl_add_affs(limb(a), false, limb(b), true), n = jacl_add_affs(limb(a), true, limb(b), true);
        if (!same(l, lift(b)) || !same(r, lift(a)) || !fp_is_zero(fl_to_fp(n.z))) { printf("affine + affine with infinity: mismatch %d\n", i); return 1; }
        checked += 4;
    }
    // the multiplier at the limb bounds the lazy formulas reach (fl_sub_c<2,1> output: limbs up to 2^29 + 2^25; the
    // lazy x3: limbs up to 2^30 + 2^27), every limb at its maximum and random patterns near it
    for (int it = 0; it < 2000; ++it) {
        Fl a, b;
        for (int i = 0; i < 9; ++i) {
            const u32 amax = (1u << 30) + (1u << 27), bmax = (1u << 29) + (1u << 25);
            a.l[i] = it == 0 ? amax : amax - (u32)(splitmix() % (it & 1 ? 16u : amax));
            b.l[i] = it == 0 ? bmax : bmax - (u32)(splitmix() % (it & 2 ? 16u : bmax));
        }
        a.l[8] &= (1u << 29) - 1; b.l[8] &= (1u << 29) - 1;       // keep the VALUES below 16p / 4p (top limb = value >> 224)
        const Fp want = fp_mul(fl_to_fp(a), fl_to_fp(b));
        if (!fp_eq(fl_to_fp(fl_mul(a, b)), want)) { printf("fl_mul at the lazy bounds: mismatch %d\n", it); return 1; }
        ++checked;
    }
    printf("ok %d\n", checked);
    return 0;
}
