// pedersen.hip — StarkWare Pedersen hash on gfx950, one lane per hash.
//
// Replaces builtins/src/pedersen/mod.rs:31-36 (pedersen_hash, which the
// reference delegates to starknet-crypto 0.6.1) where it sits on the proving
// hot path: the top N_FRIENDLY_LAYERS of every FriendlyMerkleTree
// (crypto/src/merkle/mixed.rs:106-155; src/claims.rs:10) and the single-column
// leaf rule PedersenHashFn::hash_elements (crypto/src/hash/pedersen.rs:65-76).
//
//   H(a, b) = [P0 + a_low*P1 + a_high*P2 + b_low*P3 + b_high*P4].x
//
// (a_low = low 248 bits, a_high = top 4 bits of the canonical value;
// mod.rs:25-30; points from builtins/src/pedersen/constants.rs:5-30.)
//
// Fixed-base windows of W bits (16, 18, 20, 22 or 24; SS_PED_WINDOW, default PED_DEFAULT_WINDOW).  The 252 scalar bits of an
// input are one bit string - bit i < 248 stands for 2^i P_low, bit 248 + i for 2^i P_high (the 4-bit high part has its
// own base point) - cut into ceil(252 / W) windows; the device table holds, per input and window, the 2^W - 1 non-zero
// subset sums as affine points: ceil(252 / W) Jacobian+affine mixed additions (8M + 3S) per input - 16 / 14 / 13 - against
// 32 with 8-bit windows.  Table sizes 134 MB (16 bits: Infinity-Cache resident) / 470 MB / 1.7 GB / 6.4 GB / 23.6 GB (24 bits:
// 11 additions per input, 64-byte gathers from HBM - 8 % of the 288 GB buy 18 % of the Pedersen layers' time).  Built once per process and device ON THE DEVICE from the 2 x 252 bit points, in two levels (half windows
// entry by entry, then every entry as one affine addition of its halves with batched inversions).  The final Jacobian -> affine inversion (~310 multiplications, a
// quarter of a hash) is not done per hash: the accumulate kernel leaves (X, Z) in a
// temporary and a second kernel inverts Z in per-lane chunks with Montgomery's trick
// (5 multiplications per hash + one inversion per chunk).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <mutex>
#include <vector>
#include "fp252.h"
#include "fl252.h"
#include "inv252.h"
#include "fp252_host.h"
#include "ec252.h"
#include "kernels.h"

namespace ss {


static const uint64_t PED_CANON[5][2][4] = {
    {{0x551fde4050ca6804ull, 0x716b0b1022947733ull, 0x00ee1b87eb599f16ull, 0x049ee3eba8c16007ull},
     {0xd0405d266e10268aull, 0x4e621062c0e056c1ull, 0xf346d49d06ea0ed3ull, 0x03ca0cfe4b3bc6ddull}},
    {{0x1080d17957ebe47bull, 0x8fa8120b6d56eb0cull, 0x969c748655fca9e5ull, 0x0234287dcbaffe7full},
     {0x6ed0268ee89e5615ull, 0x940135dd7a6c94ccull, 0x1e889527d41f4e39ull, 0x03b056f100f96fb2ull}},
    {{0xb7a6932dba8aa378ull, 0x99099ec1de5e3018ull, 0x3f9dab2656558f33ull, 0x04fa56f376c83db3ull},
     {0x5168f4e80ff5b54dull, 0x562761f92a7a23b4ull, 0x8113e0c0e47e4401ull, 0x03fa0984c931c9e3ull}},
    {{0x3aa372f0bd2d6997ull, 0x40c690c74709e90full, 0x764910f75b45f74bull, 0x04ba4cc166be8decull},
     {0x48151f27b24b219cull, 0xcac5c59a5ce5ae7cull, 0x4b971e46c4ede85full, 0x0040301cf5c1751full}},
    {{0xd36ff12c49a58202ull, 0x2ca65048d53fb325ull, 0x6e44cca8f61a63bbull, 0x054302dcb0e6cc1cull},
     {0x879dcc77e99c2426ull, 0xce98ad783c25561aull, 0xb348046268d8ae25ull, 0x01b77b3e37d13504ull}},
};

static constexpr int PED_WINDOWS = 31;            // 8-bit windows over the low 248 bits
static constexpr int PED_LOW_ENTRIES = PED_WINDOWS * 255;
static constexpr int PED_HIGH_ENTRIES = 15;
static constexpr int PED_PER_INPUT = PED_LOW_ENTRIES + PED_HIGH_ENTRIES;

// device table layout per input: [nwin][2^W - 1] (window w covers scalar bits W w .. W w + W - 1 of the 252)
static constexpr int PED_BITS = 252;
// the default window: the widest of 24 .. 16 bits that fits (pedersen_tables_create); measured (profiles/r03_pedersen_windows.txt): Merkle stage of recursive_2p20 53.3 / 49.8 / 47.5 / 46.3 / 43.6 ms for 16 .. 24
static constexpr int PED_MAX_WINDOWS = 16;        // the lane-split kernel gives a window to each of 16 lanes per input

struct PedersenTables {
    Aff *d_table;   // [2][nwin * span]
    Aff shift;      // P0, in the R280 domain like the table's points (x 2^280: pedersen_tables_build)
    uint32_t W, nwin, span;      // window bits, windows per input, entries per window (2^W - 1)
    uint64_t per_input;          // nwin * span
    int device, users;           // the tables are constants: ONE copy per device and process, shared by its contexts
};
static std::mutex g_ped_mutex;
static std::vector<PedersenTables *> g_ped_tables;      // at most one per device: kept while the process lives (a context
                                                        // that comes later - every test makes its own - finds it built)

// --------------------------------------------------------------- host side
static Fp canon_to_mont(const uint64_t c[4]) {
    Fp a;
    for (int i = 0; i < 4; ++i) { a.v[2 * i] = (u32)c[i]; a.v[2 * i + 1] = (u32)(c[i] >> 32); }
    return fp_to_mont(a);
}

static void batch_to_affine(std::vector<Jac> &pts, std::vector<Aff> &out) {
    const size_t n = pts.size();
    std::vector<Fp> prefix(n);
    Fp acc = fp_one();
    for (size_t i = 0; i < n; ++i) { prefix[i] = acc; acc = fp_mul(acc, pts[i].z); }
    Fp inv = fp_inv(acc);
    out.resize(n);
    for (size_t i = n; i-- > 0;) {
        Fp zi = fp_mul(inv, prefix[i]);
        inv = fp_mul(inv, pts[i].z);
        Fp zi2 = fp_sqr(zi);
        out[i].x = fp_mul(pts[i].x, zi2);
        out[i].y = fp_mul(pts[i].y, fp_mul(zi2, zi));
    }
}

// the windowed table on the host: [2][PED_PER_INPUT] affine points, plus P0
static void build_host_tables(std::vector<Aff> &aff, Aff &shift) {
    Aff pts[5];
    for (int k = 0; k < 5; ++k) { pts[k].x = canon_to_mont(PED_CANON[k][0]); pts[k].y = canon_to_mont(PED_CANON[k][1]); }
    std::vector<Jac> jac;
    jac.reserve(2 * PED_PER_INPUT);
    for (int e = 0; e < 2; ++e) {
        // low part: base 2^(8j) * P_{1+2e}
        Jac base; base.x = pts[1 + 2 * e].x; base.y = pts[1 + 2 * e].y; base.z = fp_one();
        for (int j = 0; j < PED_WINDOWS; ++j) {
            std::vector<Jac> one(1, base);
            std::vector<Aff> ba;
            batch_to_affine(one, ba);
            Jac acc = base;
            jac.push_back(acc);                       // d = 1
            acc = jac_double(base);
            jac.push_back(acc);                       // d = 2
            for (int d = 3; d <= 255; ++d) { acc = jac_add_aff(acc, ba[0]); jac.push_back(acc); }
            for (int k = 0; k < 8; ++k) base = jac_double(base);
        }
        // high part: d * P_{2+2e}, d = 1..15
        Aff hb = pts[2 + 2 * e];
        Jac acc; acc.x = hb.x; acc.y = hb.y; acc.z = fp_one();
        jac.push_back(acc);
        Jac dbl = jac_double(acc);
        jac.push_back(dbl);
        acc = dbl;
        for (int d = 3; d <= 15; ++d) { acc = jac_add_aff(acc, hb); jac.push_back(acc); }
    }
    batch_to_affine(jac, aff);
    shift = pts[0];
}

static const std::vector<Aff> &host_tables(Aff *shift) {
    static std::vector<Aff> aff;
    static Aff sh;
    static std::once_flag once;
    std::call_once(once, [] { build_host_tables(aff, sh); });
    if (shift) *shift = sh;
    return aff;
}

__global__ void pedersen_build_windows_kernel(const Aff *__restrict__ bit_points, Aff *__restrict__ table, uint32_t W, uint32_t span, uint64_t per_input);
__global__ void pedersen_join_halves_kernel(const Aff *__restrict__ half, Aff *__restrict__ table, uint32_t W, uint32_t nwin, uint32_t nhalf);
__global__ void pedersen_table_to_r280_kernel(Fp *__restrict__ coords, uint64_t n);

// the 2 x 252 bit points: bit i of input e stands for 2^i P_{1+2e} (i < 248) or 2^(i-248) P_{2+2e}
static void build_bit_points(std::vector<Aff> &out) {
    Aff pts[5];
    for (int k = 0; k < 5; ++k) { pts[k].x = canon_to_mont(PED_CANON[k][0]); pts[k].y = canon_to_mont(PED_CANON[k][1]); }
    std::vector<Jac> jac;
    for (int e = 0; e < 2; ++e) {
        Jac cur; cur.x = pts[1 + 2 * e].x; cur.y = pts[1 + 2 * e].y; cur.z = fp_one();
        for (int i = 0; i < 248; ++i) { jac.push_back(cur); cur = jac_double(cur); }
        cur.x = pts[2 + 2 * e].x; cur.y = pts[2 + 2 * e].y; cur.z = fp_one();
        for (int i = 248; i < PED_BITS; ++i) { jac.push_back(cur); cur = jac_double(cur); }
    }
    batch_to_affine(jac, out);
}

static size_t ped_table_bytes(uint32_t W) {
    const uint64_t nwin = (PED_BITS + W - 1) / W, span = (1ull << W) - 1ull;
    return (size_t)(2ull * nwin * span * sizeof(Aff));
}
static hipError_t pedersen_tables_build(hipStream_t st, uint32_t W, int device, const Aff &shift, PedersenTables **out) {
    std::vector<Aff> bits;
    build_bit_points(bits);
    PedersenTables *t = new PedersenTables;
    t->device = device; t->users = 1;
    t->shift = shift;
    t->d_table = nullptr;
    t->W = W; t->nwin = (PED_BITS + W - 1) / W; t->span = (1u << W) - 1u;
    t->per_input = (uint64_t)t->nwin * t->span;
    // Two levels: the windows of W / 2 bits straight from the bit points (2 x ceil(252 / (W/2)) x (2^(W/2) - 1) entries: thousands),
    // then every W-bit entry as ONE affine + affine addition of its two halves, the slopes' denominators inverted eight at a
    // time (6 multiplications and an eighth of an inversion per entry: 0.1 s for the 3.7 10^8 entries of W = 24, where summing
    // an entry's bit points one by one took 1.4 s)
    const uint32_t h = W / 2, nhalf = (PED_BITS + h - 1) / h, spanh = (1u << h) - 1u;
    const uint64_t per_input_half = (uint64_t)nhalf * spanh;
    Aff *d_bits = nullptr, *d_half = nullptr;
    hipError_t e = hipMalloc(&d_bits, bits.size() * sizeof(Aff));
    if (e == hipSuccess) e = hipMalloc(&d_half, 2 * per_input_half * sizeof(Aff));
    if (e == hipSuccess) e = hipMalloc(&t->d_table, 2 * t->per_input * sizeof(Aff));
    if (e == hipSuccess) e = hipMemcpyAsync(d_bits, bits.data(), bits.size() * sizeof(Aff), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        const uint64_t total = 2 * per_input_half;
        hipLaunchKernelGGL(pedersen_build_windows_kernel, dim3((uint32_t)((total + 63) / 64)), dim3(64), 0, st, d_bits, d_half, h, spanh, per_input_half);
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        const uint64_t lanes = 2ull * t->nwin * ((1ull << W) / 8);
        hipLaunchKernelGGL(pedersen_join_halves_kernel, dim3((uint32_t)((lanes + 63) / 64)), dim3(64), 0, st, d_half, t->d_table, W, t->nwin, nhalf);
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        // the accumulate kernels run in the R280 domain (ec252.h EcR280): every coordinate of the finished table times 2^24 (once per
        // process: 7.4 10^8 products for W = 24), and the shift point P0 with them
        const uint64_t coords = 4 * t->per_input;
        hipLaunchKernelGGL(pedersen_table_to_r280_kernel, dim3((uint32_t)((coords + 255) / 256)), dim3(256), 0, st, (Fp *)t->d_table, coords);
        e = hipGetLastError();
        Fp two24 = fp_zero(); two24.v[0] = 1u << 24;
        const Fp k24 = fp_to_mont(two24);
        t->shift.x = fp_mul(shift.x, k24); t->shift.y = fp_mul(shift.y, k24);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (d_bits) (void)hipFree(d_bits);
    if (d_half) (void)hipFree(d_half);
    if (e != hipSuccess) { if (t->d_table) (void)hipFree(t->d_table); delete t; (void)hipGetLastError(); return e; }
    *out = t;
    return hipSuccess;
}

// The window width: SS_PED_WINDOW if set (an error if that table does not fit), else the widest of 24 / 22 / 20 / 18 / 16 bits
// whose table fits the device's FREE memory with 4 GiB to spare (23.6 / 6.4 / 1.75 / 0.47 / 0.13 GB: ranks that share one GPU, or
// a GPU with little room, get a smaller table - 10 of the north-star proof's 218 ms - instead of an allocation error), stepping
// down again if the allocation itself fails.
hipError_t pedersen_tables_create(hipStream_t st, PedersenTables **out) {
    Aff shift;
    (void)host_tables(&shift);
    static const uint32_t widths[] = {24, 22, 20, 18, 16};
    uint32_t forced = 0;
    if (const char *e = getenv("SS_PED_WINDOW")) {
        forced = (uint32_t)strtoul(e, nullptr, 10);
        // 26: ten windows per input instead of eleven for an 86 GB table - asked for, never chosen by itself (A/B: profiles/r05_pedersen_*)
        if (forced != 16 && forced != 18 && forced != 20 && forced != 22 && forced != 24 && forced != 26) return hipErrorInvalidValue;
    }
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_ped_mutex);
    for (size_t k = 0; k < g_ped_tables.size(); ++k) {
        PedersenTables *c = g_ped_tables[k];
        if (c->device != device) continue;
        if (!forced || c->W == forced) { c->users += 1; *out = c; return hipSuccess; }     // the copy the process already has
        if (c->users == 0) {                     // another width was asked for (SS_PED_WINDOW changed): replace the idle copy
            (void)hipFree(c->d_table);
            delete c;
            g_ped_tables.erase(g_ped_tables.begin() + k);
        }
        break;
    }
    hipError_t e = hipErrorOutOfMemory;
    PedersenTables *t = nullptr;
    if (forced) e = pedersen_tables_build(st, forced, device, shift, &t);
    else {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = ~(size_t)0;
        for (uint32_t W : widths) {
            if (W != 16 && ped_table_bytes(W) + ((size_t)4 << 30) > free_b) continue;
            e = pedersen_tables_build(st, W, device, shift, &t);
            if (e != hipErrorOutOfMemory) break;
        }
    }
    if (e != hipSuccess) return e;
    g_ped_tables.push_back(t);
    *out = t;
    return hipSuccess;
}

// ss_ctx_trim: the tables no context of this process uses any more go back to the driver (they are rebuilt in ~0.1 s)
void pedersen_tables_trim() {
    std::lock_guard<std::mutex> lock(g_ped_mutex);
    for (size_t k = 0; k < g_ped_tables.size();) {
        PedersenTables *c = g_ped_tables[k];
        if (c->users == 0) {
            int prev = 0;
            const bool sw = hipGetDevice(&prev) == hipSuccess && prev != c->device && hipSetDevice(c->device) == hipSuccess;
            (void)hipFree(c->d_table);
            if (sw) (void)hipSetDevice(prev);
            delete c;
            g_ped_tables.erase(g_ped_tables.begin() + k);
        } else ++k;
    }
}

// pedersen_hash on the host, same windowed tables (used by the Fiat-Shamir coin only: the CairoVerifierPublicCoin chains ~155 of
// them per proof through reseed_with_field_elements, crypto/src/public_coin/cairo.rs).  The 8 x 32-bit arithmetic of fp252.h is
// shaped for the GPU; a CPU core does the same Montgomery product in 4 x 64-bit limbs (fp252_host.h) in a tenth of the time, and
// safegcd replaces the 251-squaring power: 49 -> 3.8 us per hash.
namespace {
struct JacH { H4 x, y, z; };
const H4 H4_ONE = {{0xffffffffffffffe1ull, 0xffffffffffffffffull, 0xffffffffffffffffull, 0x07fffffffffffdf0ull}};    // 2^256 mod p
JacH jach_double(const JacH &p) {                                     // a = 1 (the curve's coefficient), as jac_double
    const H4 xx = h4_sqr(p.x), yy = h4_sqr(p.y), yyyy = h4_sqr(yy), zz = h4_sqr(p.z);
    H4 s = h4_mul(p.x, yy); s = h4_add(s, s); s = h4_add(s, s);
    const H4 m = h4_add(h4_add(h4_add(xx, xx), xx), h4_sqr(zz));
    JacH r;
    r.x = h4_sub(h4_sqr(m), h4_add(s, s));
    H4 y8 = h4_add(yyyy, yyyy); y8 = h4_add(y8, y8); y8 = h4_add(y8, y8);
    r.y = h4_sub(h4_mul(m, h4_sub(s, r.x)), y8);
    const H4 yz = h4_mul(p.y, p.z);
    r.z = h4_add(yz, yz);
    return r;
}
JacH jach_add_aff(const JacH &p, const H4 &qx, const H4 &qy) {       // the cases of jac_add_aff (ec252.h), in its order
    if (h4_is_zero(p.z)) { JacH r; r.x = qx; r.y = qy; r.z = H4_ONE; return r; }
    const H4 zz = h4_sqr(p.z);
    const H4 u2 = h4_mul(qx, zz), s2 = h4_mul(qy, h4_mul(zz, p.z));
    const H4 h = h4_sub(u2, p.x), rr = h4_sub(s2, p.y);
    if (h4_is_zero(h)) {
        if (h4_is_zero(rr)) return jach_double(p);
        JacH o; o.x = H4_ONE; o.y = H4_ONE; o.z = H4{{0, 0, 0, 0}}; return o;
    }
    const H4 hh = h4_sqr(h), hhh = h4_mul(hh, h), v = h4_mul(p.x, hh);
    JacH r;
    r.x = h4_sub(h4_sub(h4_sqr(rr), hhh), h4_add(v, v));
    r.y = h4_sub(h4_mul(rr, h4_sub(v, r.x)), h4_mul(p.y, hhh));
    r.z = h4_mul(p.z, h);
    return r;
}
}  // namespace
Fp pedersen_hash_host(const Fp &a, const Fp &b) {
    Aff shift;
    const std::vector<Aff> &tab = host_tables(&shift);
    JacH acc; acc.x = h4_from_fp(shift.x); acc.y = h4_from_fp(shift.y); acc.z = H4_ONE;
    const Fp in[2] = {fp_from_mont(a), fp_from_mont(b)};
    for (int e = 0; e < 2; ++e) {
        const Aff *t = tab.data() + e * PED_PER_INPUT;
        for (int j = 0; j < PED_WINDOWS; ++j) {
            const u32 d = (in[e].v[j >> 2] >> (8 * (j & 3))) & 0xffu;
            if (d) { const Aff &q = t[j * 255 + (d - 1)]; acc = jach_add_aff(acc, h4_from_fp(q.x), h4_from_fp(q.y)); }
        }
        const u32 dh = (in[e].v[7] >> 24) & 0xfu;
        if (dh) { const Aff &q = t[PED_LOW_ENTRIES + (dh - 1)]; acc = jach_add_aff(acc, h4_from_fp(q.x), h4_from_fp(q.y)); }
    }
    if (h4_is_zero(acc.z)) return fp_zero();                           // unreachable for a hash; fp_inv(0) = 0 gave the same
    const H4 zi = h4_from_fp(fp_inv_safegcd(h4_to_fp(acc.z)));
    return h4_to_fp(h4_mul(acc.x, h4_sqr(zi)));
}

void pedersen_tables_destroy(PedersenTables *t) {       // a context lets go of the shared copy (which stays for the next one)
    if (!t) return;
    std::lock_guard<std::mutex> lock(g_ped_mutex);
    if (t->users > 0) t->users -= 1;
}

// ------------------------------------------------------------- device side
__device__ __forceinline__ Aff load_aff(const Aff *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    Aff r;
    r.x.v[0] = a.x; r.x.v[1] = a.y; r.x.v[2] = a.z; r.x.v[3] = a.w;
    r.x.v[4] = b.x; r.x.v[5] = b.y; r.x.v[6] = b.z; r.x.v[7] = b.w;
    r.y.v[0] = c.x; r.y.v[1] = c.y; r.y.v[2] = c.z; r.y.v[3] = c.w;
    r.y.v[4] = d.x; r.y.v[5] = d.y; r.y.v[6] = d.z; r.y.v[7] = d.w;
    return r;
}

__device__ __forceinline__ AffL load_affl(const Aff *p) {
    const Aff q = load_aff(p);
    AffL r; r.x = fl_from_fp(q.x); r.y = fl_from_fp(q.y);
    return r;
}

__device__ __forceinline__ Fp load_felt(const Fp *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    Fp r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__device__ __forceinline__ void store_felt(Fp *p, const Fp &x) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    q[1] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
}

__device__ __forceinline__ void store_aff(Aff *p, const Aff &a) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(a.x.v[0], a.x.v[1], a.x.v[2], a.x.v[3]);
    q[1] = make_uint4(a.x.v[4], a.x.v[5], a.x.v[6], a.x.v[7]);
    q[2] = make_uint4(a.y.v[0], a.y.v[1], a.y.v[2], a.y.v[3]);
    q[3] = make_uint4(a.y.v[4], a.y.v[5], a.y.v[6], a.y.v[7]);
}

// entry j (1 .. 2^W - 1) of window w of input e: the sum of the bit points of j's set bits (bits beyond the 252nd do not exist:
// the top window is narrower and its upper entries are never addressed)
__global__ __launch_bounds__(64) void pedersen_build_windows_kernel(const Aff *__restrict__ bit_points, Aff *__restrict__ table, uint32_t W,
                                                                    uint32_t span, uint64_t per_input) {
    const uint64_t idx = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (idx >= 2 * per_input) return;
    const uint32_t e = (uint32_t)(idx / per_input);
    const uint64_t r = idx % per_input;
    const uint32_t w = (uint32_t)(r / span), j = (uint32_t)(r % span) + 1u, pos = w * W;
    const Aff *bits = bit_points + (size_t)e * PED_BITS;
    JacL acc; acc.x = fl_one(); acc.y = fl_one(); acc.z = fl_zero();
    int terms = 0;
#pragma unroll 1
    for (uint32_t b = 0; b < W && pos + b < (uint32_t)PED_BITS; ++b)
        if ((j >> b) & 1u) { acc = jacl_add_aff(acc, load_affl(bits + pos + b)); ++terms; }
    Aff o;
    if (terms == 0) { o.x = fp_zero(); o.y = fp_zero(); }                     // only bits that do not exist: never addressed
    else if (terms == 1) { o.x = fl_to_fp(acc.x); o.y = fl_to_fp(acc.y); }    // z = 1
    else {
        const Fl zi = fn_inv(acc.z), zi2 = fn_sqr(zi);
        o.x = fl_to_fp(fn_mul(acc.x, zi2));
        o.y = fl_to_fp(fn_mul(acc.y, fn_mul(zi2, zi)));
    }
    store_aff(table + idx, o);
}

// entry j = lo + 2^(W/2) hi of window w = entry lo of half window 2w + entry hi of half window 2w + 1 (affine points: the chord
// rule, its denominator x_lo - x_hi inverted together with those of the lane's other seven entries - same hi, consecutive lo)
__global__ __launch_bounds__(64) void pedersen_join_halves_kernel(const Aff *__restrict__ half, Aff *__restrict__ table, uint32_t W,
                                                                  uint32_t nwin, uint32_t nhalf) {
    constexpr int CH = 8;
    const uint32_t h = W / 2, spanh = (1u << h) - 1u, span = (1u << W) - 1u;
    const uint64_t per_w = (1ull << W) / CH, c = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (c >= 2ull * nwin * per_w) return;
    const uint32_t e = (uint32_t)(c / ((uint64_t)nwin * per_w)), w = (uint32_t)((c / per_w) % nwin), j0 = (uint32_t)(c % per_w) * CH;
    const uint32_t hi = j0 >> h;
    const Aff *lo_tab = half + ((size_t)e * nhalf + 2 * w) * spanh;
    const bool has_hi_tab = 2 * w + 1 < nhalf;               // the narrower top window may have no upper half: those entries are never addressed
    const Aff *hi_tab = half + ((size_t)e * nhalf + 2 * w + 1) * spanh;
    Aff *out = table + ((size_t)e * nwin + w) * span;
    Aff P; P.x = fp_zero(); P.y = fp_zero();
    if (hi && has_hi_tab) P = load_aff(hi_tab + (hi - 1));
    const Fl px = fl_from_fp(P.x), py = fl_from_fp(P.y);
    Fl prefix[CH], run = fl_one();
#pragma unroll
    for (int k = 0; k < CH; ++k) {                           // prefix products of the denominators (1 where there is nothing to add)
        const uint32_t lo = (j0 + k) & spanh;
        prefix[k] = run;
        if (hi && lo && has_hi_tab) {
            Fl d = fn_sub(fl_from_fp(load_felt(&lo_tab[lo - 1].x)), px);
            if (fn_is_zero(d)) d = fl_one();                 // equal abscissae (does not happen for these multiples): left as garbage, not as a fault
            run = fn_mul(run, d);
        }
    }
    Fl inv = fn_inv(run);
#pragma unroll
    for (int k = CH - 1; k >= 0; --k) {
        const uint32_t j = j0 + k, lo = j & spanh;
        if (j == 0) continue;                                // no entry for the zero digit
        Aff o;
        if (!hi || !has_hi_tab) {
            if (!lo || hi) { o.x = fp_zero(); o.y = fp_zero(); } else o = load_aff(lo_tab + (lo - 1));
        } else if (!lo) {
            o = P;
        } else {
            const Aff Q = load_aff(lo_tab + (lo - 1));
            const Fl qx = fl_from_fp(Q.x), qy = fl_from_fp(Q.y);
            Fl d = fn_sub(qx, px);
            if (fn_is_zero(d)) d = fl_one();
            const Fl dinv = fn_mul(inv, prefix[k]);
            inv = fn_mul(inv, d);
            const Fl lam = fn_mul(fn_sub(qy, py), dinv);
            const Fl x3 = fn_sub(fn_sub(fn_sqr(lam), px), qx);
            const Fl y3 = fn_sub(fn_mul(lam, fn_sub(px, x3)), py);
            o.x = fl_to_fp(x3); o.y = fl_to_fp(y3);
        }
        store_aff(out + (j - 1), o);
    }
}

// the finished table into the R280 domain: coordinate c (a Montgomery image x 2^256) -> x 2^280 = c * 2^24
__global__ __launch_bounds__(256) void pedersen_table_to_r280_kernel(Fp *__restrict__ coords, uint64_t n) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp two24 = fp_zero(); two24.v[0] = 1u << 24;
    store_felt(coords + i, fp_mul(load_felt(coords + i), fp_to_mont(two24)));
}

// the canonical scalar as a shift register: its low W bits are the next window's digit (no dynamically indexed limbs)
template <int W>
__device__ __forceinline__ u32 ped_next_digit(Fp &c) {
    const u32 d = c.v[0] & ((1u << W) - 1u);
#pragma unroll
    for (int i = 0; i < 7; ++i) c.v[i] = (c.v[i] >> W) | (c.v[i + 1] << (32 - W));
    c.v[7] >>= W;
    return d;
}

// One hash per lane accumulates in XYZZ coordinates (ec252.h: 8M + 2S per mixed addition, a squaring fewer than the Jacobian form;
// the finish kernel then divides X by ZZ itself instead of squaring an inverse), in the R280 DOMAIN (values x 2^280, the table's
// points stored that way: every product is the ten-step reduction alone, 25 instructions fewer than the 2^256 form's - round 6).
// X and ZZ leave as x 2^280 images; the finish kernel reads them as Montgomery images of X 2^24 and ZZ 2^24: its inverse of ZZ
// is ZZ^-1 2^232, and the 2^256-form product with X 2^280 is (X / ZZ) 2^256 - the interchange form, no conversion anywhere.
typedef XyzzL PedAcc;
__device__ __forceinline__ PedAcc ped_acc_start(const Aff &shift) { return PedAcc{fl_from_fp(shift.x), fl_from_fp(shift.y), fl_one_r280(), fl_one_r280()}; }
__device__ __forceinline__ PedAcc ped_acc_add(const PedAcc &acc, const AffL &q) { return xyzzl_add_aff_d<EcR280>(acc, q); }
__device__ __forceinline__ const Fl &ped_acc_den(const PedAcc &acc) { return acc.zz; }

// acc += scalar (canonical integer limbs) over input slot e
template <int W>
// (fresh: the accumulator is still the affine shift point - its first addition is affine + affine, 4M + 2S instead of 8M + 2S)
__device__ __forceinline__ void ped_accumulate(PedAcc &acc, bool &fresh, Fp canon, const Aff *__restrict__ table, uint64_t per_input, int e) {
    constexpr uint32_t span = (1u << W) - 1u;
    constexpr int nwin = (PED_BITS + W - 1) / W;
    const Aff *tab = table + (size_t)e * per_input;
#pragma unroll 1
    for (int w = 0; w < nwin; ++w) {
        const u32 d = ped_next_digit<W>(canon);
        if (d) {
            const AffL q = load_affl(tab + (size_t)w * span + (d - 1));
            if (fresh) { acc = xyzzl_add_affs_d<EcR280>(AffL{acc.x, acc.y}, q); fresh = false; }
            else acc = ped_acc_add(acc, q);
        }
    }
}

// both inputs canonical (< p); leaves X and the denominator's root (XYZZ: ZZ itself, x = X / ZZ; Jacobian: Z, x = X / Z^2) of
// P0 + a-part + b-part as weakly reduced images
template <int W>
__device__ __forceinline__ void ped_jacobian(const Fp &a, const Fp &b, const Aff *__restrict__ table, uint64_t per_input, const Aff &shift,
                                             Fp *__restrict__ x_out, Fp *__restrict__ z_out) {
    PedAcc acc = ped_acc_start(shift);
    bool fresh = true;
    ped_accumulate<W>(acc, fresh, a, table, per_input, 0);
    ped_accumulate<W>(acc, fresh, b, table, per_input, 1);
    store_felt(x_out, fl_pack(acc.x));            // fn_* results are normalised and < 2p
    store_felt(z_out, fl_pack(ped_acc_den(acc)));
}

// 32 big-endian bytes -> canonical integer mod p
__device__ __forceinline__ Fp be_bytes_to_canon(const uint8_t *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    Fp v;
    v.v[7] = __builtin_bswap32(a.x); v.v[6] = __builtin_bswap32(a.y); v.v[5] = __builtin_bswap32(a.z); v.v[4] = __builtin_bswap32(a.w);
    v.v[3] = __builtin_bswap32(b.x); v.v[2] = __builtin_bswap32(b.y); v.v[1] = __builtin_bswap32(b.z); v.v[0] = __builtin_bswap32(b.w);
    // value < 2^256 < 32p: subtract p while >= p (digests entering the tree are < 2^160 or < p already)
    for (int i = 0; i < 32; ++i) {
        Fp r = fp_reduce_once(v);
        if (fp_eq(r, v)) break;
        v = r;
    }
    return v;
}
__device__ __forceinline__ void canon_to_be_bytes(const Fp &c, uint8_t *out) {
    uint4 *q = reinterpret_cast<uint4 *>(out);
    q[0] = make_uint4(__builtin_bswap32(c.v[7]), __builtin_bswap32(c.v[6]), __builtin_bswap32(c.v[5]), __builtin_bswap32(c.v[4]));
    q[1] = make_uint4(__builtin_bswap32(c.v[3]), __builtin_bswap32(c.v[2]), __builtin_bswap32(c.v[1]), __builtin_bswap32(c.v[0]));
}
// ---- phase 1: accumulate.  xz[k] = X, xz[count + k] = Z of hash k ---------------------------
// inputs as Montgomery felts: a[k * a_stride] (or the canonical constant a_const when a == nullptr), same for b
struct PedFeltArgs {
    const Fp *a; uint64_t a_stride; Fp a_const;
    const Fp *b; uint64_t b_stride; Fp b_const;
};
template <int W>
__global__ __launch_bounds__(64) void pedersen_acc_felts_kernel(const Aff *__restrict__ table, uint64_t per_input, Aff shift, PedFeltArgs g,
                                                                uint64_t count, Fp *__restrict__ xz) {
    const uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (k >= count) return;
    const Fp ca = g.a ? fp_from_mont(load_felt(g.a + k * g.a_stride)) : g.a_const;
    const Fp cb = g.b ? fp_from_mont(load_felt(g.b + k * g.b_stride)) : g.b_const;
    ped_jacobian<W>(ca, cb, table, per_input, shift, xz + k, xz + count + k);
}
// inputs as 32-byte big-endian digests: (in[2k], in[2k+1])
template <int W>
__global__ __launch_bounds__(64) void pedersen_acc_pairs_kernel(const Aff *__restrict__ table, uint64_t per_input, Aff shift,
                                                                const uint8_t *__restrict__ in, uint64_t count,
                                                                Fp *__restrict__ xz) {
    const uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (k >= count) return;
    const Fp ca = be_bytes_to_canon(in + 64 * k), cb = be_bytes_to_canon(in + 64 * k + 32);
    ped_jacobian<W>(ca, cb, table, per_input, shift, xz + k, xz + count + k);
}

// Small levels (the top of every tree: a few thousand hashes or fewer, one level after the other) are latency bound: 22 - 32
// dependent mixed additions per hash with most of the chip idle.  There, 32 lanes share one hash (pedersen_pairs_small_kernel below).
__device__ __forceinline__ Fl fl_shfl_xor(const Fl &a, int mask) {
    Fl r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = (u32)__shfl_xor((int)a.l[i], mask, 64);
    return r;
}
static constexpr uint64_t PED_SPLIT_MAX = 8192;             // hashes per level at or below which 32 lanes share a hash (measured: 2048 .. 32768,
                                                            // profiles/r03_pedersen_small_levels.txt)

// The same level in ONE launch, shaped for latency (the top dozen levels of every tree are a chain of launches that each wait
// for the one before: 32 lanes per hash, most of the chip idle):
//   - the table points are affine, so the first butterfly round is an affine + affine addition (4M + 2S against 12M + 4S);
//   - with fewer than 16 windows per input a lane is idle anyway: one of them carries the shift point P0 into the butterfly
//     instead of a seventh, dependent addition at the end;
//   - lane 0 of the hash inverts its own Z - variable-time division steps (inv252.h): nothing shares the wave's time with it -
//     and writes the digest: no second launch, no round trip of (X, Z) through memory.
// ---- four lanes, one value: the lanes of a QUAD (lanes 4j .. 4j + 3) read each other's registers with DPP quad permutes - full
// rate, no LDS.  The butterfly below reduces over lane bits 0 and 1 first, so from its second round on the four lanes of a quad
// hold the same two points, and at its end the same sum: work that is a few independent multiplications deep is dealt out to
// the quad by ROLE (= lane & 3) and read back.
template <int K> __device__ __forceinline__ u32 ped_quad(u32 v) { return (u32)__builtin_amdgcn_mov_dpp((int)v, K * 0x55, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ u32 ped_quad_perm(u32 v) { return (u32)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); }
template <int K> __device__ __forceinline__ Fl fl_quad(const Fl &mine) {      // role K's value
    Fl r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = ped_quad<K>(mine.l[i]);
    return r;
}
// role's operand of a level, picked with masks: a ternary over the four values invites the compiler to pick an ADDRESS instead - the
// values parked in scratch, a branchy pointer select and a scratch load per limb (seen in the first version's assembly)
__device__ __forceinline__ Fl ped_by_role(uint32_t role, const Fl &a0, const Fl &a1, const Fl &a2, const Fl &a3) {
    const u32 m0 = role == 0u ? ~0u : 0u, m1 = role == 1u ? ~0u : 0u, m2 = role == 2u ? ~0u : 0u, m3 = role == 3u ? ~0u : 0u;
    Fl r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = (a0.l[i] & m0) | (a1.l[i] & m1) | (a2.l[i] & m2) | (a3.l[i] & m3);
    return r;
}
// A Jacobian addition over a quad: its 16 multiplications are five levels of at most four independent ones, so every lane does
// ONE multiplication per level - its role's.  Five dependent multiplications instead of sixteen; the lanes agree on (p, q) in this
// order and all of them end with the same (x3, y3, z3).
__device__ __forceinline__ JacL jacl_add_quad(const JacL &p, const JacL &q, uint32_t role) {
    Fl m = fl_mul_r280(ped_by_role(role, p.z, q.z, p.z, p.z), ped_by_role(role, p.z, q.z, q.z, p.z));
    const Fl z1z1 = fl_quad<0>(m), z2z2 = fl_quad<1>(m), z1z2 = fl_quad<2>(m);
    m = fl_mul_r280(ped_by_role(role, p.x, q.x, q.z, p.z), ped_by_role(role, z2z2, z1z1, z2z2, z1z1));
    const Fl u1 = fl_quad<0>(m), u2 = fl_quad<1>(m), t1 = fl_quad<2>(m), t2 = fl_quad<3>(m);
    const Fl h = fn_sub(u2, u1);
    m = fl_mul_r280(ped_by_role(role, p.y, q.y, h, z1z2), ped_by_role(role, t1, t2, h, h));
    const Fl s1 = fl_quad<0>(m), s2 = fl_quad<1>(m), hh = fl_quad<2>(m), z3 = fl_quad<3>(m);
    const Fl rr = fn_sub(s2, s1);
    m = fl_mul_r280(ped_by_role(role, h, u1, rr, rr), ped_by_role(role, hh, hh, rr, rr));
    const Fl hhh = fl_quad<0>(m), v = fl_quad<1>(m), rr2 = fl_quad<2>(m);
    JacL r;
    r.x = fn_sub(fn_sub(rr2, hhh), fn_dbl(v));
    const Fl vx = fn_sub(v, r.x);
    m = fl_mul_r280(ped_by_role(role, rr, s1, rr, s1), ped_by_role(role, vx, hhh, vx, hhh));
    r.y = fn_sub(fl_quad<0>(m), fl_quad<1>(m));
    r.z = z3;
    // the exceptional cases, as jacl_add decides them (every lane of the quad sees the same p and q, so they agree)
    if (__builtin_expect(fn_is_zero(p.z) || fn_is_zero(q.z) || fn_is_zero(h), 0)) {
        if (fn_is_zero(p.z)) return q;
        if (fn_is_zero(q.z)) return p;
        if (fn_is_zero(rr)) return jacl_double_d<EcR280>(p);
        JacL o; o.x = fl_one_r280(); o.y = fl_one_r280(); o.z = fl_zero(); return o;
    }
    return r;
}

// safegcd (inv252.h) over a quad.  An outer iteration is 30 division steps on the low words - serial, every lane does them - and
// the update of the four rows d, e, f, g by the steps' matrix: 72 multiply-adds that are a third of the inversion on one lane.
// Here every lane keeps ONE pair of rows - roles 0, 1: (d, e); roles 2, 3: (f, g) - updates ONE row (role 0: d, 1: e, 2: f, 3: g)
// and reads its pair's other row back.  20 iterations = 600 steps cover every input (590 suffice, inv252.h); a lane whose g is
// zero idles through the rest: the quad permutes sit outside every branch.  x canonical (any integer < p) -> x^-1 mod p, 0 -> 0.
__device__ __forceinline__ Fp sg_inverse_quad(const Fp &x, uint32_t role) {
    const bool de = role < 2u, second = (role & 1u) != 0u;
    S30 A, B;                                               // (d, e) = (0, 1) or (f, g) = (p, x)
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bit = 30 * i, w = bit >> 5, sft = bit & 31;
        uint64_t two = x.v[w];
        if (w + 1 < 8) two |= (uint64_t)x.v[w + 1] << 32;
        const int32_t gi = (int32_t)((uint32_t)(two >> sft) & (uint32_t)SG_M30);
        A.v[i] = de ? 0 : sg_modulus(i);
        B.v[i] = de ? (i == 0 ? 1 : 0) : gi;
    }
    int32_t eta = -1;
    u32 busy = 1;
#pragma unroll 1
    for (int it = 0; it < 20; ++it) {
        // the low words of (f, g) live on roles 2 and 3
        const u32 f0 = ped_quad<2>((u32)A.v[0] | ((u32)A.v[1] << 30)), g0 = ped_quad<2>((u32)B.v[0] | ((u32)B.v[1] << 30));
        S30 row = second ? B : A;                           // an idle lane hands its row back unchanged
        if (busy) {
            int32_t t[4];
            eta = sg_divsteps_30_var(eta, f0, g0, t);
            const int64_t c0 = second ? t[2] : t[0], c1 = second ? t[3] : t[1];
            // rows of (d, e): + m p with m chosen to clear the low 30 bits and keep the row in (-2p, p) (sg_update_de); rows of
            // (f, g) divide exactly: m = 0
            int32_t md = 0;
            int64_t c = c0 * A.v[0] + c1 * B.v[0];
            if (de) {
                const int32_t sa = A.v[8] >> 31, sb = B.v[8] >> 31;
                md = ((int32_t)c0 & sa) + ((int32_t)c1 & sb);
                md -= (int32_t)(((uint32_t)c + (uint32_t)md) & (uint32_t)SG_M30);
                c += (int64_t)sg_modulus(0) * md;
            }
            c >>= 30;
#pragma unroll
            for (int i = 1; i < 9; ++i) {
                c += c0 * A.v[i] + c1 * B.v[i];
                if (sg_modulus(i)) c += (int64_t)sg_modulus(i) * md;
                row.v[i - 1] = (int32_t)c & SG_M30; c >>= 30;
            }
            row.v[8] = (int32_t)c;
        }
        int32_t nz = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            A.v[i] = (int32_t)ped_quad_perm<0xA0>((u32)row.v[i]);       // quad_perm [0, 0, 2, 2]: the pair's first row
            B.v[i] = (int32_t)ped_quad_perm<0xF5>((u32)row.v[i]);       // quad_perm [1, 1, 3, 3]: the pair's second row
            nz |= B.v[i];
        }
        busy = ped_quad<2>((u32)(nz != 0)) & busy;          // g (roles 2, 3) decides for the quad
    }
    // d from role 0, the sign of f from role 2
    S30 d;
#pragma unroll
    for (int i = 0; i < 9; ++i) d.v[i] = (int32_t)ped_quad<0>((u32)A.v[i]);
    const bool fneg = (int32_t)ped_quad<2>((u32)A.v[8]) < 0;
    return sg_finish(d, fneg);
}

// The same level in ONE launch, shaped for latency (the top dozen levels of every tree are a chain of launches that each wait
// for the one before: 32 lanes per hash, most of the chip idle):
//   - the table points are affine, so the first butterfly round is an affine + affine addition (4M + 2S against 12M + 4S);
//   - with fewer than 16 windows per input a lane is idle anyway: one of them carries the shift point P0 into the butterfly
//     instead of a seventh, dependent addition at the end;
//   - the four Jacobian rounds and the inversion of Z run over quads (above): 5 dependent multiplications per addition instead
//     of 16, a quarter of the inversion's matrix updates per lane;
//   - lane 0 of the hash writes the digest: no second launch, no round trip of (X, Z) through memory.
template <int W>
__global__ __launch_bounds__(64) void pedersen_pairs_small_kernel(const Aff *__restrict__ table, uint64_t per_input, Aff shift,
                                                                  const uint8_t *__restrict__ in, uint64_t count, uint8_t *__restrict__ out) {
    constexpr uint32_t span = (1u << W) - 1u, nwin = (PED_BITS + W - 1) / W;
    static_assert(nwin <= (uint32_t)PED_MAX_WINDOWS, "one lane per window: at most 16 windows per input");
    constexpr bool shift_in_lane = nwin < 16;
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t k = t >> 5;
    if (k >= count) return;                                 // whole 32-lane groups leave together
    const uint32_t sub = (uint32_t)t & 31u, e = sub >> 4, w = sub & 15u, role = sub & 3u;
    Fp c = be_bytes_to_canon(in + 64 * k + 32 * e);
    u32 d = 0;                                              // this lane's digit: window w of the scalar (lanes w >= nwin have none)
#pragma unroll 1
    for (uint32_t q = 0; q <= w && q < nwin; ++q) { const u32 dq = ped_next_digit<W>(c); if (q == w) d = dq; }
    AffL pt; pt.x = fl_one_r280(); pt.y = fl_one_r280();
    int inf = 1;
    if (d) { pt = load_affl(table + (size_t)e * per_input + (size_t)w * span + (d - 1)); inf = 0; }
    if (shift_in_lane && sub == 15u) { pt.x = fl_from_fp(shift.x); pt.y = fl_from_fp(shift.y); inf = 0; }
    JacL acc;
    {
        AffL o; o.x = fl_shfl_xor(pt.x, 1); o.y = fl_shfl_xor(pt.y, 1);
        const int oinf = __shfl_xor(inf, 1, 64);
        // both lanes of a pair add in the same order (the even lane's point first): the same representative on both
        const u32 up = (sub & 1u) ? ~0u : 0u;
        AffL a, b;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const u32 dx = (o.x.l[i] ^ pt.x.l[i]) & up, dy = (o.y.l[i] ^ pt.y.l[i]) & up;
            a.x.l[i] = pt.x.l[i] ^ dx; a.y.l[i] = pt.y.l[i] ^ dy;
            b.x.l[i] = o.x.l[i] ^ dx; b.y.l[i] = o.y.l[i] ^ dy;
        }
        const int di = (oinf ^ inf) & (int)up;
        acc = jacl_add_affs_d<EcR280>(a, (inf ^ di) != 0, b, (oinf ^ di) != 0);
    }
#pragma unroll 1
    for (uint32_t m = 2; m <= 16; m <<= 1) {
        JacL o; o.x = fl_shfl_xor(acc.x, (int)m); o.y = fl_shfl_xor(acc.y, (int)m); o.z = fl_shfl_xor(acc.z, (int)m);
        const u32 up = (sub & m) ? ~0u : 0u;                  // the lanes of a quad agree on the order: the lower half's sum first
        JacL a, b;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const u32 dx = (o.x.l[i] ^ acc.x.l[i]) & up, dy = (o.y.l[i] ^ acc.y.l[i]) & up, dz = (o.z.l[i] ^ acc.z.l[i]) & up;
            a.x.l[i] = acc.x.l[i] ^ dx; a.y.l[i] = acc.y.l[i] ^ dy; a.z.l[i] = acc.z.l[i] ^ dz;
            b.x.l[i] = o.x.l[i] ^ dx; b.y.l[i] = o.y.l[i] ^ dy; b.z.l[i] = o.z.l[i] ^ dz;
        }
        acc = jacl_add_quad(a, b, role);                     // (quad permutes inside: one call, every lane of the wave in it)
    }
    if (!shift_in_lane) {
        AffL sh; sh.x = fl_from_fp(shift.x); sh.y = fl_from_fp(shift.y);
        acc = jacl_add_aff_d<EcR280>(acc, sh);
    }
    // x = X / Z^2 on every lane of the hash (they hold the same sum); Z = 0 - a point at infinity, unreachable for a hash - inverts
    // to 0 and gives x = 0, as pedersen_finish_kernel
    // The sum is in the R280 domain: acc.z holds z 2^280, its plain inverse is z^-1 2^-280, and the 2^256-form products below want
    // zi = z^-1 2^244 - then X 2^280 * zi^2 comes out as (X / z^2) 2^256, the interchange form: one Montgomery product with 2^780
    Fp k780;
    k780.v[0] = 0xdf18d8e4u; k780.v[1] = 0x177d1406u; k780.v[2] = 0xfcc06cc7u; k780.v[3] = 0x5706677fu;
    k780.v[4] = 0x00018754u; k780.v[5] = 0xd84f8363u; k780.v[6] = 0xc0a6666bu; k780.v[7] = 0x05f79873u;
    const Fl zi = fl_from_fp(fp_mul(sg_inverse_quad(fl_to_fp(acc.z), role), k780));
    const Fp x = fl_to_fp(fn_mul(acc.x, fn_sqr(zi)));        // acc.x: normalised and < 2p from either addition
    if (sub == 0u) canon_to_be_bytes(fp_from_mont(x), out + 32 * k);
}

// ---- phase 2: x = X / ZZ (Jacobian build: X / Z^2), `chunk` hashes per lane (k = j * lanes + lane), one inversion per lane ----
// Montgomery's trick with the prefix products in tmp[2 count ..).  A point at infinity (Z = 0:
// unreachable for a hash, kept for totality) yields x = 0 as the per-hash formula did.
template <bool BYTES>
__global__ __launch_bounds__(64) void pedersen_finish_kernel(Fp *__restrict__ tmp, uint64_t count, uint64_t lanes,
                                                             Fp *__restrict__ out_felts, uint8_t *__restrict__ out_bytes) {
    const uint64_t c = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (c >= lanes) return;
    const Fp *X = tmp, *Z = tmp + count;
    Fp *P = tmp + 2 * count;
    Fl run = fl_one();
    uint64_t k = c;
    for (; k < count; k += lanes) {
        Fl z = fl_from_fp(load_felt(Z + k));
        if (fn_is_zero(z)) z = fl_one();
        store_felt(P + k, fl_pack(run));
        run = fn_mul(run, z);
    }
    // safegcd (inv252.h): ~3x less latency than the 251-squaring power - what a small level is bound by
    Fl inv = fl_from_fp(fp_inv_safegcd(fl_to_fp(run)));
    while (k > c) {
        k -= lanes;
        Fl z = fl_from_fp(load_felt(Z + k));
        const bool infinity = fn_is_zero(z);
        if (infinity) z = fl_one();
        const Fl zi = fn_mul(inv, fl_from_fp(load_felt(P + k)));
        inv = fn_mul(inv, z);
        Fp x = fl_to_fp(fn_mul(fl_from_fp(load_felt(X + k)), zi));     // (X 2^280) (ZZ^-1 2^232) 2^-256 = (X / ZZ) 2^256 (ped_acc_start's note)
        if (infinity) x = fp_zero();
        if (BYTES) canon_to_be_bytes(fp_from_mont(x), out_bytes + 32 * k);
        else store_felt(out_felts + k, x);
    }
}

static uint64_t finish_lanes(uint64_t count) {
    // hashes per inversion: 5 products per hash + ~130 products' worth of safegcd per chunk, against enough lanes to fill the chip
    // (SS_PED_FINISH_CHUNK: the chunk of levels of >= 2^18 hashes, tuning only)
    static const uint64_t big = [] { const char *e = getenv("SS_PED_FINISH_CHUNK"); const long v = e ? atol(e) : 0; return v >= 1 && v <= 64 ? (uint64_t)v : 8ull; }();
    const uint64_t chunk = count >= (1ull << 18) ? big : count >= (1ull << 16) ? 8 : count >= (1ull << 13) ? 4 : 1;
    return (count + chunk - 1) / chunk;
}
static hipError_t launch_finish(hipStream_t st, Fp *tmp, uint64_t count, Fp *out_felts, uint8_t *out_bytes) {
    const uint64_t lanes = finish_lanes(count);
    const dim3 grid((uint32_t)((lanes + 63) / 64)), block(64);
    if (out_bytes) hipLaunchKernelGGL(pedersen_finish_kernel<true>, grid, block, 0, st, tmp, count, lanes, (Fp *)nullptr, out_bytes);
    else hipLaunchKernelGGL(pedersen_finish_kernel<false>, grid, block, 0, st, tmp, count, lanes, out_felts, (uint8_t *)nullptr);
    return hipGetLastError();
}
// the window width is a template parameter of the accumulate kernels (digits by constant shifts, tables by constant strides)
#define PED_DISPATCH(W_, CALL)                                                 \
    switch (W_) {                                                              \
        case 16: { constexpr int W = 16; CALL; } break;                        \
        case 18: { constexpr int W = 18; CALL; } break;                        \
        case 20: { constexpr int W = 20; CALL; } break;                        \
        case 22: { constexpr int W = 22; CALL; } break;                        \
        case 26: { constexpr int W = 26; CALL; } break;                        \
        default: { constexpr int W = 24; CALL; } break;                        \
    }
static hipError_t launch_acc_felts(hipStream_t st, const PedersenTables *t, const PedFeltArgs &g, uint64_t count, Fp *tmp) {
    PED_DISPATCH(t->W, hipLaunchKernelGGL(pedersen_acc_felts_kernel<W>, dim3((uint32_t)((count + 63) / 64)), dim3(64), 0, st, t->d_table, t->per_input,
                                          t->shift, g, count, tmp));
    return hipGetLastError();
}

// tmp: PEDERSEN_TMP_FELTS_PER_HASH * n felts of device scratch (X, Z, prefix products, chained digest)
hipError_t launch_pedersen_felts(hipStream_t st, const PedersenTables *t, const Fp *a, const Fp *b,
                                 uint64_t n, Fp *out, Fp *tmp) {
    if (n == 0) return hipSuccess;
    PedFeltArgs g;
    g.a = a; g.a_stride = 1; g.a_const = fp_zero(); g.b = b; g.b_stride = 1; g.b_const = fp_zero();
    hipError_t e = launch_acc_felts(st, t, g, n, tmp);
    if (e != hipSuccess) return e;
    return launch_finish(st, tmp, n, out, nullptr);
}
hipError_t launch_pedersen_pairs(hipStream_t st, const PedersenTables *t, const uint8_t *in, uint64_t count,
                                 uint8_t *out, Fp *tmp) {
    if (count == 0) return hipSuccess;
    static const uint64_t small_max = getenv("SS_PED_SMALL_MAX") ? strtoull(getenv("SS_PED_SMALL_MAX"), nullptr, 10) : PED_SPLIT_MAX;   // A/B
    if (count <= small_max) {
        PED_DISPATCH(t->W, hipLaunchKernelGGL(pedersen_pairs_small_kernel<W>, dim3((uint32_t)((count * 32 + 63) / 64)), dim3(64), 0, st,
                                              t->d_table, t->per_input, t->shift, in, count, out));
        return hipGetLastError();
    }
    PED_DISPATCH(t->W, hipLaunchKernelGGL(pedersen_acc_pairs_kernel<W>, dim3((uint32_t)((count + 63) / 64)), dim3(64), 0, st, t->d_table,
                                          t->per_input, t->shift, in, count, tmp));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_finish(st, tmp, count, nullptr, out);
}
// PedersenHashFn::hash_elements([l0, l1]) = H(H(H(0, l0), l1), 2): three rounds over all pairs
hipError_t launch_pedersen_felt_pairs(hipStream_t st, const PedersenTables *t, const Fp *felts,
                                      uint64_t count, uint8_t *out, Fp *tmp) {
    if (count == 0) return hipSuccess;
    Fp *h = tmp + 3 * count;
    Fp two = fp_zero(); two.v[0] = 2;
    PedFeltArgs g;
    g.a = nullptr; g.a_stride = 0; g.a_const = fp_zero(); g.b = felts; g.b_stride = 2; g.b_const = fp_zero();
    hipError_t e = launch_acc_felts(st, t, g, count, tmp);
    if (e == hipSuccess) e = launch_finish(st, tmp, count, h, nullptr);
    g.a = h; g.a_stride = 1; g.b = felts + 1;
    if (e == hipSuccess) e = launch_acc_felts(st, t, g, count, tmp);
    if (e == hipSuccess) e = launch_finish(st, tmp, count, h, nullptr);
    g.b = nullptr; g.b_stride = 0; g.b_const = two;
    if (e == hipSuccess) e = launch_acc_felts(st, t, g, count, tmp);
    if (e == hipSuccess) e = launch_finish(st, tmp, count, nullptr, out);
    return e;
}

}  // namespace ss
