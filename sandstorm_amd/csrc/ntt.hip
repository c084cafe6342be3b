// ntt.hip — batched NTT / iNTT / LDE over Fp252 for gfx950.
//
// Replaces ministark's Matrix::interpolate / Matrix::evaluate (rows N1/N2 of
// SURVEY.md §8a; call sites src/lib.rs:17-26).  Convention (pinned by
// builtins/src/pedersen/periodic.rs:1183-1209): forward = evaluation at
// offset * w^k in natural order, w = 3^((p-1)/n).
//
// Structure
//   * forward  = radix-2 decimation-in-time   network, bit-reversed in -> natural out
//   * inverse  = radix-2 decimation-in-frequency network, natural in -> bit-reversed out
//     (the exact inverse network, stage by stage), so an LDE (iNTT then coset
//     NTT) never needs a transpose or a bit-reversal pass; the coset offset and
//     its powers are folded into the per-stage twiddle tables at plan time,
//     1/n is a 2^-k partial Montgomery step in the last inverse pass, and the
//     zero-padded half of the LDE input is never materialised (`log_expand`).
//   * log2(n) stages are grouped into passes; one pass = one kernel launch
//     that streams every column once (2 * n * 32 B of HBM traffic), keeps a
//     2048-element tile in LDS in the lazy 9-limb form, and runs up to 11
//     stages on it in register groups of 2 (forward) or 3 (inverse) stages per
//     LDS round trip; strided passes read/write HBM from their first/last group.
//   * butterflies multiply by twiddles kept in R280 limb form (fl252.h), and
//     reduce only where a bound needs it (radix_stage).
//   * pass 0 ("contig") covers the stages whose butterflies span <= 2048
//     adjacent elements: the tile is one contiguous 64 KiB block.  The other
//     passes ("strided") take 2^r rows x T adjacent elements; T >= 16 keeps
//     every global access a >= 512 B contiguous run.
//   * LDS layout: limbs 0-3 and 4-7 in two 16-byte planes, limb 8 in a dword
//     plane, XOR-swizzled so that every group shift is bank-conflict free
//     (lds_slot / lds_top_slot).
// What bounds it (profiles/r01_ntt_ablation.txt): VALU issue - ~300 instructions
// per butterfly, 185 of them the multiplication.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "fp252.h"
#include "fl252.h"
#include "kernels.h"

#ifdef SS_NTT_ABL_NOBAR      // timing ablation only: no workgroup barriers
#define NTT_SYNC() do {} while (0)
#else
#define NTT_SYNC() __syncthreads()
#endif

namespace ss {

#ifndef SS_NTT_LOG_TILE
#define SS_NTT_LOG_TILE 11
#endif
static constexpr int LOG_TILE_MAX = SS_NTT_LOG_TILE;          // 2048 elements
// per direction (forward DIT / inverse DIF): stages per register group (radix 2^G), threads per
// workgroup and waves per SIMD the kernel is compiled for
#ifndef SS_NTT_GMAX
#define SS_NTT_GMAX 2
#endif
#ifndef SS_NTT_THREADS
#define SS_NTT_THREADS 512
#endif
#ifndef SS_NTT_OCC
#define SS_NTT_OCC (SS_NTT_THREADS / 128)
#endif
#ifndef SS_NTT_GMAX_DIF
#define SS_NTT_GMAX_DIF 3          // fewer group boundaries = fewer weak reductions of the DIF sums
#endif
#ifndef SS_NTT_THREADS_DIF
#define SS_NTT_THREADS_DIF 256
#endif
#ifndef SS_NTT_OCC_DIF
#define SS_NTT_OCC_DIF (SS_NTT_THREADS_DIF / 128)
#endif

// Conflict-free LDS indexing by XOR swizzle (no padding).  A register group at shift sh makes the low
// lane bits walk element-index bits {sh+G.. } and/or {0..sh-1}; the bank index is a linear map of the
// element bits chosen so that, for every shift the group schedule produces (0, G, 2G, ... in the
// contiguous pass, >= 4 in strided passes), the bits the lanes walk map injectively:
//   16-byte planes (a ds_*_b128 serves 16 lanes per cycle: 4 index bits must be distinct)
//     G = 2:  s = e ^ (x | x << 2),            x = e[5:4]
//     G = 3:  s = e ^ (e[4] | e[5] << 1 | e[6] * 0b1100)
//   dword plane (64 lanes, 64 banks: 6 bits)
//     G = 2:  s = e ^ (y * 0b010101),          y = e[7:6]
//     G = 3:  s = e ^ (z | z << 3),            z = e[8:6]
// the tile layout follows the kernel's register-group size, also in its shorter tail groups
#define LAYOUT_G(dif) ((dif) ? SS_NTT_GMAX_DIF : SS_NTT_GMAX)
template <int G>
__device__ __forceinline__ int lds_slot(int e) {
    if (G == 2) { const int x = (e >> 4) & 3; return e ^ (x | (x << 2)); }
    return e ^ (((e >> 4) & 1) | (((e >> 5) & 1) << 1) | (((e >> 6) & 1) * 12));
}
template <int G>
__device__ __forceinline__ int lds_top_slot(int e) {
    if (G == 2) { const int y = (e >> 6) & 3; return e ^ (y * 21); }
    const int z = (e >> 6) & 7;
    return e ^ (z | (z << 3));
}

// LDS tile in the lazy form: limbs 0-3 and 4-7 in two 16-byte planes, limb 8 in a dword plane,
// XOR-swizzled (above).  2048 elements = 72 KiB: two workgroups per CU.
// Pointers carry the LDS address space explicitly: through a plain `uint4 *` in a struct the
// compiler lost it on some paths and emitted flat_load_dword for the top-limb plane (flat accesses
// count on vmcnt as well as lgkmcnt, so the exchange waited behind outstanding global traffic).
typedef u32 lds_u32x4_t __attribute__((ext_vector_type(4)));
typedef lds_u32x4_t __attribute__((address_space(3))) *lds_u32x4_ptr;
typedef u32 __attribute__((address_space(3))) *lds_u32_ptr;
struct Tile {
    lds_u32x4_ptr lo, hi;
    lds_u32_ptr top;
};
template <int LG>
__device__ __forceinline__ Fl lds_load(const Tile &t, int e) {
#ifdef SS_NTT_ABL_NOLDS
    { Fl r; for (int i = 0; i < 9; ++i) r.l[i] = (u32)e * 2654435761u + i; r.l[8] &= 0xfffffffu; return r; }
#endif
    const int s = lds_slot<LG>(e);
    const lds_u32x4_t a = t.lo[s], b = t.hi[s];
    Fl r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = t.top[lds_top_slot<LG>(e)];
    return r;
}
template <int LG>
__device__ __forceinline__ void lds_store(const Tile &t, int e, const Fl &x) {
#ifdef SS_NTT_ABL_NOLDS      // timing ablation only: no LDS traffic (keeps one conditional store alive)
    if (x.l[0] == 0xdeadbeefu && x.l[5] == 77u) t.top[lds_top_slot<LG>(e)] = x.l[8];
    return;
#endif
    const int s = lds_slot<LG>(e);
    t.lo[s] = lds_u32x4_t{x.l[0], x.l[1], x.l[2], x.l[3]};
    t.hi[s] = lds_u32x4_t{x.l[4], x.l[5], x.l[6], x.l[7]};
    t.top[lds_top_slot<LG>(e)] = x.l[8];
}
__device__ __forceinline__ Fp gload(const Fp *p) {
#ifdef SS_NTT_ABL_NOGL       // timing ablation only: no global loads
    { Fp r; for (int i = 0; i < 8; ++i) r.v[i] = (u32)(size_t)p * 2246822519u + i; r.v[7] &= 0x7ffffffu; return r; }
#endif
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    Fp r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__device__ __forceinline__ void gstore(Fp *p, const Fp &x) {
#ifdef SS_NTT_ABL_NOGL       // timing ablation only: (almost) no global stores
    if (!(x.v[0] == 0xdeadbeefu && x.v[3] == 77u)) return;
#endif
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    q[1] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
}

struct PassParams {
    uint32_t log_n;       // transform size
    uint32_t s0;          // first global stage of this pass
    uint32_t r;           // stages in this pass
    uint32_t log_tile;    // elements per workgroup tile (<= LOG_TILE_MAX)
    uint32_t u_first;     // first local stage actually executed (= log_expand in the expanding pass)
    uint32_t log_expand;  // source index = element index >> log_expand
    uint32_t scale_pow2;  // DIF only: multiply outputs by 2^-scale_pow2 (0 = off)
    uint32_t contig;      // 1: s0 == 0, tile is a contiguous block
    uint32_t final_pass;  // 1: last pass of the transform, outputs are canonical (< p);
                          // 0: outputs are weakly reduced 256-bit images (< 2^252) for the next pass
    uint32_t ncols;       // columns of the batch (grid = tiles * ncols workgroups)
    uint32_t xcd_map;     // 1: the ncols workgroups of one tile are consecutive on ONE XCD (see the kernel)
};

// One butterfly stage (local stage u + ST) on the 2^G register-resident elements, in the
// lazy 9 x 28-bit form (fl252.h): no carry chains, products are 81 in-place
// v_mad_u64_u32.  Reductions are issued only where a bound needs them:
//   DIT  a' = a + b t,  b' = a - b t + 2p   (b t is a fresh product: normalised, < 2p).
//        Values grow by <= 2p and limbs by < 1.13 * 2^28 per stage, and fl_mul takes any
//        multiplicand with u32 limbs and value < 2^256, so a whole pass (<= 11 stages from
//        inputs < 2^252: values < 24p < 2^256, limbs < 13.4 * 2^28 < 2^32) runs WITHOUT any
//        reduction; the LDS tile holds the raw limbs and the pass's store reduces once.
//   DIF  a' = a + b,    b' = (a - b + C p) t  with C = 2, 8, 16 for the 1st, 2nd, 3rd
//        stage of a group (the sums' limbs double each stage).  Group inputs are < 2^252 and
//        normalised; products leave the group as they are (< 1.75p, normalised), only the
//        sums (the even outputs of the group's last stage) are weakly reduced.
template <bool DIF, int G, int ST>
__device__ __forceinline__ void radix_stage(Fl (&x)[1 << G], const Fp *__restrict__ tw, const PassParams &p,
                                            uint32_t u, uint32_t jlow, uint32_t lbits) {
    if (ST >= G) return;
    const uint32_t s = p.s0 + u + ST;               // global stage
    const uint64_t tw_total = (1ull << p.log_n) - 1ull, tw_stage = (1ull << s) - 1ull;
    const uint4 *tw_lo = reinterpret_cast<const uint4 *>(tw) + tw_stage, *tw_hi = tw_lo + tw_total;
    const u32 *tw_top = reinterpret_cast<const u32 *>(reinterpret_cast<const uint4 *>(tw) + 2 * tw_total) + tw_stage;
    constexpr int STC = ST < G ? ST : 0;
    // position of this stage inside the group's execution order (DIF runs ST = G-1 .. 0)
    constexpr int ORD = DIF ? (G - 1 - STC) : STC;
#pragma unroll
    for (int pr = 0; pr < (1 << G) / 2; ++pr) {
        const int m = ((pr >> STC) << (STC + 1)) | (pr & ((1 << STC) - 1));
        const uint32_t k = ((jlow + ((uint32_t)(m & ((1 << STC) - 1)) << u)) << p.s0) | lbits;
#ifdef SS_NTT_ABL_NOTW      // timing ablation only (wrong results): no twiddle loads
        Fl t = x[m]; t.l[0] += k;
#else
        Fl t;                                      // R280 form: canonical, normalised limbs
        {
            const uint4 a4 = tw_lo[k], b4 = tw_hi[k];
            t.l[0] = a4.x; t.l[1] = a4.y; t.l[2] = a4.z; t.l[3] = a4.w;
            t.l[4] = b4.x; t.l[5] = b4.y; t.l[6] = b4.z; t.l[7] = b4.w;
            t.l[8] = tw_top[k];
        }
#endif
        const Fl a = x[m], b = x[m | (1 << STC)];
        if (DIF) {
            x[m] = fl_add(a, b);
            const Fl d = ORD == 0 ? fl_sub_c<2, 1>(a, b) : ORD == 1 ? fl_sub_c<8, 2>(a, b) : fl_sub_c<16, 4>(a, b);
#ifdef SS_NTT_ABL_NOMUL     // timing ablation only: no modular multiplication
            x[m | (1 << STC)] = fl_add(d, t);
#else
            x[m | (1 << STC)] = fl_mul_r280(d, t);
#endif
        } else {
#ifdef SS_NTT_ABL_NOMUL
            const Fl bt = fl_add(b, t);
#else
            const Fl bt = fl_mul_r280(b, t);
#endif
            x[m] = fl_add(a, bt);
            x[m | (1 << STC)] = fl_sub_c<2, 1>(a, bt);
        }
    }
}

// global element index of tile-local element e (see the kernel's load phase)
__device__ __forceinline__ uint64_t tile_gindex(const PassParams &p, uint32_t tile, uint32_t e) {
    if (p.contig) return ((uint64_t)tile << p.log_tile) + e;
    const uint32_t log_t = p.log_tile - p.r;
    const uint32_t dq = e & ((1u << log_t) - 1u), j = e >> log_t;
    const uint64_t q = ((uint64_t)tile << log_t) + dq;
    return ((q >> p.s0) << (p.s0 + p.r)) | ((uint64_t)j << p.s0) | (q & ((1ull << p.s0) - 1ull));
}

// One radix-2^G register group on local stages [u, u+G).  from_global / to_global fuse the
// pass's HBM traffic into its first / last group (strided passes: lane <-> adjacent element, so
// the accesses stay coalesced), saving two LDS round trips and two barriers per pass.
template <bool DIF, int G>
__device__ __forceinline__ void radix_group(const Tile &t, const Fp *__restrict__ tw, const PassParams &p, uint32_t u,
                                            uint32_t tile, bool last_group, bool from_global, bool to_global,
                                            const Fp *__restrict__ src, Fp *__restrict__ dst) {
    const uint32_t log_t = p.log_tile - p.r;            // log2(T)
    const uint32_t eshift = p.contig ? 0u : log_t;
    const uint32_t items = (1u << p.log_tile) >> G;
    const uint32_t sh = eshift + u;
    for (uint32_t tau = threadIdx.x; tau < items; tau += blockDim.x) {
        const uint32_t low = tau & ((1u << sh) - 1u);
        const uint32_t high = tau >> sh;
        const uint32_t ebase = (high << (sh + G)) | low;
        // J of element 0 and the low global bits L
        uint32_t jbase, lbits;
        if (p.contig) {
            jbase = ebase & ((1u << p.r) - 1u);
            lbits = 0;
        } else {
            jbase = ebase >> log_t;
            const uint32_t q = (tile << log_t) + (ebase & ((1u << log_t) - 1u));
            lbits = q & ((1u << p.s0) - 1u);
        }
        const uint32_t jlow = jbase & ((1u << u) - 1u);
        Fl x[1 << G];
#pragma unroll
        for (int m = 0; m < (1 << G); ++m) {
            const uint32_t e = ebase + ((uint32_t)m << sh);
            if (from_global) x[m] = fl_from_fp(gload(src + (tile_gindex(p, tile, e) >> p.log_expand)));
            else x[m] = lds_load<LAYOUT_G(DIF)>(t, e);
        }
        if (DIF) {
            if (G >= 3) radix_stage<DIF, G, 2>(x, tw, p, u, jlow, lbits);
            if (G >= 2) radix_stage<DIF, G, 1>(x, tw, p, u, jlow, lbits);
            radix_stage<DIF, G, 0>(x, tw, p, u, jlow, lbits);
        } else {
            radix_stage<DIF, G, 0>(x, tw, p, u, jlow, lbits);
            if (G >= 2) radix_stage<DIF, G, 1>(x, tw, p, u, jlow, lbits);
            if (G >= 3) radix_stage<DIF, G, 2>(x, tw, p, u, jlow, lbits);
        }
#pragma unroll
        for (int m = 0; m < (1 << G); ++m) {
            const uint32_t e = ebase + ((uint32_t)m << sh);
            // DIF: odd outputs of the group's last stage are fresh products (normalised, < 2^252)
            const bool product = DIF && (m & 1);
            if (to_global) {
                Fp out;
                if (p.final_pass) {                        // leaving the transform: canonical image (< p)
                    out = product ? fp_reduce_once(fl_pack(x[m])) : fl_to_fp(x[m]);
                    if (DIF && p.scale_pow2) out = fp_div_pow2(out, p.scale_pow2);
                } else {                                   // next pass re-limbs any image < 2^256
                    out = product ? fl_pack(x[m]) : fl_pack(fl_weak_reduce(x[m]));
                }
                gstore(dst + tile_gindex(p, tile, e), out);
            } else if (DIF) {
                lds_store<LAYOUT_G(DIF)>(t, e, product ? x[m] : fl_weak_reduce(x[m]));     // < 2^252, limbs < 2^28
            } else {
                lds_store<LAYOUT_G(DIF)>(t, e, x[m]);                                      // raw lazy limbs (see radix_stage)
            }
        }
    }
}

template <bool DIF, int G>
__device__ __forceinline__ void run_group(const Tile &t, const Fp *tw, const PassParams &p, uint32_t u, uint32_t tile,
                                          bool last, bool fg, bool tg, const Fp *src, Fp *dst) {
    radix_group<DIF, G>(t, tw, p, u, tile, last, fg, tg, src, dst);
}

template <bool DIF>
__global__ __launch_bounds__(DIF ? SS_NTT_THREADS_DIF : SS_NTT_THREADS, DIF ? SS_NTT_OCC_DIF : SS_NTT_OCC) void ntt_pass_kernel(ColPtrs cols, const Fp *__restrict__ tw,
                                                               PassParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NTT_GMAX = DIF ? SS_NTT_GMAX_DIF : SS_NTT_GMAX;
    const uint32_t tile_elems = 1u << p.log_tile;
    const uint32_t slots = tile_elems;
    Tile t;
    t.lo = (lds_u32x4_ptr)smem;
    t.hi = t.lo + slots;
    t.top = (lds_u32_ptr)(t.hi + slots);
    // Workgroup -> (tile, column).  A strided pass reads one 36-byte twiddle per butterfly and stage, indexed by the
    // tile's low bits: the same table slice for every column of that tile.  Workgroup b runs on XCD b % 8 (observed;
    // speed only), so the columns of a tile are made 8 ids apart and adjacent in time: they share the slice in that
    // XCD's L2 instead of fetching it once per column through the fabric.
    uint32_t tile, col;
    if (p.xcd_map) {
        const uint32_t per = 8u * p.ncols, grp = blockIdx.x / per, rem = blockIdx.x - grp * per;
        col = rem >> 3;
        tile = (grp << 3) | (rem & 7u);
    } else {
        const uint32_t tiles = 1u << (p.log_n - p.log_tile);
        col = blockIdx.x >> (p.log_n - p.log_tile);
        tile = blockIdx.x & (tiles - 1u);
    }
    // select this block's column with scalar compares: a dynamically indexed by-value
    // kernarg struct would be copied to scratch
    const void *src_v = cols.src[0];
    void *dst_v = cols.dst[0];
#pragma unroll
    for (int c = 1; c < MAX_COLS; ++c)
        if (col == (unsigned)c) { src_v = cols.src[c]; dst_v = cols.dst[c]; }
    const Fp *__restrict__ src = reinterpret_cast<const Fp *>(src_v);
    Fp *__restrict__ dst = reinterpret_cast<Fp *>(dst_v);

    // Strided passes read/write HBM from the first/last register group directly; the
    // contiguous pass (tile = one 64 KiB block, first group of stride 1) stages through LDS.
    const bool fuse = !p.contig;
    if (!fuse) {
        for (uint32_t x = threadIdx.x; x < tile_elems; x += blockDim.x)
            lds_store<LAYOUT_G(DIF)>(t, x, fl_from_fp(gload(src + (tile_gindex(p, tile, x) >> p.log_expand))));
        NTT_SYNC();
    }

    if (!DIF) {
        uint32_t u = p.u_first;
        bool first = true;
        while (u < p.r) {
            const uint32_t g = (p.r - u) >= (uint32_t)NTT_GMAX ? (uint32_t)NTT_GMAX : (p.r - u);
            const bool last = (u + g >= p.r);
            const bool fg = fuse && first, tg = fuse && last;
            if (NTT_GMAX >= 3 && g == 3) run_group<false, NTT_GMAX >= 3 ? 3 : 1>(t, tw, p, u, tile, last, fg, tg, src, dst);
            else if (g == 2) run_group<false, 2>(t, tw, p, u, tile, last, fg, tg, src, dst);
            else run_group<false, 1>(t, tw, p, u, tile, last, fg, tg, src, dst);
            u += g;
            first = false;
            if (!last || !fuse) NTT_SYNC();
        }
    } else {
        uint32_t u = p.r;
        bool first = true;
        while (u > 0) {
            const uint32_t g = u >= (uint32_t)NTT_GMAX ? (uint32_t)NTT_GMAX : u;
            u -= g;
            const bool last = (u == 0);
            const bool fg = fuse && first, tg = fuse && last;
            if (NTT_GMAX >= 3 && g == 3) run_group<true, NTT_GMAX >= 3 ? 3 : 1>(t, tw, p, u, tile, last, fg, tg, src, dst);
            else if (g == 2) run_group<true, 2>(t, tw, p, u, tile, last, fg, tg, src, dst);
            else run_group<true, 1>(t, tw, p, u, tile, last, fg, tg, src, dst);
            first = false;
            if (!last || !fuse) NTT_SYNC();
        }
    }

    if (!fuse) {
        for (uint32_t x = threadIdx.x; x < tile_elems; x += blockDim.x) {
            Fp out;
            if (p.final_pass) {
                out = fl_to_fp(lds_load<LAYOUT_G(DIF)>(t, x));
                if (DIF && p.scale_pow2) out = fp_div_pow2(out, p.scale_pow2);
            } else {
                out = fl_pack(fl_weak_reduce(lds_load<LAYOUT_G(DIF)>(t, x)));
            }
            gstore(dst + tile_gindex(p, tile, x), out);
        }
    }
}

// ---------------------------------------------------------------- twiddles
// T_s[k] = h^(n / 2^(s+1)) * (r^(n / 2^(s+1)))^k,  k < 2^s, stored at (2^s - 1) + k.
// r^e comes from two host-computed tables: pow_lo[e & 4095] * pow_hi[e >> 12].
// The plan is stored in "R280 form" (fl252.h: t * 2^280 mod p as nine 28-bit limbs) in three planes
// (limbs 0-3, limbs 4-7 as 16-byte vectors, limb 8 as a dword; NTT_PLAN_ENTRY_BYTES = 36 per entry): the
// butterfly's fl_mul_r280 then needs neither a re-limbing of the twiddle nor the 4-bit tail of the
// 2^256 reduction (~185 instead of 223 + 12 VALU instructions per butterfly multiplication).
__global__ void twiddle_kernel(Fp *__restrict__ tw, const Fp *__restrict__ pow_lo,
                               const Fp *__restrict__ pow_hi, const Fp *__restrict__ hpow,
                               uint32_t log_n, int h_is_one) {
    const uint64_t total = (1ull << log_n) - 1ull;
    uint4 *plane_lo = reinterpret_cast<uint4 *>(tw), *plane_hi = plane_lo + total;
    u32 *plane_top = reinterpret_cast<u32 *>(plane_hi + total);
    for (uint64_t idx = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t s = 63u - (uint32_t)__clzll(idx + 1ull);
        const uint64_t k = idx + 1ull - (1ull << s);
        const uint64_t e = k << (log_n - 1u - s);          // exponent of the n-th root, < n/2
        Fp t = fp_mul(gload(pow_lo + (e & 4095ull)), gload(pow_hi + (e >> 12)));
        if (!h_is_one) t = fp_mul(t, gload(hpow + s));
        const Fl l = fl_to_r280(t);
        plane_lo[idx] = make_uint4(l.l[0], l.l[1], l.l[2], l.l[3]);
        plane_hi[idx] = make_uint4(l.l[4], l.l[5], l.l[6], l.l[7]);
        plane_top[idx] = l.l[8];
    }
}

// ------------------------------------------------------- bit-reversal swap
__global__ void bitrev_kernel(Fp *__restrict__ a, uint32_t log_n) {
    const uint64_t n = 1ull << log_n;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t j = __brevll(i) >> (64u - log_n);
        if (i < j) {
            Fp x = gload(a + i), y = gload(a + j);
            gstore(a + i, y);
            gstore(a + j, x);
        }
    }
}

// ------------------------------------------------------------- mul bench
__global__ void mul_bench_kernel(const Fp *__restrict__ a, const Fp *__restrict__ b, Fp *__restrict__ out,
                                 uint64_t n, uint32_t reps) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        Fp x = gload(a + i);
        const Fp y = gload(b + i);
        for (uint32_t k = 0; k < reps; ++k) x = fp_mul(x, y);
        gstore(out + i, x);
    }
}

// ------------------------------------------------------------ host launch
static inline size_t pass_lds_bytes(uint32_t log_tile) {
    size_t e = (size_t)1 << log_tile;
    return 2 * e * sizeof(uint4) + e * sizeof(u32);
}

hipError_t launch_ntt_pass(hipStream_t st, bool dif, const ColPtrs &cols, uint32_t ncols, const Fp *tw,
                           uint32_t log_n, uint32_t s0, uint32_t r, uint32_t log_tile, uint32_t u_first,
                           uint32_t log_expand, uint32_t scale_pow2, bool final_pass) {
    PassParams p;
    p.final_pass = final_pass ? 1u : 0u;
    p.log_n = log_n; p.s0 = s0; p.r = r; p.log_tile = log_tile; p.u_first = u_first;
    p.log_expand = log_expand; p.scale_pow2 = scale_pow2; p.contig = (s0 == 0);
    const uint32_t tiles = 1u << (log_n - log_tile);
    static const bool no_xcd_map = getenv("SS_NTT_NO_XCD_MAP") != nullptr;        // A/B switch for profiling
    p.ncols = ncols;
    p.xcd_map = (tiles >= 8 && ncols > 1 && !no_xcd_map) ? 1u : 0u;
    dim3 grid(tiles * ncols), block(dif ? SS_NTT_THREADS_DIF : SS_NTT_THREADS);
    const size_t lds = pass_lds_bytes(log_tile);
    if (dif) hipLaunchKernelGGL(ntt_pass_kernel<true>, grid, block, lds, st, cols, tw, p);
    else hipLaunchKernelGGL(ntt_pass_kernel<false>, grid, block, lds, st, cols, tw, p);
    return hipGetLastError();
}

hipError_t launch_twiddles(hipStream_t st, Fp *tw, const Fp *pow_lo, const Fp *pow_hi, const Fp *hpow,
                           uint32_t log_n, bool h_is_one) {
    const uint64_t total = (1ull << log_n) - 1ull;
    uint32_t blocks = (uint32_t)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(twiddle_kernel, dim3(blocks), dim3(256), 0, st, tw, pow_lo, pow_hi, hpow, log_n,
                       h_is_one ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_bitrev(hipStream_t st, Fp *a, uint32_t log_n) {
    const uint64_t n = 1ull << log_n;
    uint32_t blocks = (uint32_t)((n + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(bitrev_kernel, dim3(blocks), dim3(256), 0, st, a, log_n);
    return hipGetLastError();
}

hipError_t launch_mul_bench(hipStream_t st, const Fp *a, const Fp *b, Fp *out, uint64_t n, uint32_t reps) {
    uint32_t blocks = (uint32_t)((n + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(mul_bench_kernel, dim3(blocks), dim3(256), 0, st, a, b, out, n, reps);
    return hipGetLastError();
}

int ntt_log_tile_max() { return LOG_TILE_MAX; }

// tiles above 64 KiB of dynamic LDS need the per-function opt-in
hipError_t ntt_set_func_attributes() {
    const int bytes = (int)pass_lds_bytes(LOG_TILE_MAX);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt_pass_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt_pass_kernel<false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

}  // namespace ss
