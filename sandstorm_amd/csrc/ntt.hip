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
// Between two phases of a pass in which every wave only touches ITS OWN chunk of the tile (see chunk_private below) nothing
// has to wait for another wave: the LDS executes one wave's instructions in order, so the wave's reads see its earlier writes;
// the fence only keeps the compiler from moving them across.  (The host emulation of the tests has no waves: a barrier there.)
#ifndef SS_NTT_WAVE_PRIVATE
#define SS_NTT_WAVE_PRIVATE 1      // A/B: 0 = a workgroup barrier between all phases (rounds 1-3)
#endif
#if defined(HIPEMU)
#define NTT_WAVE_SYNC() __syncthreads()
#elif defined(SS_NTT_ABL_NOBAR)
#define NTT_WAVE_SYNC() do {} while (0)
#else
#define NTT_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                             __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
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
#ifndef SS_NTT_PRELOAD_TW
#define SS_NTT_PRELOAD_TW 1        // the twiddles of a radix-2 / radix-4 register group loaded at its start (A/B: 0 = where they are used)
#endif
#ifndef SS_NTT_OCC_DIF
#define SS_NTT_OCC_DIF (SS_NTT_THREADS_DIF / 128)
#endif
#ifndef SS_NTT_TW_LDS
// 1: a strided CTI pass's 2^r - 1 tree nodes staged in LDS once per tile instead of a global load per butterfly group (VERDICT r4
// #2; built and MEASURED in round 5, profiles/r05_ntt_twiddles_lds.txt: batch LDE 44.0-44.4 ms from global memory, 44.2-44.6 ms from
// LDS - the loads it replaces were L1 hits of 16 lanes per address already, and the tile pays a barrier for the staging).  Off.
#define SS_NTT_TW_LDS 0
#endif
// The wave-private phases (NTT_WAVE_SYNC; chunk ownership = threadIdx.x >> 6, log_waves = log2(blockDim.x >> 6)) are written for
// 64-lane wavefronts and workgroups that are a power-of-two number of them: anything else would corrupt tiles silently.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HIPEMU) && !defined(__GFX9__)
#error "ntt.hip: the wave-private phases assume the 64-lane wavefronts of gfx9 (gfx950)"
#endif
static_assert(SS_NTT_THREADS % 64 == 0 && (SS_NTT_THREADS & (SS_NTT_THREADS - 1)) == 0, "SS_NTT_THREADS: a power-of-two number of 64-lane waves");
static_assert(SS_NTT_THREADS_DIF % 64 == 0 && (SS_NTT_THREADS_DIF & (SS_NTT_THREADS_DIF - 1)) == 0, "SS_NTT_THREADS_DIF: a power-of-two number of 64-lane waves");

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
// the tile layout follows the kernel's register-group size, also in its shorter tail groups.
// Every one of these maps is LINEAR over GF(2): slot(a ^ b) = slot(a) ^ slot(b).  The 2^G elements of a register group are
// ebase | (m << sh) with ebase zero where m << sh has bits, so their slots are slot(ebase) ^ slot(m << sh): one swizzle per
// lane and group, the second term wave-uniform (scalar ALU), ONE v_xor per element and plane kind (round 4; before it every
// element paid its own swizzle on load and on store: ~17 vector instructions per butterfly).
static constexpr int MODE_DIT = NTT_MODE_DIT, MODE_DIF = NTT_MODE_DIF, MODE_CTI = NTT_MODE_CTI;     // kernels.h
#define LAYOUT_G(mode) ((mode) == MODE_DIF ? SS_NTT_GMAX_DIF : SS_NTT_GMAX)
template <int G>
SS_HD int lds_slot(int e) {
    if (G == 2) { const int x = (e >> 4) & 3; return e ^ (x | (x << 2)); }
    return e ^ (((e >> 4) & 1) | (((e >> 5) & 1) << 1) | (((e >> 6) & 1) * 12));
}
template <int G>
SS_HD int lds_top_slot(int e) {
    if (G == 2) { const int y = (e >> 6) & 3; return e ^ (y * 21); }
    const int z = (e >> 6) & 7;
    return e ^ (z | (z << 3));
}

// LDS tile in the lazy form: limbs 0-3 and 4-7 in two 16-byte planes, limb 8 in a dword plane,
// XOR-swizzled (above).  2048 elements = 72 KiB: two workgroups per CU.  The planes sit at FIXED byte offsets of one static
// array (dword plane first so that all three fit the 16-bit offset field of a ds instruction): an access is the lane's byte
// address in a register plus an immediate, and the LDS address space is in the pointer type (a plain pointer lost it on some
// paths: flat_load_dword counts on vmcnt as well as lgkmcnt, so the exchange waited behind outstanding global traffic).
typedef u32 lds_u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned char __attribute__((address_space(3))) *lds_bytes_ptr;
typedef lds_u32x4_t __attribute__((address_space(3))) *lds_u32x4_ptr;
typedef u32 __attribute__((address_space(3))) *lds_u32_ptr;
static constexpr u32 LDS_TOP_OFF = 0, LDS_LO_OFF = 4u << LOG_TILE_MAX, LDS_HI_OFF = LDS_LO_OFF + (16u << LOG_TILE_MAX),
                     LDS_TILE_BYTES = LDS_HI_OFF + (16u << LOG_TILE_MAX);
// Behind the tile: the twiddles of a strided CTI pass.  Its 2^r rows are one subtree of the remainder tree for ALL of the tile's
// adjacent columns q, so a tile reads 2^r - 1 <= 127 distinct plan entries (level by level contiguous in the plan) where a DIT pass
// reads 2^r - 1 per q: staged once per tile (4.5 KB), read back with 16 lanes per address - instead of one 36-byte global load per
// butterfly group and lane with its 64-bit address arithmetic.  Local index of node jj of the pass's stage v: (2^(r-1-v) - 1) + jj.
static constexpr u32 LDS_TW_MAX = 128, LDS_TW_LO_OFF = LDS_TILE_BYTES, LDS_TW_HI_OFF = LDS_TW_LO_OFF + 16u * LDS_TW_MAX,
                     LDS_TW_TOP_OFF = LDS_TW_HI_OFF + 16u * LDS_TW_MAX, LDS_TOTAL_BYTES = LDS_TW_TOP_OFF + 4u * LDS_TW_MAX;
struct Tile {
    lds_bytes_ptr base;
    bool tw_staged;              // this pass reads its twiddles from the LDS copy (wave-uniform)
};
__device__ __forceinline__ Fl lds_tw_load(const Tile &t, u32 local) {
    const lds_u32x4_t a = *(lds_u32x4_ptr)(t.base + LDS_TW_LO_OFF + (local << 4)), b = *(lds_u32x4_ptr)(t.base + LDS_TW_HI_OFF + (local << 4));
    Fl r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = *(lds_u32_ptr)(t.base + LDS_TW_TOP_OFF + (local << 2));
    return r;
}
struct LdsAddr {          // byte offsets of one element inside the 16-byte planes / the dword plane
    u32 a16, a4;
};
template <int LG>
__device__ __forceinline__ LdsAddr lds_addr(u32 e) {
    LdsAddr a;
    a.a16 = (u32)lds_slot<LG>((int)e) << 4;
    a.a4 = (u32)lds_top_slot<LG>((int)e) << 2;
    return a;
}
__device__ __forceinline__ LdsAddr operator^(const LdsAddr &a, const LdsAddr &b) {
    LdsAddr r;
    r.a16 = a.a16 ^ b.a16; r.a4 = a.a4 ^ b.a4;
    return r;
}
__device__ __forceinline__ Fl lds_load(const Tile &t, const LdsAddr &at) {
#ifdef SS_NTT_ABL_NOLDS
    { Fl r; for (int i = 0; i < 9; ++i) r.l[i] = at.a16 * 2654435761u + i; r.l[8] &= 0xfffffffu; return r; }
#endif
    const lds_u32x4_t a = *(lds_u32x4_ptr)(t.base + LDS_LO_OFF + at.a16), b = *(lds_u32x4_ptr)(t.base + LDS_HI_OFF + at.a16);
    Fl r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = *(lds_u32_ptr)(t.base + LDS_TOP_OFF + at.a4);
    return r;
}
__device__ __forceinline__ void lds_store(const Tile &t, const LdsAddr &at, const Fl &x) {
#ifdef SS_NTT_ABL_NOLDS      // timing ablation only: no LDS traffic (keeps one conditional store alive)
    if (x.l[0] == 0xdeadbeefu && x.l[5] == 77u) *(lds_u32_ptr)(t.base + LDS_TOP_OFF + at.a4) = x.l[8];
    return;
#endif
    *(lds_u32x4_ptr)(t.base + LDS_LO_OFF + at.a16) = lds_u32x4_t{x.l[0], x.l[1], x.l[2], x.l[3]};
    *(lds_u32x4_ptr)(t.base + LDS_HI_OFF + at.a16) = lds_u32x4_t{x.l[4], x.l[5], x.l[6], x.l[7]};
    *(lds_u32_ptr)(t.base + LDS_TOP_OFF + at.a4) = x.l[8];
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
    uint32_t scale_pow2;  // inverse only: multiply outputs by 2^-scale_pow2 (0 = off)
    uint32_t contig;      // 1: s0 == 0, tile is a contiguous block
    uint32_t final_pass;  // 1: last pass of the transform, outputs are canonical (< p);
                          // 0: outputs are weakly reduced 256-bit images (< 2^252) for the next pass
    uint32_t ncols;       // columns of the batch (grid = tiles * ncols workgroups)
    uint32_t xcd_map;     // 1: the ncols workgroups of one tile are consecutive on ONE XCD (see the kernel)
    uint32_t log_tiles;   // tiles of this launch (2^log_tiles): all of the transform's, or those of a window (below)
    uint32_t tile0;       // first tile of the launch
    // A WINDOWED top pass (s0 + r == log_n; one transform spread over several GPUs, sharded.cpp): this rank holds, of every one
    // of the 2^r rows j, the elements q in [win0, win0 + 2^win_log) - stored row after row: element (j, q) of the transform,
    // global index (j << s0) | q, lives at (j << win_log) + (q - win0) of the column's buffer.  win_log1 = win_log + 1, 0 = off.
    uint32_t win_log1, win0;
    uint32_t cti_trivial; // CTI: the plan's offset is one, so zeta = 1 at the root and the left node of level 1 (radix_stage)
    // entries per plane of the plan: 2^log_n - 1, or fewer for the plan of a windowed pass - which only holds what the window's
    // butterflies read.  DIT / DIF: stage s0 + v (v < r) needs T_s[(jl << s0) | q] for jl < 2^v and the window's q; stored at
    // ((2^v - 1 + jl) << win_log) + (q - win0) (twiddle_kernel, `window` mode).  CTI: its top r levels are the whole plan of a
    // transform of 2^r points (T_level[J] = w^-(bitrev(J) n / 2^(level+1)) is the same for every n).
    uint32_t tw_entries;
};
// where element g of the transform lives in the column's buffer
__device__ __forceinline__ uint32_t mem_index(const PassParams &p, uint32_t g) {
    if (!p.win_log1) return g;
    return ((g >> p.s0) << (p.win_log1 - 1u)) + ((g & ((1u << p.s0) - 1u)) - p.win0);
}

// the plan's three planes (limbs 0-3, limbs 4-7, limb 8) of n - 1 entries each
struct TwPlanes {
    const uint4 *lo, *hi;
    const u32 *top;
};
__device__ __forceinline__ Fl tw_load(const TwPlanes &tw, uint32_t idx) {      // R280 form: canonical, normalised limbs
    const uint4 a4 = tw.lo[idx], b4 = tw.hi[idx];
    Fl t;
    t.l[0] = a4.x; t.l[1] = a4.y; t.l[2] = a4.z; t.l[3] = a4.w;
    t.l[4] = b4.x; t.l[5] = b4.y; t.l[6] = b4.z; t.l[7] = b4.w;
    t.l[8] = tw.top[idx];
    return t;
}

// One butterfly stage (local stage u + ST) on the 2^G register-resident elements, in the
// lazy 9 x 28-bit form (fl252.h): no carry chains, products are 81 in-place
// v_mad_u64_u32.  Reductions are issued only where a bound needs them.  Three networks (MODE):
//   DIT  forward, bit-reversed in -> natural out, Cooley-Tukey butterflies, stages ascending:
//        a' = a + b t,  b' = a - b t + 2p   (b t is a fresh product: normalised, < 2p); the twiddle of a butterfly is
//        T_s[k], k = the low s bits of its index (plan in natural order, the coset folded in).
//        Values grow by <= 2p and limbs by < 1.13 * 2^28 per stage, and fl_mul takes any
//        multiplicand with u32 limbs and value < 2^256, so a whole pass (<= 11 stages from
//        inputs < 2^252: values < 24p < 2^256, limbs < 13.4 * 2^28 < 2^32) runs WITHOUT any
//        reduction; the LDS tile holds the raw limbs and the pass's store reduces once.
//   DIF  inverse, natural in -> bit-reversed out, Gentleman-Sande butterflies, stages descending (the DIT network backwards):
//        a' = a + b,    b' = (a - b + C p) t  with C = 2, 8, 16 for the 1st, 2nd, 3rd
//        stage of a group (the sums' limbs double each stage).  Group inputs are < 2^252 and
//        normalised; products leave the group as they are (< 1.75p, normalised), only the
//        sums (the even outputs of the group's last stage) are weakly reduced.  Folds a coset's h^-j into its twiddles;
//        kept for the inverse transforms over a coset (composition, DEEP).
//   CTI  inverse over the subgroup itself (offset 1: every trace column), natural in -> bit-reversed out, with the DIT mode's
//        Cooley-Tukey butterflies (round 4): the remainder tree of x^n - 1.  The node of level m (m = log n - 1 - s, s the
//        stage = log2 of the pair distance) that holds the elements [J 2^(s+1), (J+1) 2^(s+1)) is the polynomial modulo
//        x^(2^(s+1)) - rho_J; its children are the remainders modulo x^(2^s) -+ zeta_J, zeta_J^2 = rho_J:
//        (lo, hi) -> (lo + zeta_J hi, lo - zeta_J hi) - ONE twiddle per node, zeta_J = w^-(bitrev_m(J) 2^s)
//        (plan in bit-reversed order per level, PLAN_BITREV), leaves = the values at w^-bitrev(J): n c_bitrev(J).
//        Same lazy discipline as DIT (no reduction inside a pass, no sums to reduce), the same 512-lane radix-4 groups instead
//        of the DIF mode's 256-lane radix-8 ones, three twiddle loads per group of four butterflies (the two lower-stage
//        nodes are neighbours in the plan), and the first two levels need one product instead of four: zeta = 1 at the
//        root and at the left node of level 1.
// plan index of the twiddle of pair `pr` of stage ST of a register group (see radix_stage)
template <int MODE, int G, int ST>
__device__ __forceinline__ uint32_t stage_tw_index(const PassParams &p, uint32_t u, uint32_t g0, int pr) {
    constexpr int STC = ST < G ? ST : 0;
    const uint32_t gsh = p.s0 + u;                  // element m of the group has global index g0 + (m << gsh)
    const uint32_t s = gsh + STC;                   // global stage
    if (MODE == MODE_CTI) {
        const uint32_t level = p.log_n - 1u - s;
        return ((1u << level) - 1u) + ((g0 >> (gsh + G)) << (G - 1 - STC)) + (uint32_t)(pr >> STC);
    }
    const uint32_t klow = g0 & ((1u << gsh) - 1u);
    const int m = ((pr >> STC) << (STC + 1)) | (pr & ((1 << STC) - 1));
    // the partner index bits of this pair move the row: + (m_low << u) rows, a window-length each in a window's plan
    if (p.win_log1)
        return ((((1u << (s - p.s0)) - 1u) + (klow >> p.s0)) << (p.win_log1 - 1u)) + ((klow & ((1u << p.s0) - 1u)) - p.win0)
               + ((uint32_t)(m & ((1 << STC) - 1)) << (u + p.win_log1 - 1u));
    return ((1u << s) - 1u) + klow + ((uint32_t)(m & ((1 << STC) - 1)) << gsh);
}

template <int MODE, int G, int ST, bool PRE>
__device__ __forceinline__ void radix_stage(Fl (&x)[1 << G], const TwPlanes &tw, const PassParams &p,
                                            uint32_t u, uint32_t g0, bool top_group, const Fl *pre) {
    if (ST >= G) return;
    constexpr bool DIF = MODE == MODE_DIF;
    constexpr int STC = ST < G ? ST : 0;
    const uint32_t gsh = p.s0 + u;
    const uint32_t s = gsh + STC;                   // global stage
    // position of this stage inside the group's execution order (DIF runs ST = G-1 .. 0)
    constexpr int ORD = DIF ? (G - 1 - STC) : STC;
    // CTI: the transform's first two levels (the top register group of its first pass) have zeta = 1 at the root and at the
    // left node of level 1; (wave-uniform branch)
    const bool lvl0 = MODE == MODE_CTI && top_group && p.cti_trivial && s + 1u == p.log_n;
    const bool lvl1 = MODE == MODE_CTI && top_group && p.cti_trivial && s + 2u == p.log_n;
#pragma unroll
    for (int pr = 0; pr < (1 << G) / 2; ++pr) {
        const int m = ((pr >> STC) << (STC + 1)) | (pr & ((1 << STC) - 1));
        const Fl a = x[m], b = x[m | (1 << STC)];
        if (MODE == MODE_CTI && (lvl0 || (lvl1 && (pr >> STC) == 0))) {
            // inputs of level 0 are canonical; of level 1's left node the level-0 sums (< 2p, limbs < 2^29)
            x[m] = fl_add(a, b);
            x[m | (1 << STC)] = lvl0 ? fl_sub_c<2, 1>(a, b) : fl_sub_c<8, 2>(a, b);
            continue;
        }
#ifdef SS_NTT_ABL_NOTW      // timing ablation only (wrong results): no twiddle loads
        Fl t = x[m]; t.l[0] += stage_tw_index<MODE, G, ST>(p, u, g0, pr);
#else
        const Fl t = PRE ? pre[pr] : tw_load(tw, stage_tw_index<MODE, G, ST>(p, u, g0, pr));
#endif
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

// global element index of tile-local element e (see the kernel's load phase); log_n <= 30: 32 bits
__device__ __forceinline__ uint32_t tile_gindex(const PassParams &p, uint32_t tile, uint32_t e) {
    if (p.contig) return (tile << p.log_tile) + e;
    const uint32_t log_t = p.log_tile - p.r;
    const uint32_t dq = e & ((1u << log_t) - 1u), j = e >> log_t;
    const uint32_t q = (tile << log_t) + dq;
    return ((q >> p.s0) << (p.s0 + p.r)) | (j << p.s0) | (q & ((1u << p.s0) - 1u));
}

// what a pass stores: the transform's canonical result (final pass) or a weakly reduced 256-bit image for the next pass
template <int MODE>
__device__ __forceinline__ Fp pass_output(const PassParams &p, const Fl &x, bool product) {
    Fp out;
#ifdef SS_NTT_ABL_NOCONV     // timing ablation only (wrong results): a pass leaves its limbs as they are - what a 9 x 28-bit format BETWEEN passes would
    if (!p.final_pass) {     // save at a pass's end (VERDICT r5 "next" 3a), without the ninth word's traffic such a format adds: an upper bound
        for (int i = 0; i < 8; ++i) out.v[i] = x.l[i];
        out.v[7] ^= x.l[8];
        return out;
    }
#endif
    if (p.final_pass) {                        // leaving the transform: canonical image (< p)
        out = product ? fp_reduce_once(fl_pack(x)) : fl_to_fp(x);
        if (MODE != MODE_DIT && p.scale_pow2) out = fp_div_pow2(out, p.scale_pow2);
    } else if (MODE == MODE_DIF) {             // next pass re-limbs any image < 2^256; its first differences want < 2^252
        out = product ? fl_pack(x) : fl_pack(fl_weak_reduce(x));
    } else {                                   // DIT / CTI: limbs < 14.2 * 2^28 after a pass (radix_stage); one carry chain, < 2^252 + 2^229
        out = fl_pack(fl_weak_reduce1(x));
    }
    return out;
}

// One radix-2^G register group on local stages [u, u+G).  from_global / to_global fuse the
// pass's HBM traffic into its first / last group (strided passes: lane <-> adjacent element, so
// the accesses stay coalesced), saving two LDS round trips and two barriers per pass.
template <int MODE, int G>
__device__ __forceinline__ void radix_group(const Tile &t, const TwPlanes &tw, const PassParams &p, uint32_t u,
                                            uint32_t tile, bool top_group, bool from_global, bool to_global,
                                            const Fp *__restrict__ src, Fp *__restrict__ dst) {
    constexpr bool DIF = MODE == MODE_DIF;
    constexpr int LG = LAYOUT_G(MODE);
    const uint32_t log_t = p.log_tile - p.r;            // log2(T)
    const uint32_t eshift = p.contig ? 0u : log_t;
    const uint32_t items = (1u << p.log_tile) >> G;
    const uint32_t sh = eshift + u;
    const uint32_t gsh = p.s0 + u;
    // wave-uniform halves of the elements' LDS addresses and global indices (see lds_slot)
    LdsAddr am[1 << G];
#pragma unroll
    for (int m = 0; m < (1 << G); ++m) am[m] = lds_addr<LG>((uint32_t)m << sh);
    // element m of a group: global index g0 + (m << gsh); in the buffer + m * step (a window stores row j = g >> s0 at j << win_log)
    const size_t dst_step = p.win_log1 ? (size_t)1 << (u + p.win_log1 - 1u) : (size_t)1 << gsh;
    const size_t src_step = p.win_log1 ? dst_step : ((size_t)1 << gsh) >> p.log_expand;
    // lane -> items: wave w takes the items [w * ipw, (w + 1) * ipw), 64 at a time - the elements of ONE chunk of the tile whenever
    // the group's stride allows (chunk_private); a group with fewer items than lanes leaves the upper lanes idle
    const uint32_t nthreads = blockDim.x, ipw = items / (nthreads >> 6);
    const bool whole = items >= nthreads;
    const uint32_t tau0 = whole ? (threadIdx.x >> 6) * ipw + (threadIdx.x & 63u) : threadIdx.x;
    const uint32_t tau_end = whole ? tau0 + ipw : (threadIdx.x < items ? tau0 + 1u : tau0);
    for (uint32_t tau = tau0; tau < tau_end; tau += 64u) {
        const uint32_t low = tau & ((1u << sh) - 1u);
        const uint32_t high = tau >> sh;
        const uint32_t ebase = (high << (sh + G)) | low;
        const uint32_t g0 = tile_gindex(p, tile, ebase);
        const LdsAddr ab = lds_addr<LG>(ebase);
        // the group's twiddles first (radix 2 and 4: at most three distinct ones, 27 registers): their loads are in flight while
        // the elements arrive and the first products run - issued where they are used, a wave waited a few hundred cycles on every
        // second product (the waves of a workgroup move in step from barrier to barrier, so nobody fills the gap)
        constexpr bool PRE = SS_NTT_PRELOAD_TW && G <= 2 && MODE != MODE_DIF;
        Fl tw0[PRE ? (1 << G) / 2 : 1], tw1[PRE ? (1 << G) / 2 : 1];
        if (PRE) {                                   // (the two trivial levels of a CTI transform's top group read theirs for nothing)
            if (MODE == MODE_CTI && t.tw_staged) {   // wave-uniform branch: the tile's staged tree nodes (see LDS_TW_*)
                const uint32_t jhi = (ebase >> log_t) >> (u + G);          // the group's rows are (jhi << G | m) << u | (row bits below u)
#pragma unroll
                for (int pr = 0; pr < (1 << G) / 2; ++pr) {
                    tw0[PRE ? pr : 0] = lds_tw_load(t, ((1u << (p.r - 1u - u)) - 1u) + (jhi << (G - 1)) + (uint32_t)pr);
                    if (G >= 2) tw1[PRE ? pr : 0] = lds_tw_load(t, ((1u << (p.r - 2u - u)) - 1u) + (jhi << (G >= 2 ? G - 2 : 0)) + (uint32_t)(pr >> 1));
                }
            } else {
#pragma unroll
                for (int pr = 0; pr < (1 << G) / 2; ++pr) {
                    tw0[PRE ? pr : 0] = tw_load(tw, stage_tw_index<MODE, G, 0>(p, u, g0, pr));
                    if (G >= 2) tw1[PRE ? pr : 0] = tw_load(tw, stage_tw_index<MODE, G, 1>(p, u, g0, pr));
                }
            }
        }
        Fl x[1 << G];
        if (from_global) {
            const Fp *sp = src + (mem_index(p, g0) >> p.log_expand);
#pragma unroll
            for (int m = 0; m < (1 << G); ++m) {
#ifdef SS_NTT_ABL_NOCONV     // ... and at its start: the 8 words taken as limbs (EVERY pass, the first one too: more than the format would save)
                const Fp raw = gload(sp + (size_t)m * src_step);
                for (int i = 0; i < 8; ++i) x[m].l[i] = raw.v[i];
                x[m].l[8] = raw.v[7] >> 28;
#else
                x[m] = fl_from_fp(gload(sp + (size_t)m * src_step));
#endif
            }
        } else {
#pragma unroll
            for (int m = 0; m < (1 << G); ++m) x[m] = lds_load(t, ab ^ am[m]);
        }
        if (MODE != MODE_DIT) {
            if (G >= 3) radix_stage<MODE, G, 2, false>(x, tw, p, u, g0, top_group, nullptr);
            if (G >= 2) radix_stage<MODE, G, 1, PRE>(x, tw, p, u, g0, top_group, PRE ? tw1 : nullptr);
            radix_stage<MODE, G, 0, PRE>(x, tw, p, u, g0, top_group, PRE ? tw0 : nullptr);
        } else {
            radix_stage<MODE, G, 0, PRE>(x, tw, p, u, g0, top_group, PRE ? tw0 : nullptr);
            if (G >= 2) radix_stage<MODE, G, 1, PRE>(x, tw, p, u, g0, top_group, PRE ? tw1 : nullptr);
            if (G >= 3) radix_stage<MODE, G, 2, false>(x, tw, p, u, g0, top_group, nullptr);
        }
        if (to_global) {
            Fp *dp = dst + mem_index(p, g0);
#pragma unroll
            for (int m = 0; m < (1 << G); ++m)     // DIF: odd outputs of the group's last stage are fresh products (normalised, < 2^252)
                gstore(dp + (size_t)m * dst_step, pass_output<MODE>(p, x[m], DIF && (m & 1)));
        } else {
#pragma unroll
            for (int m = 0; m < (1 << G); ++m)     // DIF: sums reduced to < 2^252, limbs < 2^28; DIT / CTI: raw lazy limbs (see radix_stage)
                lds_store(t, ab ^ am[m], DIF && !(m & 1) ? fl_weak_reduce(x[m]) : x[m]);
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(MODE == MODE_DIF ? SS_NTT_THREADS_DIF : SS_NTT_THREADS, MODE == MODE_DIF ? SS_NTT_OCC_DIF : SS_NTT_OCC)
void ntt_pass_kernel(ColPtrs cols, const Fp *__restrict__ tw_plan, PassParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[(SS_NTT_TW_LDS && MODE == MODE_CTI) ? LDS_TOTAL_BYTES : LDS_TILE_BYTES];
    constexpr int NTT_GMAX = MODE == MODE_DIF ? SS_NTT_GMAX_DIF : SS_NTT_GMAX;
    constexpr int LG = LAYOUT_G(MODE);
    const uint32_t tile_elems = 1u << p.log_tile;
    Tile t;
    t.base = (lds_bytes_ptr)smem;
    t.tw_staged = false;
    TwPlanes tw;
    {
        const uint64_t tw_total = p.tw_entries;
        tw.lo = reinterpret_cast<const uint4 *>(tw_plan);
        tw.hi = tw.lo + tw_total;
        tw.top = reinterpret_cast<const u32 *>(tw.lo + 2 * tw_total);
    }
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
        col = blockIdx.x >> p.log_tiles;
        tile = blockIdx.x & ((1u << p.log_tiles) - 1u);
    }
    tile += p.tile0;
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
    if (SS_NTT_TW_LDS && MODE == MODE_CTI && fuse && !p.win_log1 && p.log_tile - p.r <= p.s0 && p.r >= 2 && p.r <= 7 && p.u_first == 0) {
        // the tile's columns q share q >> s0 (2^(log_tile - r) <= 2^s0 adjacent ones): node jj of stage v is plan entry
        // (2^level - 1) + (qhi << (r - 1 - v)) + jj, level = log_n - 1 - s0 - v
        t.tw_staged = true;
        const uint32_t i = threadIdx.x;
        if (i < (1u << p.r) - 1u) {
            const uint32_t k = 31u - (uint32_t)__builtin_clz(i + 1u), v = p.r - 1u - k, jj = i + 1u - (1u << k);
            const uint32_t qhi = (tile << (p.log_tile - p.r)) >> p.s0, level = p.log_n - 1u - p.s0 - v;
            const uint32_t idx = ((1u << level) - 1u) + (qhi << k) + jj;
            const uint4 a4 = tw.lo[idx], b4 = tw.hi[idx];
            *(lds_u32x4_ptr)(t.base + LDS_TW_LO_OFF + (i << 4)) = lds_u32x4_t{a4.x, a4.y, a4.z, a4.w};
            *(lds_u32x4_ptr)(t.base + LDS_TW_HI_OFF + (i << 4)) = lds_u32x4_t{b4.x, b4.y, b4.z, b4.w};
            *(lds_u32_ptr)(t.base + LDS_TW_TOP_OFF + (i << 2)) = tw.top[idx];
        }
        __syncthreads();
    }
    // A tile is 2^log_chunk elements per wave.  A phase is chunk-private when wave w touches the elements [w << log_chunk,
    // (w + 1) << log_chunk) only: the contiguous pass's load and store phases by construction, a register group when its
    // elements' stride stays inside a chunk (sh + G <= log_chunk; radix_group deals a wave the items of one chunk).  Between two
    // such phases the waves do not wait for each other: a contiguous pass of 11 stages has 3 workgroup barriers instead of 7, a
    // strided one of 7 stages 2 instead of 3 - and the waves drift apart, so one's LDS round trip hides behind another's products.
    const uint32_t log_waves = 31u - (uint32_t)__builtin_clz(blockDim.x >> 6);
    const uint32_t log_chunk = p.log_tile - log_waves;
    const bool chunks = SS_NTT_WAVE_PRIVATE && tile_elems >= blockDim.x;
    auto group_private = [&](uint32_t u, uint32_t g) {
        return chunks && (tile_elems >> g) >= blockDim.x && (fuse ? p.log_tile - p.r : 0u) + u + g <= log_chunk;
    };
    bool prev_private = false;
    auto phase_sync = [&](bool cur_private) {          // between the previous phase and the one about to start
        if (prev_private && cur_private) NTT_WAVE_SYNC(); else NTT_SYNC();
        prev_private = cur_private;
    };
    const uint32_t x0 = chunks ? ((threadIdx.x >> 6) << log_chunk) + (threadIdx.x & 63u) : threadIdx.x;
    const uint32_t x_step = chunks ? 64u : blockDim.x;
    const uint32_t x_end = chunks ? (((threadIdx.x >> 6) + 1u) << log_chunk) : tile_elems;
    if (!fuse) {
        for (uint32_t x = x0; x < x_end; x += x_step)
            lds_store(t, lds_addr<LG>(x), fl_from_fp(gload(src + (mem_index(p, tile_gindex(p, tile, x)) >> p.log_expand))));
        prev_private = chunks;
    }

    if (MODE == MODE_DIT) {
        uint32_t u = p.u_first;
        bool first = true;
        while (u < p.r) {
            const uint32_t g = (p.r - u) >= (uint32_t)NTT_GMAX ? (uint32_t)NTT_GMAX : (p.r - u);
            const bool last = (u + g >= p.r);
            const bool fg = fuse && first, tg = fuse && last;
            if (fg) prev_private = group_private(u, g); else phase_sync(group_private(u, g));
            if (NTT_GMAX >= 3 && g == 3) radix_group<MODE, NTT_GMAX >= 3 ? 3 : 1>(t, tw, p, u, tile, false, fg, tg, src, dst);
            else if (g == 2) radix_group<MODE, 2>(t, tw, p, u, tile, false, fg, tg, src, dst);
            else radix_group<MODE, 1>(t, tw, p, u, tile, false, fg, tg, src, dst);
            u += g;
            first = false;
        }
    } else {
        uint32_t u = p.r;
        bool first = true;
        const bool top_pass = p.s0 + p.r == p.log_n;
        while (u > 0) {
            const uint32_t g = u >= (uint32_t)NTT_GMAX ? (uint32_t)NTT_GMAX : u;
            u -= g;
            const bool last = (u == 0);
            const bool fg = fuse && first, tg = fuse && last, top = top_pass && first;
            if (fg) prev_private = group_private(u, g); else phase_sync(group_private(u, g));
            if (NTT_GMAX >= 3 && g == 3) radix_group<MODE, NTT_GMAX >= 3 ? 3 : 1>(t, tw, p, u, tile, top, fg, tg, src, dst);
            else if (g == 2) radix_group<MODE, 2>(t, tw, p, u, tile, top, fg, tg, src, dst);
            else radix_group<MODE, 1>(t, tw, p, u, tile, top, fg, tg, src, dst);
            first = false;
        }
    }

    if (!fuse) {
        phase_sync(chunks);
        for (uint32_t x = x0; x < x_end; x += x_step)
            gstore(dst + mem_index(p, tile_gindex(p, tile, x)), pass_output<MODE>(p, lds_load(t, lds_addr<LG>(x)), false));
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
                               uint32_t log_n, int h_is_one, int bitrev_levels, uint32_t win_s0, uint32_t win_log, uint32_t win0,
                               uint64_t total) {
    // total = 2^log_n - 1 entries (stage s at 2^s - 1), or - win_s0 != 0 - the plan of a windowed top pass (PassParams.tw_entries):
    // entry ((2^v - 1 + jl) << win_log) + dq is T_(win_s0 + v)[(jl << win_s0) | (win0 + dq)]
    uint4 *plane_lo = reinterpret_cast<uint4 *>(tw), *plane_hi = plane_lo + total;
    u32 *plane_top = reinterpret_cast<u32 *>(plane_hi + total);
    for (uint64_t idx = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t s = 63u - (uint32_t)__clzll(idx + 1ull);
        uint64_t k = idx + 1ull - (1ull << s);
        if (win_s0) {
            const uint64_t t = idx >> win_log;
            const uint32_t v = 63u - (uint32_t)__clzll(t + 1ull);
            s = win_s0 + v;
            k = ((t + 1ull - (1ull << v)) << win_s0) | (win0 + (idx & ((1ull << win_log) - 1ull)));
        }
        // PLAN_BITREV (the CTI network, ntt_pass_kernel): entry J of level s is the natural plan's entry bitrev_s(J)
        if (bitrev_levels && s) k = __brevll(k) >> (64u - s);
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
hipError_t launch_ntt_pass(hipStream_t st, int mode, const ColPtrs &cols, uint32_t ncols, const Fp *tw,
                           uint32_t log_n, uint32_t s0, uint32_t r, uint32_t log_tile, uint32_t u_first,
                           uint32_t log_expand, uint32_t scale_pow2, bool final_pass, bool cti_trivial, const NttWindow *win,
                           uint64_t tw_entries) {
    PassParams p;
    p.tw_entries = tw_entries ? (uint32_t)tw_entries : (uint32_t)((1ull << log_n) - 1ull);
    p.final_pass = final_pass ? 1u : 0u;
    p.log_n = log_n; p.s0 = s0; p.r = r; p.log_tile = log_tile; p.u_first = u_first;
    p.log_expand = log_expand; p.scale_pow2 = scale_pow2; p.contig = (s0 == 0);
    p.cti_trivial = cti_trivial ? 1u : 0u;
    p.log_tiles = log_n - log_tile; p.tile0 = 0; p.win_log1 = 0; p.win0 = 0;
    if (win) {                                   // the tiles of the window only; a tile = 2^r rows x 2^(log_tile - r) adjacent q
        const uint32_t log_t = log_tile - r;
        if (s0 + r != log_n || s0 == 0 || log_expand || win->log_len < log_t || (win->first & ((1u << win->log_len) - 1u))) return hipErrorInvalidValue;
        p.log_tiles = win->log_len - log_t;
        p.tile0 = win->first >> log_t;
        p.win_log1 = win->log_len + 1u;
        p.win0 = win->first;
    }
    const uint32_t tiles = 1u << p.log_tiles;
    static const bool no_xcd_map = getenv("SS_NTT_NO_XCD_MAP") != nullptr;        // A/B switch for profiling
    p.ncols = ncols;
    p.xcd_map = (tiles >= 8 && ncols > 1 && !no_xcd_map) ? 1u : 0u;
    dim3 grid(tiles * ncols), block(mode == NTT_MODE_DIF ? SS_NTT_THREADS_DIF : SS_NTT_THREADS);
    if (mode == NTT_MODE_DIF) hipLaunchKernelGGL(ntt_pass_kernel<MODE_DIF>, grid, block, 0, st, cols, tw, p);
    else if (mode == NTT_MODE_CTI) hipLaunchKernelGGL(ntt_pass_kernel<MODE_CTI>, grid, block, 0, st, cols, tw, p);
    else hipLaunchKernelGGL(ntt_pass_kernel<MODE_DIT>, grid, block, 0, st, cols, tw, p);
    return hipGetLastError();
}

hipError_t launch_twiddles(hipStream_t st, Fp *tw, const Fp *pow_lo, const Fp *pow_hi, const Fp *hpow,
                           uint32_t log_n, bool h_is_one, bool bitrev_levels, const NttWindow *win, uint32_t win_stages) {
    const uint64_t total = win ? (((uint64_t)1 << win_stages) - 1ull) << win->log_len : (1ull << log_n) - 1ull;
    uint32_t blocks = (uint32_t)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(twiddle_kernel, dim3(blocks), dim3(256), 0, st, tw, pow_lo, pow_hi, hpow, log_n,
                       h_is_one ? 1 : 0, bitrev_levels ? 1 : 0, win ? log_n - win_stages : 0u, win ? win->log_len : 0u, win ? win->first : 0u, total);
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

// The tile is a static 72 KiB LDS array (above the 64 KiB default of a launch): nothing to opt into for static LDS; kept as
// the hook the context calls at creation.
hipError_t ntt_set_func_attributes() { return hipSuccess; }

}  // namespace ss
