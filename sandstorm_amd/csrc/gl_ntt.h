// One pass of the 64-bit field's transform as the device runs it - the tile's index arithmetic, the register groups with their
// butterflies and twiddle indexing - and the plan that splits a transform into passes.  csrc/goldilocks.hip wraps this in the kernel
// (tile in LDS, a barrier between groups) and csrc/capi.hip launches the plan's passes; tests/cpp/gl_ntt_host_test.cpp compiles the
// SAME text for the host (threadIdx / blockDim as globals, one "lane" at a time, groups in order) and tests/test_gl64_host.py holds
// every transform size, direction and blow-up to the oracle without a GPU.
#pragma once
#include <stdint.h>
#include "gl64.h"

namespace ss {

// ---- the plan: pass i runs global stages [s0, s0 + r).  The first pass is a contiguous tile of up to 2^log_tile_max elements;
// the others are strided tiles of 2^r rows x >= 32 adjacent elements (256-byte global runs), the remaining stages split evenly.
struct GlPass { uint32_t s0, r; };
static inline int gl_plan_passes_into(uint32_t log_n, uint32_t log_tile_max, GlPass out[8], uint32_t log_min_run = 5) {
    int np = 0;
    const uint32_t r0 = log_n < log_tile_max ? log_n : log_tile_max;
    out[np++] = GlPass{0, r0};
    uint32_t rem = log_n - r0;
    if (rem) {
        const uint32_t rmax = log_tile_max - log_min_run;      // rows of a strided tile: 2^log_min_run adjacent elements per global run
        const uint32_t k = (rem + rmax - 1) / rmax;
        uint32_t s0 = r0;
        for (uint32_t i = 0; i < k; ++i) {
            const uint32_t r = rem / (k - i) + ((rem % (k - i)) ? 1 : 0);
            out[np++] = GlPass{s0, r};
            s0 += r; rem -= r;
        }
    }
    return np;
}

struct GlPassParams {
    uint32_t log_n, s0, r, log_tile, u_first, log_expand, contig;
    uint64_t scale;              // DIF last pass: multiply outputs by this (1/n); 1 = off
};

// global element index of tile-local element e (ntt.hip tile_gindex: contiguous block, or 2^r rows x T adjacent elements)
__device__ __forceinline__ uint64_t gl_tile_gindex(const GlPassParams &p, uint32_t tile, uint32_t e) {
    if (p.contig) return ((uint64_t)tile << p.log_tile) + e;
    const uint32_t log_t = p.log_tile - p.r;
    const uint32_t dq = e & ((1u << log_t) - 1u), j = e >> log_t;
    const uint64_t q = ((uint64_t)tile << log_t) + dq;
    return ((q >> p.s0) << (p.s0 + p.r)) | ((uint64_t)j << p.s0) | (q & ((1ull << p.s0) - 1ull));
}

// LDS slot of tile element e: XOR swizzle (no padding).  A register group at shift sh makes 32 consecutive lanes walk
// either 32 consecutive elements (sh >= 5) or elements 16 apart (sh = 0, 4); e ^ ((e >> 4) & 31) maps both onto 32 distinct
// 8-byte slots modulo 32, i.e. all 64 banks once per ds_*_b64.
__device__ __forceinline__ uint32_t gl_slot(uint32_t e) { return e ^ ((e >> 4) & 31u); }

// One radix-2^G register group on local stages [u, u + G): each thread holds 2^G elements (element m at ebase + (m << sh))
// and runs G butterfly stages on them before the tile is touched again - ceil(13 / 4) = 4 LDS round trips per pass instead
// of 13.  The first / last group of a pass exchange with HBM directly (FG / TG).
// Twiddle plan as in the 252-bit path: T_s[k] at (2^s - 1) + k, k < 2^s, T_s[k] = h^(n / 2^(s+1)) * r^(k n / 2^(s+1)).
// Everything a group decides per ELEMENT is a template parameter: whether it reads HBM or the tile, writes HBM or the tile, and
// whether the tile is a contiguous block.  As run-time flags (this file's first form) the compiler addressed both memories through
// one generic pointer (flat_load / flat_store, which wait on both counters) and branched around every element's address
// computation; now a group's sixteen loads are sixteen global_load (or ds_read) instructions issued back to back, with its fifteen
// twiddles: the transforms of a 2^20-step proof went from 31.0 to 24.7 ms (profiles/r02_end2_*).
// Sums and differences of the decimation-in-time network stay lazy words (gl64.h) and are made canonical where the pass stores them.
template <bool DIF, int G, bool FG, bool TG, bool CONTIG>
__device__ __forceinline__ void gl_group(uint64_t *tile_lds, const uint64_t *__restrict__ tw, const GlPassParams &p, uint32_t u, uint32_t tile,
                                           const uint64_t *__restrict__ src, uint64_t *__restrict__ dst) {
    const uint32_t log_t = p.log_tile - p.r, eshift = CONTIG ? 0u : log_t;
    const uint32_t items = (1u << p.log_tile) >> G, sh = eshift + u;
    auto gindex = [&](uint32_t e) -> uint64_t {
        if (CONTIG) return ((uint64_t)tile << p.log_tile) + e;
        const uint32_t dq = e & ((1u << log_t) - 1u), j = e >> log_t;
        const uint64_t q = ((uint64_t)tile << log_t) + dq;
        return ((q >> p.s0) << (p.s0 + p.r)) | ((uint64_t)j << p.s0) | (q & ((1ull << p.s0) - 1ull));
    };
    for (uint32_t tau = threadIdx.x; tau < items; tau += blockDim.x) {
        const uint32_t low = tau & ((1u << sh) - 1u), high = tau >> sh;
        const uint32_t ebase = (high << (sh + G)) | low;
        uint32_t jbase, lbits;
        if (CONTIG) { jbase = ebase & ((1u << p.r) - 1u); lbits = 0; }
        else {
            jbase = ebase >> log_t;
            const uint32_t q = (tile << log_t) + (ebase & ((1u << log_t) - 1u));
            lbits = q & ((1u << p.s0) - 1u);
        }
        const uint32_t jlow = jbase & ((1u << u) - 1u);
        // the group's 2^G - 1 twiddles (2^ST distinct ones in stage ST) and its 2^G elements: all loads issued before the first butterfly
        uint64_t x[1 << G], wv[1 << G];
#pragma unroll
        for (int ST = 0; ST < G; ++ST) {
            const uint64_t *tws = tw + ((1ull << (p.s0 + u + ST)) - 1ull);
#pragma unroll
            for (int ml = 0; ml < (1 << ST); ++ml) wv[(1 << ST) - 1 + ml] = tws[((jlow + ((uint32_t)ml << u)) << p.s0) | lbits];
        }
#pragma unroll
        for (int m = 0; m < (1 << G); ++m) {
            const uint32_t e = ebase + ((uint32_t)m << sh);
            if (FG) x[m] = src[gindex(e) >> p.log_expand];
            else x[m] = tile_lds[gl_slot(e)];
        }
#pragma unroll
        for (int step = 0; step < G; ++step) {
            const int ST = DIF ? (G - 1 - step) : step;
#pragma unroll
            for (int pr = 0; pr < (1 << G) / 2; ++pr) {
                const int m = ((pr >> ST) << (ST + 1)) | (pr & ((1 << ST) - 1));
                const uint64_t w = wv[(1 << ST) - 1 + (m & ((1 << ST) - 1))];
                const uint64_t a = x[m], b = x[m | (1 << ST)];
                if (DIF) {
                    x[m] = gl_add(a, b);
                    x[m | (1 << ST)] = gl_mul(gl_sub_lazy(a, b), w);
                } else {
                    const uint64_t bt = gl_mul(b, w);
                    x[m] = gl_add_lazy(a, bt);
                    x[m | (1 << ST)] = gl_sub_lazy(a, bt);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < (1 << G); ++m) {
            const uint32_t e = ebase + ((uint32_t)m << sh);
            if (TG) {
                uint64_t v = DIF ? x[m] : gl_canon(x[m]);
                if (DIF && p.scale != 1ull) v = gl_mul(v, p.scale);
                dst[gindex(e)] = v;
            } else {
                tile_lds[gl_slot(e)] = x[m];
            }
        }
    }
}
template <bool DIF, int G, bool CONTIG>
__device__ __forceinline__ void gl_group_dispatch(uint64_t *tile_lds, const uint64_t *__restrict__ tw, const GlPassParams &p, uint32_t u, uint32_t tile,
                                                  bool fg, bool tg, const uint64_t *__restrict__ src, uint64_t *__restrict__ dst) {
    if (fg && tg) gl_group<DIF, G, true, true, CONTIG>(tile_lds, tw, p, u, tile, src, dst);
    else if (fg) gl_group<DIF, G, true, false, CONTIG>(tile_lds, tw, p, u, tile, src, dst);
    else if (tg) gl_group<DIF, G, false, true, CONTIG>(tile_lds, tw, p, u, tile, src, dst);
    else gl_group<DIF, G, false, false, CONTIG>(tile_lds, tw, p, u, tile, src, dst);
}

}  // namespace ss
