// Host build of the 64-bit field's transform pass (sandstorm_amd/csrc/gl_ntt.h: the text csrc/goldilocks.hip compiles for the device -
// tile index arithmetic, register groups, twiddle indexing, lazy butterflies - and the pass plan csrc/capi.hip launches).
// One "lane" at a time, a pass's groups in order (what the kernel's barrier guarantees), tiles one after the other.
// tests/test_gl64_host.py holds every size, direction and blow-up to the oracle.
// usage: gl_ntt_host_test <in> <out>; in = u64 {log_n, inverse, log_expand, scale, log_tile_max, n_src}, src[n_src], tw[2^log_n - 1]; out = dst[2^log_n]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define __device__
#define __forceinline__ inline
struct Dim3 { unsigned x, y, z; };
static Dim3 threadIdx, blockDim;
#include "gl_ntt.h"
using namespace ss;

// goldilocks.hip gl_ntt_pass_kernel<DIF, CONTIG>, one tile: the groups in order, every lane of a group before the next group
template <bool DIF, bool CONTIG>
static void run_tile(uint64_t *lds, const uint64_t *tw, const GlPassParams &p, uint32_t tile, const uint64_t *src, uint64_t *dst) {
    const uint32_t first = DIF ? 0u : p.u_first, total = p.r - first;
    uint32_t done = 0;
    while (done < total) {
        const uint32_t g = (total - done) >= 4 ? 4u : (total - done);
        const uint32_t u = DIF ? (p.r - done - g) : (first + done);
        const bool fg = done == 0, tg = done + g == total;
        for (threadIdx.x = 0; threadIdx.x < blockDim.x; ++threadIdx.x) {
            switch (g) {
            case 4: gl_group_dispatch<DIF, 4, CONTIG>(lds, tw, p, u, tile, fg, tg, src, dst); break;
            case 3: gl_group_dispatch<DIF, 3, CONTIG>(lds, tw, p, u, tile, fg, tg, src, dst); break;
            case 2: gl_group_dispatch<DIF, 2, CONTIG>(lds, tw, p, u, tile, fg, tg, src, dst); break;
            default: gl_group_dispatch<DIF, 1, CONTIG>(lds, tw, p, u, tile, fg, tg, src, dst); break;
            }
        }
        done += g;
    }
    if (total == 0)
        for (uint32_t e = 0; e < (1u << p.log_tile); ++e) dst[gl_tile_gindex(p, tile, e)] = src[gl_tile_gindex(p, tile, e) >> p.log_expand];
}
// goldilocks.hip launch_gl_ntt_pass
static void run_pass(bool dif, const uint64_t *src, uint64_t *dst, const uint64_t *tw, uint32_t log_n, uint32_t s0, uint32_t r, uint32_t log_tile,
                     uint32_t u_first, uint32_t log_expand, uint64_t scale) {
    GlPassParams p;
    p.log_n = log_n; p.s0 = s0; p.r = r; p.log_tile = log_tile; p.u_first = u_first; p.log_expand = log_expand; p.contig = (s0 == 0); p.scale = scale;
    std::vector<uint64_t> lds((size_t)1 << log_tile, 0xDEADBEEFDEADBEEFull);
    for (uint32_t tile = 0; tile < (1u << (log_n - log_tile)); ++tile) {
        if (p.contig) { if (dif) run_tile<true, true>(lds.data(), tw, p, tile, src, dst); else run_tile<false, true>(lds.data(), tw, p, tile, src, dst); }
        else { if (dif) run_tile<true, false>(lds.data(), tw, p, tile, src, dst); else run_tile<false, false>(lds.data(), tw, p, tile, src, dst); }
    }
}

template <class T>
static void rd(FILE *f, std::vector<T> &v, size_t n) { v.resize(n); if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } }

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<uint64_t> hdr, src, tw;
    rd(f, hdr, 6);
    const uint32_t log_n = (uint32_t)hdr[0], log_expand = (uint32_t)hdr[2], lt = (uint32_t)hdr[4];
    const bool inverse = hdr[1] != 0;
    rd(f, src, hdr[5]); rd(f, tw, ((size_t)1 << log_n) - 1);
    fclose(f);
    blockDim = {256, 1, 1};
    std::vector<uint64_t> dst((size_t)1 << log_n, 0);
    GlPass passes[8];
    const int np = gl_plan_passes_into(log_n, lt, passes);
    const uint32_t log_tile = log_n < lt ? log_n : lt;
    if (!inverse) {                                            // capi.hip gl_run_forward
        if (log_expand > passes[0].r) return 3;
        for (int i = 0; i < np; ++i)
            run_pass(false, i == 0 ? src.data() : dst.data(), dst.data(), tw.data(), log_n, passes[i].s0, passes[i].r, log_tile, i == 0 ? log_expand : 0,
                     i == 0 ? log_expand : 0, 1);
    } else {                                                   // capi.hip gl_run_inverse
        for (int i = np; i-- > 0;)
            run_pass(true, i == np - 1 ? src.data() : dst.data(), dst.data(), tw.data(), log_n, passes[i].s0, passes[i].r, log_tile, 0, 0, i == 0 ? hdr[3] : 1);
    }
    f = fopen(argv[2], "wb");
    fwrite(dst.data(), 8, dst.size(), f);
    fclose(f);
    return 0;
}
