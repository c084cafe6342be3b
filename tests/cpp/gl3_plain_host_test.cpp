// Host build of the generated composition kernel of the 64-bit field (sandstorm_amd/csrc/quotient_gen_plain_gl.inc, what
// tools/gen_quotient_gl.py writes and csrc/goldilocks.hip includes) over csrc/gl64.h - the device compiles the same two files.
// tests/test_gl64_host.py runs it over a whole evaluation domain against the oracle's constraint VM.
// usage: gl3_plain_host_test <in> <out> ; in = u64 header {ncols, N, ntable_words, ntables, nconsts, log_blowup, offset, w, blocks, threads},
// columns [ncols][N], tables, tdesc u32 [ntables][2], consts [nconsts][3]; out = [N][3]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gl64.h"

#define __global__
#define __launch_bounds__(x)
struct Dim3 { unsigned x, y, z; };
static Dim3 gridDim, blockDim, blockIdx, threadIdx;       // the "launch": main() walks the grid one lane at a time

namespace ss {
struct Gl3VmArgs {                                         // csrc/goldilocks.hip, same member names
    const uint32_t *code;
    const uint64_t *consts, *tables;
    const uint32_t *tdesc;
    const uint64_t *cols[16];
    uint64_t *slots, *out;
    uint64_t offset, w;
    uint32_t n_instr, log_blowup;
    uint64_t N;
};
#include QG_INC
}  // namespace ss

template <class T>
static void rd(FILE *f, std::vector<T> &v, size_t n) { v.resize(n); if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } }

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<uint64_t> hdr; rd(f, hdr, 10);
    const uint64_t ncols = hdr[0], N = hdr[1];
    std::vector<std::vector<uint64_t>> cols(ncols);
    for (auto &c : cols) rd(f, c, N);
    std::vector<uint64_t> tables, consts; std::vector<uint32_t> tdesc;
    rd(f, tables, hdr[2]); rd(f, tdesc, 2 * hdr[3]); rd(f, consts, 3 * hdr[4]);
    fclose(f);
    if (hdr[4] != ss::GL3_PLAIN_N_CONSTS || hdr[3] != ss::GL3_PLAIN_N_TABLES) { fprintf(stderr, "not the generated program's shape\n"); return 3; }
    std::vector<uint64_t> out(3 * N, ~0ull);
    ss::Gl3VmArgs a{};
    a.consts = consts.data(); a.tables = tables.data(); a.tdesc = tdesc.data(); a.out = out.data();
    for (uint64_t c = 0; c < ncols; ++c) a.cols[c] = cols[c].data();
    a.log_blowup = (uint32_t)hdr[5]; a.offset = hdr[6]; a.w = hdr[7]; a.N = N;
    gridDim = {(unsigned)hdr[8], 1, 1}; blockDim = {(unsigned)hdr[9], 1, 1};
    for (unsigned b = 0; b < gridDim.x; ++b)
        for (unsigned t = 0; t < blockDim.x; ++t) { blockIdx = {b, 0, 0}; threadIdx = {t, 0, 0}; ss::gl3_plain_kernel(a); }
    f = fopen(argv[2], "wb");
    fwrite(out.data(), 8, out.size(), f);
    fclose(f);
    return 0;
}
