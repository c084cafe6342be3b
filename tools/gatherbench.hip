// gatherbench.hip — what a random table read costs on gfx950: the Pedersen kernels read one 64-byte affine point per window out of a
// table of up to 23.6 GB (csrc/pedersen.hip).  Every lane reads READS entries of BYTES bytes at pseudo-random indices of a table of
// the given size (independent reads: the addresses do not depend on the data) and folds them; reported: G reads/s and GB/s of
// payload.  With --pmc FETCH_SIZE around it the fabric bytes per read follow.
//   hipcc -O3 --offload-arch=gfx950 -o _build/gatherbench gatherbench.hip && _build/gatherbench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <int BYTES, int READS>
__global__ __launch_bounds__(256) void gather_kernel(const uint4 *__restrict__ table, uint64_t entries, uint32_t *__restrict__ out, int inflight) {
    const uint64_t lane = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t x = lane * 0x9E3779B97F4A7C15ull + 0x243F6A8885A308D3ull;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int r = 0; r < READS; r += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
            const uint64_t idx = x % entries;
            const uint4 *p = table + idx * (BYTES / 16);
#pragma unroll
            for (int q = 0; q < BYTES / 16; ++q) { const uint4 v = p[q]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
        }
        if (inflight == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // one group of reads at a time (a dependent chain's shape)
    }
    out[lane] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

template <int BYTES>
int run(const uint4 *table, uint64_t table_bytes, uint32_t *out, uint64_t lanes, int inflight) {
    constexpr int READS = 64;
    const uint64_t entries = table_bytes / BYTES;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    gather_kernel<BYTES, READS><<<dim3((uint32_t)(lanes / 256)), dim3(256)>>>(table, entries, out, inflight);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    gather_kernel<BYTES, READS><<<dim3((uint32_t)(lanes / 256)), dim3(256)>>>(table, entries, out, inflight);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double reads = (double)lanes * READS;
    printf("table %6.2f GB  entry %3d B  %s: %7.2f G reads/s  %7.1f GB/s payload  (%.3f ms)\n", table_bytes / 1e9, BYTES,
           inflight == 1 ? "4 in flight per lane" : "free-running        ", reads / ms / 1e6, reads * BYTES / ms / 1e6, ms);
    return 0;
}

int main() {
    const uint64_t max_bytes = 24ull << 30;
    uint4 *table = nullptr;
    uint32_t *out = nullptr;
    const uint64_t lanes = 1ull << 22;                           // 4 M lanes = 16 waves per SIMD's worth of work queued
    CHECK(hipMalloc(&table, max_bytes));
    CHECK(hipMalloc(&out, lanes * 4));
    CHECK(hipMemset(table, 1, max_bytes));
    for (uint64_t gb : {1ull, 6ull, 24ull})
        for (int inflight : {0, 1}) {
            if (run<32>(table, gb << 30, out, lanes, inflight)) return 1;
            if (run<64>(table, gb << 30, out, lanes, inflight)) return 1;
            if (run<128>(table, gb << 30, out, lanes, inflight)) return 1;
        }
    // the Pedersen kernels' own demand for scale: 22 reads of 64 B per hash, 12.6 M hashes per recursive_2p20 proof in ~25 ms
    printf("(pedersen_acc_pairs_kernel<24>: 22 x 64 B per hash, ~0.5 G hashes/s = ~11 G reads/s, ~0.7 TB/s payload)\n");
    CHECK(hipFree(table)); CHECK(hipFree(out));
    return 0;
}
