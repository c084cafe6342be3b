// mulbench.hip — Fp252 multiplier shoot-out on gfx950: the 8 x 32-bit form (fp252.h)
// against the lazy 9 x 28-bit form (fl252.h), as dependent chains (4 independent
// chains per lane) and as NTT-style butterflies.  Reports G mulmod/s for the chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../sandstorm_amd/csrc/fp252.h"
#include "../sandstorm_amd/csrc/fl252.h"
using namespace ss;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 256;

__global__ __launch_bounds__(256) void k_fp_mul(const Fp *in, Fp *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    Fp x0 = in[4 * i], x1 = in[4 * i + 1], x2 = in[4 * i + 2], x3 = in[4 * i + 3];
    const Fp y = in[(4 * i + 5) & 1023];
    for (int it = 0; it < ITERS; ++it) { x0 = fp_mul(x0, y); x1 = fp_mul(x1, y); x2 = fp_mul(x2, y); x3 = fp_mul(x3, y); }
    out[i] = fp_add(fp_add(x0, x1), fp_add(x2, x3));
}
__global__ __launch_bounds__(256) void k_fl_mul(const Fp *in, Fp *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    Fl x0 = fl_from_fp(in[4 * i]), x1 = fl_from_fp(in[4 * i + 1]), x2 = fl_from_fp(in[4 * i + 2]), x3 = fl_from_fp(in[4 * i + 3]);
    const Fl y = fl_from_fp(in[(4 * i + 5) & 1023]);
    for (int it = 0; it < ITERS; ++it) { x0 = fl_mul(x0, y); x1 = fl_mul(x1, y); x2 = fl_mul(x2, y); x3 = fl_mul(x3, y); }
    out[i] = fp_add(fp_add(fl_to_fp(x0), fl_to_fp(x1)), fp_add(fl_to_fp(x2), fl_to_fp(x3)));
}
__global__ __launch_bounds__(256) void k_fl_mul0(const Fp *in, Fp *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    Fl x0 = fl_from_fp(in[4 * i]), x1 = fl_from_fp(in[4 * i + 1]), x2 = fl_from_fp(in[4 * i + 2]), x3 = fl_from_fp(in[4 * i + 3]);
    const Fl y = fl_from_fp(in[(4 * i + 5) & 1023]);
    for (int it = 0; it < ITERS; ++it) { x0 = fl_mul(x0, y); x1 = fl_mul(x1, y); x2 = fl_mul(x2, y); x3 = fl_mul(x3, y); }
    out[i] = fp_add(fp_add(fl_to_fp(x0), fl_to_fp(x1)), fp_add(fl_to_fp(x2), fl_to_fp(x3)));
}
// radix-4 style: 2 butterflies per iteration on 4 values (DIT: b*w, a+bw, a-bw)
__global__ __launch_bounds__(256) void k_fp_bfly(const Fp *in, Fp *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    Fp x0 = in[4 * i], x1 = in[4 * i + 1], x2 = in[4 * i + 2], x3 = in[4 * i + 3];
    const Fp w = in[(4 * i + 5) & 1023];
    for (int it = 0; it < ITERS; ++it) {
        Fp t = fp_mul(x1, w); x1 = fp_sub(x0, t); x0 = fp_add(x0, t);
        t = fp_mul(x3, w); x3 = fp_sub(x2, t); x2 = fp_add(x2, t);
        t = fp_mul(x2, w); x2 = fp_sub(x0, t); x0 = fp_add(x0, t);
        t = fp_mul(x3, w); x3 = fp_sub(x1, t); x1 = fp_add(x1, t);
    }
    out[i] = fp_add(fp_add(x0, x1), fp_add(x2, x3));
}
__global__ __launch_bounds__(256) void k_fl_bfly(const Fp *in, Fp *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    Fl x0 = fl_from_fp(in[4 * i]), x1 = fl_from_fp(in[4 * i + 1]), x2 = fl_from_fp(in[4 * i + 2]), x3 = fl_from_fp(in[4 * i + 3]);
    const Fl w = fl_from_fp(in[(4 * i + 5) & 1023]);
    for (int it = 0; it < ITERS; ++it) {
        Fl t = fl_mul(x1, w); x1 = fl_sub8p(x0, t); x0 = fl_add(x0, t);
        t = fl_mul(x3, w); x3 = fl_sub8p(x2, t); x2 = fl_add(x2, t);
        t = fl_mul(x2, w); x2 = fl_sub8p(x0, t); x0 = fl_add(x0, t);
        t = fl_mul(x3, w); x3 = fl_sub8p(x1, t); x1 = fl_add(x1, t);
        if ((it & 1) == 1) { x0 = fl_weak_reduce(x0); x1 = fl_weak_reduce(x1); x2 = fl_weak_reduce(x2); x3 = fl_weak_reduce(x3); }
    }
    out[i] = fp_add(fp_add(fl_to_fp(x0), fl_to_fp(x1)), fp_add(fl_to_fp(x2), fl_to_fp(x3)));
}

template <typename K>
int run(const char *name, K kernel, const Fp *d_in, Fp *d_out, std::vector<Fp> *host_out) {
    const int blocks = 256 * 8, threads = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, d_in, d_out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, d_in, d_out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double muls = 3.0 * blocks * threads * (double)ITERS * 4.0;
    printf("%-28s %8.3f ms  %8.2f G mulmod/s\n", name, ms / 3, muls / (ms * 1e-3) / 1e9);
    host_out->resize(blocks * threads);
    CHECK(hipMemcpy(host_out->data(), d_out, host_out->size() * sizeof(Fp), hipMemcpyDeviceToHost));
    return 0;
}

int main() {
    const int n = 256 * 8 * 256;
    std::vector<Fp> h(4 * n);
    uint64_t s = 88172645463325252ull;
    for (auto &e : h) { for (int k = 0; k < 8; ++k) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; e.v[k] = (uint32_t)s; } e.v[7] &= 0x07ffffffu; }
    Fp *d_in, *d_out;
    CHECK(hipMalloc(&d_in, h.size() * sizeof(Fp))); CHECK(hipMalloc(&d_out, n * sizeof(Fp)));
    CHECK(hipMemcpy(d_in, h.data(), h.size() * sizeof(Fp), hipMemcpyHostToDevice));
    std::vector<Fp> a, b, c, d;
    if (run("fp252 (8x32) mul chain", k_fp_mul, d_in, d_out, &a)) return 1;
    if (run("fl252 (9x28 lazy) mul chain", k_fl_mul, d_in, d_out, &b)) return 1;
    std::vector<Fp> b0;
    if (run("fl252 mul chain (shift/add)", k_fl_mul0, d_in, d_out, &b0)) return 1;
    if (run("fp252 butterflies", k_fp_bfly, d_in, d_out, &c)) return 1;
    if (run("fl252 lazy butterflies", k_fl_bfly, d_in, d_out, &d)) return 1;
    int bad = 0;
    for (int i = 0; i < n; ++i) { if (!fp_eq(a[i], b[i])) ++bad; if (!fp_eq(c[i], d[i])) ++bad; }
    printf("results agree: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    return bad != 0;
}
