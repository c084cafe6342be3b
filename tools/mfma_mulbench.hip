// mfma_mulbench.hip — ONE feasibility number for the multiplier: the 252-bit product with a SHARED multiplier
// (an NTT stage's twiddle across a batch's columns, DEEP's coefficients) as an int8 matrix product on the MFMA units.
//   a = sum_i a_i 2^(8 i), 32 signed 8-bit digits;  a * b = sum_j c_j 2^(8 j),  c_j = sum_i a_i b_(j-i), 63 columns:
//   [32 elements x 32 digits] x Toeplitz(b) [32 x 64]  =  two v_mfma_i32_32x32x32_i8 per 32 products (half the Toeplitz matrix
//   is zeros: 4096 multiply-adds issued per product for the 1024 it needs).
// What comes out is 64 int32 COLUMN SUMS per product, spread over the lanes (column = lane & 31, 16 elements per lane): before they
// can be digits again - for the Montgomery step's two further Toeplitz products with the constants, or for the next butterfly -
// every column has to give its upper bits to the next one: across LANES, three rounds for 21-bit sums.
// Reported: G products/s of (1) the two MFMAs alone, (2) with three carry rounds (lane shifts by ds_bpermute), against the
// 135 G/s of a WHOLE modular product on the vector ALU (tools/mulbench.hip fl_mul_r280: 81 v_mad_u64_u32 + a ten-step reduction
// on 17 64-bit columns held by ONE lane).  Build: make -C tools; run on the GPU box: tools/_build/mfma_mulbench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int ITERS = 512;

// A operand of 32x32x32: lane l holds row (l & 31), k = 16 (l >> 5) .. + 15 (16 int8 in 4 VGPRs); B: column (l & 31), the same k range
// C/D: column (l & 31), rows (r & 3) + 8 (r >> 2) + 4 (l >> 5), r < 16
template <int CARRY>
__global__ __launch_bounds__(256) void k_toeplitz(const int8_t *a_digits, const int8_t *b_digits, int *out, int iters) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int8_t *A = a_digits + (size_t)wave * 32 * 32;          // 32 elements x 32 digits
    const int8_t *b = b_digits + (size_t)wave * 32;               // this wave's shared multiplier
    v4i fa, fb_lo, fb_hi;
    {
        int8_t t[16];
        for (int k = 0; k < 16; ++k) t[k] = A[(lane & 31) * 32 + 16 * (lane >> 5) + k];
        __builtin_memcpy(&fa, t, 16);
        // Toeplitz(b)[k][j] = b[j - k] (0 <= j - k < 32): columns j = 0..31 (lo) and 32..63 (hi)
        for (int k = 0; k < 16; ++k) { const int kk = 16 * (lane >> 5) + k, j = (lane & 31) - kk; t[k] = j >= 0 ? b[j] : 0; }
        __builtin_memcpy(&fb_lo, t, 16);
        for (int k = 0; k < 16; ++k) { const int kk = 16 * (lane >> 5) + k, j = 32 + (lane & 31) - kk; t[k] = j < 32 ? b[j] : 0; }
        __builtin_memcpy(&fb_hi, t, 16);
    }
    v16i lo = {0}, hi = {0};
    int sink = 0;
    for (int it = 0; it < iters; ++it) {
        v16i zero = {0};
        lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb_lo, zero, 0, 0, 0);
        hi = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb_hi, zero, 0, 0, 0);
        if (CARRY) {
            // three rounds: digit = v & 0xff (signed low byte), carry = (v - digit) >> 8 goes to the next column = the next lane
            // (column 31 of `lo` feeds column 0 of `hi`)
#pragma unroll
            for (int round = 0; round < 3; ++round)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dl = (int)(int8_t)lo[r], cl = (lo[r] - dl) >> 8, dh = (int)(int8_t)hi[r], ch = (hi[r] - dh) >> 8;
                    const int from_lo = __shfl_up(cl, 1, 32), wrap = __shfl(cl, 31, 32), from_hi = __shfl_up(ch, 1, 32);
                    lo[r] = dl + ((lane & 31) ? from_lo : 0);
                    hi[r] = dh + ((lane & 31) ? from_hi : wrap);
                }
        }
        // the next product's digits would come from these; keep the chain alive without adding work
        fa[0] ^= (lo[0] + hi[15]) & 0x01010101;
        sink += lo[3] ^ hi[7];
    }
    int acc = sink;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += lo[r] * (r + 1) + hi[r] * (r + 17);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (iters == 1) {                                              // validation launch: the raw column sums of wave 0
        if (wave == 0)
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                out[4096 + row * 64 + (lane & 31)] = lo[r];
                out[4096 + row * 64 + 32 + (lane & 31)] = hi[r];
            }
    }
}

template <int CARRY>
int run(const char *name, const int8_t *d_a, const int8_t *d_b, int *d_out, int blocks) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_toeplitz<CARRY>, dim3(blocks), dim3(256), 0, 0, d_a, d_b, d_out, ITERS);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_toeplitz<CARRY>, dim3(blocks), dim3(256), 0, 0, d_a, d_b, d_out, ITERS);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double products = (double)reps * blocks * 4 /* waves */ * 32.0 * ITERS;
    printf("%-64s %8.3f ms  %9.1f G products/s\n", name, ms / reps, products / (ms * 1e-3) / 1e9);
    return 0;
}

int main() {
    const int blocks = 256 * 8, waves = blocks * 4;
    std::vector<int8_t> a((size_t)waves * 32 * 32), b((size_t)waves * 32);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (int8_t)(s >> 24); };
    for (auto &v : a) v = rnd();
    for (auto &v : b) v = rnd();
    int8_t *d_a, *d_b; int *d_out;
    CHECK(hipMalloc(&d_a, a.size())); CHECK(hipMalloc(&d_b, b.size())); CHECK(hipMalloc(&d_out, sizeof(int) * ((size_t)blocks * 256 + 4096 + 32 * 64)));
    CHECK(hipMemcpy(d_a, a.data(), a.size(), hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_b, b.data(), b.size(), hipMemcpyHostToDevice));
    // validation: wave 0's 32 x 64 column sums against the schoolbook convolution
    hipLaunchKernelGGL(k_toeplitz<0>, dim3(1), dim3(256), 0, 0, d_a, d_b, d_out, 1);
    CHECK(hipDeviceSynchronize());
    std::vector<int> got(32 * 64);
    CHECK(hipMemcpy(got.data(), d_out + 4096, got.size() * sizeof(int), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int e = 0; e < 32; ++e)
        for (int j = 0; j < 64; ++j) {
            int want = 0;
            for (int i = 0; i < 32; ++i) { const int k = j - i; if (k >= 0 && k < 32) want += (int)a[e * 32 + i] * (int)b[k]; }
            if (want != got[e * 64 + j]) ++bad;
        }
    printf("column sums of 32 products against the schoolbook convolution: %s (%d of 2048 wrong)\n", bad ? "MISMATCH" : "equal", bad);
    run<0>("Toeplitz product, two v_mfma_i32_32x32x32_i8 per 32 products", d_a, d_b, d_out, blocks);
    run<1>("  + three carry rounds over the 64 columns (lane shifts)", d_a, d_b, d_out, blocks);
    printf("for comparison (tools/mulbench.hip, same chip): a WHOLE modular product on the vector ALU, fl_mul_r280: ~135 G/s\n");
    return bad != 0;
}
