// fma_mulbench.hip — ONE measured number for the FP64-FMA multiplier ("next" item 4; the MFMA idea got its measured
// kill in round 5, this one had only an argument).
//
// The modular product of the 252-bit field in radix 2^52 on the double-precision FMA pipe (Emmart / Zheng / Weems, "Faster Modular
// Exponentiation Using Double Precision Floating Point Arithmetic on the GPU", ARITH 2018): a 52 x 52 -> 104-bit partial product is
//     hi = fma_rz(a, b, 2^104)                 // 2^104 + floor(a b / 2^52) 2^52: the mantissa IS the high half
//     lo = fma_rz(a, b, (2^104 + 2^52) - hi)   // 2^52 + (a b mod 2^52):      the mantissa IS the low half
// and the halves are accumulated as the doubles' BIT PATTERNS in 64-bit integer columns (the exponent patterns are subtracted once
// per column).  p = 2^251 + 17 2^192 + 1 in this radix is (1, 0, 0, 17 2^36, 2^43) with p = 1 mod 2^52, so a Montgomery step is
// m = -t_i mod 2^52 and t += m p 2^(52 i) with m p3 and m p4 as shifts (R = 2^260).  Needs round-toward-zero for doubles: the wave's
// MODE register is set at kernel start (gfx9 has no per-instruction rounding) and kept from the compiler (set_round_toward_zero_f64).
//
// Counted per product: 25 partial products x (2 FMA + 1 subtraction + 2 64-bit integer additions) + 5 reduction steps x ~12 64-bit
// integer operations + carries + the way back to doubles ~ 210 instructions, nearly all on half-rate pipes (FMA f64 4.31 cycles,
// 64-bit add / shift 4.1 - 4.3: profiles/r05_ubench_instruction_rates.txt) against fl_mul_r280's 185 (111 v_mad_u64_u32 at 4.55, the
// rest full rate).  This file measures it: bit-exact against fp252.h first, then dependent chains and butterflies, beside the 9 x 28-bit
// form's numbers from the same launch shapes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "../sandstorm_amd/csrc/fp252.h"
#include "../sandstorm_amd/csrc/fl252.h"
using namespace ss;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 256;
struct F52 { double v[5]; };                 // integer limbs in [0, 2^52), value sum v[i] 2^(52 i), Montgomery form with R = 2^260

// HW_REG_MODE, FP_ROUND bits 3:2 (double / half precision) = 3: round toward zero.  The compiler's mode pass puts the register back
// to round-to-nearest in front of every double-precision instruction IT emits (a first version, with __builtin_fma, came out of the
// compiler with `s_setreg hwreg(HW_REG_MODE, 2, 2), 0` before the first product and failed the check on every operand pair whose
// low half rounds up), so every double-precision operation of this file is inline assembly the pass does not see, and the lane index
// every load depends on goes through the instruction that sets the mode.
__device__ __forceinline__ int set_round_toward_zero_f64(int x) {
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3" : "+v"(x));
    return x;
}
__device__ __forceinline__ double fma_rz(double a, double b, double c) { double r; asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ double add_f64(double a, double b) { double r; asm("v_add_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double sub_f64(double a, double b) { double r; asm("v_add_f64 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ u64 bits_of(double x) { return (u64)__double_as_longlong(x); }
__device__ __forceinline__ double double_of(u64 b) { return __longlong_as_double((long long)b); }

// a * b * 2^-260 mod p, limbs normalised, result < 2 p
__device__ __forceinline__ F52 f52_mul(const F52 &a, const F52 &b) {
    const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
    const u64 HI_PAT = 0x467ull << 52, LO_PAT = 0x433ull << 52, MASK = (1ull << 52) - 1;
    u64 t[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) t[k] = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const double h = fma_rz(a.v[i], b.v[j], C1);
            const double l = fma_rz(a.v[i], b.v[j], sub_f64(C2, h));
            t[i + j + 1] += bits_of(h);
            t[i + j] += bits_of(l);
        }
    // the exponent patterns: column k got lo parts of its (5 - |k - 4|) products and hi parts of column k - 1's
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const int n_lo = k <= 8 ? 5 - (k < 4 ? 4 - k : k - 4) : 0, n_hi = k >= 1 ? 5 - ((k - 1) < 4 ? 4 - (k - 1) : (k - 1) - 4) : 0;
        t[k] -= (u64)n_lo * LO_PAT + (u64)n_hi * HI_PAT;
    }
    // five Montgomery steps: m = -t_i mod 2^52; t += m (1 + 17 2^36 2^156 + 2^43 2^208) 2^(52 i)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const u64 m = (0 - t[i]) & MASK;
        t[i + 1] += (t[i] + m) >> 52;
        const u64 m17 = (m << 4) + m;                       // < 2^57
        t[i + 3] += (m17 << 36) & MASK;
        t[i + 4] += (m17 >> 16) + ((m << 43) & MASK);
        t[i + 5] += m >> 9;
    }
    F52 r;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const u64 x = t[5 + i] + c;
        c = x >> 52;
        r.v[i] = sub_f64(double_of(LO_PAT | (x & MASK)), 0x1p52);  // the integer as a double
    }
    // (c is zero: the result is below 2 p < 2^253)
    return r;
}
// sums and differences with the limbs carried back into [0, 2^52) (through the integer pipe: a sum of two limbs reaches 2^53)
__device__ __forceinline__ F52 f52_add(const F52 &x, const F52 &y) {
    const u64 LO_PAT = 0x433ull << 52, MASK = (1ull << 52) - 1;
    F52 r;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const u64 s = (bits_of(add_f64(x.v[i], 0x1p52)) & MASK) + (bits_of(add_f64(y.v[i], 0x1p52)) & MASK) + c;
        c = s >> 52;
        r.v[i] = sub_f64(double_of(LO_PAT | (s & MASK)), 0x1p52);
    }
    return r;
}
__device__ __forceinline__ F52 f52_sub4p(const F52 &x, const F52 &y) {         // x - y + 4 p (y < 4 p)
    const u64 LO_PAT = 0x433ull << 52, MASK = (1ull << 52) - 1;
    const u64 P4[5] = {4, 0, 0, 68ull << 36, 4ull << 43};                       // 4 p, limb 4 above 52 bits: carried below
    F52 r;
    long long c = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const long long s = (long long)(bits_of(add_f64(x.v[i], 0x1p52)) & MASK) - (long long)(bits_of(add_f64(y.v[i], 0x1p52)) & MASK) + (long long)P4[i] + c;
        c = s >> 52;
        r.v[i] = sub_f64(double_of(LO_PAT | ((u64)s & MASK)), 0x1p52);
    }
    // (c is zero: the values are below 8 p < 2^255, limb 4 below 2^47)
    return r;
}

__device__ __forceinline__ F52 f52_load(const u64 *p) {       // 4 x u64 (an integer below 2^256, here below p) -> 5 limbs of 52 bits
    const u64 MASK = (1ull << 52) - 1, LO_PAT = 0x433ull << 52;
    const u64 w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3];
    const u64 l[5] = {w0 & MASK, ((w0 >> 52) | (w1 << 12)) & MASK, ((w1 >> 40) | (w2 << 24)) & MASK, ((w2 >> 28) | (w3 << 36)) & MASK, w3 >> 16};
    F52 r;
#pragma unroll
    for (int i = 0; i < 5; ++i) r.v[i] = sub_f64(double_of(LO_PAT | l[i]), 0x1p52);
    return r;
}
__device__ __forceinline__ void f52_store(u64 *p, const F52 &x) {
    const u64 MASK = (1ull << 52) - 1;
    u64 l[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) l[i] = bits_of(add_f64(x.v[i], 0x1p52)) & MASK;
    p[0] = l[0] | (l[1] << 52); p[1] = (l[1] >> 12) | (l[2] << 40); p[2] = (l[2] >> 24) | (l[3] << 28); p[3] = (l[3] >> 36) | (l[4] << 16);
}

__global__ __launch_bounds__(256) void k_f52_check(const u64 *a, const u64 *b, u64 *out, int n) {
    const int i = set_round_toward_zero_f64(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    f52_store(out + 4 * i, f52_mul(f52_load(a + 4 * i), f52_load(b + 4 * i)));
}
__global__ __launch_bounds__(256) void k_f52_mul(const u64 *in, u64 *out) {
    const int i = set_round_toward_zero_f64(blockIdx.x * blockDim.x + threadIdx.x);
    F52 x0 = f52_load(in + 16 * i), x1 = f52_load(in + 16 * i + 4), x2 = f52_load(in + 16 * i + 8), x3 = f52_load(in + 16 * i + 12);
    const F52 y = f52_load(in + 4 * ((4 * i + 5) & 1023));
    for (int it = 0; it < ITERS; ++it) { x0 = f52_mul(x0, y); x1 = f52_mul(x1, y); x2 = f52_mul(x2, y); x3 = f52_mul(x3, y); }
    f52_store(out + 4 * i, f52_add(f52_add(x0, x1), f52_add(x2, x3)));
}
// x - q p with q = floor(x / 2^251) - 1 (or 0): brings a lazy sum back below 3 p, limbs normalised (what fl_weak_reduce is to the 9 x 28-bit form)
__device__ __forceinline__ F52 f52_weak(const F52 &x) {
    const u64 LO_PAT = 0x433ull << 52, MASK = (1ull << 52) - 1;
    u64 l[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) l[i] = bits_of(add_f64(x.v[i], 0x1p52)) & MASK;
    u64 q = l[4] >> 43;
    q -= q != 0;
    const long long sub[5] = {(long long)q, 0, 0, (long long)(q * (17ull << 36)), (long long)(q << 43)};      // q p, limb by limb (q < 2^9: every term below 2^52)
    F52 r;
    long long c = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const long long t = (long long)l[i] - sub[i] + c;
        c = t >> 52;
        r.v[i] = sub_f64(double_of(LO_PAT | ((u64)t & MASK)), 0x1p52);
    }
    return r;
}
// the same radix-4 style loop as k_fl_bfly: 4 products, 4 sums, 4 differences per round, the lazy values brought back every second round
__global__ __launch_bounds__(256) void k_f52_bfly(const u64 *in, u64 *out) {
    const int i = set_round_toward_zero_f64(blockIdx.x * blockDim.x + threadIdx.x);
    F52 x0 = f52_load(in + 16 * i), x1 = f52_load(in + 16 * i + 4), x2 = f52_load(in + 16 * i + 8), x3 = f52_load(in + 16 * i + 12);
    const F52 w = f52_load(in + 4 * ((4 * i + 5) & 1023));
    for (int it = 0; it < ITERS; ++it) {                    // (a product is below 2 p whatever its operands below 2^260; sums grow by < 2 p, differences by < 4 p per round)
        F52 t = f52_mul(x1, w); x1 = f52_sub4p(x0, t); x0 = f52_add(x0, t);
        t = f52_mul(x3, w); x3 = f52_sub4p(x2, t); x2 = f52_add(x2, t);
        t = f52_mul(x2, w); x2 = f52_sub4p(x0, t); x0 = f52_add(x0, t);
        t = f52_mul(x3, w); x3 = f52_sub4p(x1, t); x1 = f52_add(x1, t);
        if ((it & 1) == 1) { x0 = f52_weak(x0); x1 = f52_weak(x1); x2 = f52_weak(x2); x3 = f52_weak(x3); }
    }
    f52_store(out + 4 * i, f52_add(f52_add(x0, x1), f52_add(x2, x3)));
}
// the 9 x 28-bit form on the same shapes (tools/mulbench.hip's kernels, repeated here so that both run in one process on one clock)
__global__ __launch_bounds__(256) void k_fl_mul(const Fp *in, Fp *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    Fl x0 = fl_from_fp(in[4 * i]), x1 = fl_from_fp(in[4 * i + 1]), x2 = fl_from_fp(in[4 * i + 2]), x3 = fl_from_fp(in[4 * i + 3]);
    const Fl y = fl_from_fp(in[(4 * i + 5) & 1023]);
    for (int it = 0; it < ITERS; ++it) { x0 = fl_mul(x0, y); x1 = fl_mul(x1, y); x2 = fl_mul(x2, y); x3 = fl_mul(x3, y); }
    out[i] = fp_add(fp_add(fl_to_fp(x0), fl_to_fp(x1)), fp_add(fl_to_fp(x2), fl_to_fp(x3)));
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

static uint64_t sm(uint64_t &s) { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

int main() {
    const int N_CHECK = 1 << 16;
    // ---- bit-exact first: f52_mul(a, b) = a b 2^-260 mod p for random a, b < p, against fp252.h:
    //      fp_mul(x, y) = x y 2^-256, so a b 2^-260 = fp_mul(fp_mul(a, b), 2^252 mod p)
    std::vector<uint64_t> ha(4 * N_CHECK), hb(4 * N_CHECK), hout(4 * N_CHECK);
    uint64_t seed = 0x53414E44;
    auto rnd_below_p = [&](uint64_t *w) { for (int k = 0; k < 4; ++k) w[k] = sm(seed); w[3] &= (1ull << 59) - 1; };     // < 2^251 < p
    for (int i = 0; i < N_CHECK; ++i) { rnd_below_p(&ha[4 * i]); rnd_below_p(&hb[4 * i]); }
    // edge operands: 0, 1, p - 1, all-ones limbs
    const uint64_t pm1[4] = {0, 0, 0, 0x0800000000000011ull};
    memset(&ha[0], 0, 32); memset(&ha[4], 0, 32); ha[4] = 1; memcpy(&ha[8], pm1, 32); memcpy(&hb[8], pm1, 32); memcpy(&ha[12], pm1, 32); hb[12] = 1; hb[13] = hb[14] = hb[15] = 0;
    uint64_t *da, *db, *dout;
    CHECK(hipMalloc(&da, 32 * N_CHECK)); CHECK(hipMalloc(&db, 32 * N_CHECK)); CHECK(hipMalloc(&dout, 32 * N_CHECK));
    CHECK(hipMemcpy(da, ha.data(), 32 * N_CHECK, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb.data(), 32 * N_CHECK, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_f52_check, dim3(N_CHECK / 256), dim3(256), 0, 0, da, db, dout, N_CHECK);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(hout.data(), dout, 32 * N_CHECK, hipMemcpyDeviceToHost));
    Fp k252 = fp_zero();                                     // 2^252 mod p = 2^252 - p = 2^251 - 17 2^192 - 1
    { uint64_t w[4] = {~0ull, ~0ull, ~0ull, (1ull << 59) - 17 - 1}; for (int k = 0; k < 4; ++k) { k252.v[2 * k] = (u32)w[k]; k252.v[2 * k + 1] = (u32)(w[k] >> 32); } }
    auto fp_of = [](const uint64_t *w) { Fp r; for (int k = 0; k < 4; ++k) { r.v[2 * k] = (u32)w[k]; r.v[2 * k + 1] = (u32)(w[k] >> 32); } return r; };
    int bad = 0;
    for (int i = 0; i < N_CHECK; ++i) {
        const Fp want = fp_mul(fp_mul(fp_of(&ha[4 * i]), fp_of(&hb[4 * i])), k252);
        Fp got = fp_of(&hout[4 * i]);
        got = fp_reduce_once(got);                           // the FMA product returns a value below 2 p
        if (!fp_eq(got, want)) { if (bad < 4) printf("MISMATCH at %d: got %016llx.. want %016llx..\n", i, (unsigned long long)hout[4 * i], (unsigned long long)(((uint64_t)want.v[1] << 32) | want.v[0])); ++bad; }
    }
    printf("f52_mul (5 x 52-bit limbs on v_fma_f64, R = 2^260) against fp252.h on %d random and edge operand pairs: %s\n", N_CHECK, bad ? "MISMATCH" : "bit-identical");
    if (bad) return 2;

    // ---- rates
    int dev = 0;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int threads = prop.multiProcessorCount * 4 * 8 * 64;           // 8 waves per SIMD
    std::vector<uint64_t> hin(16 * (size_t)threads);
    for (size_t i = 0; i < hin.size(); i += 4) rnd_below_p(&hin[i]);
    uint64_t *din, *dres;
    CHECK(hipMalloc(&din, hin.size() * 8)); CHECK(hipMalloc(&dres, 32 * (size_t)threads));
    CHECK(hipMemcpy(din, hin.data(), hin.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto time_it = [&](const char *name, auto launch, double ops_per_thread, const char *unit) {
        launch();
        (void)hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0, 0);
            launch();
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        const double rate = ops_per_thread * threads / (best * 1e-3) / 1e9;
        printf("%-34s %8.3f ms  %8.1f G %s/s\n", name, best, rate, unit);
        return rate;
    };
    const dim3 grid(threads / 256), block(256);
    const double fl_chain = time_it("fl_mul   9 x 28-bit, chain", [&] { hipLaunchKernelGGL(k_fl_mul, grid, block, 0, 0, (const Fp *)din, (Fp *)dres); }, 4.0 * ITERS, "mulmod");
    const double f52_chain = time_it("f52_mul  5 x 52-bit FMA, chain", [&] { hipLaunchKernelGGL(k_f52_mul, grid, block, 0, 0, (const u64 *)din, dres); }, 4.0 * ITERS, "mulmod");
    const double fl_bfly = time_it("fl   butterflies", [&] { hipLaunchKernelGGL(k_fl_bfly, grid, block, 0, 0, (const Fp *)din, (Fp *)dres); }, 4.0 * ITERS, "butterfly");
    const double f52_bfly = time_it("f52  butterflies", [&] { hipLaunchKernelGGL(k_f52_bfly, grid, block, 0, 0, (const u64 *)din, dres); }, 4.0 * ITERS, "butterfly");
    printf("FMA form / 9 x 28-bit form: chain %.2f x, butterflies %.2f x  (the bar for a plan was 1.2 x as a butterfly)\n", f52_chain / fl_chain, f52_bfly / fl_bfly);
    return 0;
}
