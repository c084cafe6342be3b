// ubench.hip — instruction-throughput probes behind the Fp252 multiplier design
// on gfx950 (DESIGN.md "Field arithmetic").  Each probe is a loop of inline-asm
// instructions of one flavour on 8 independent register sets per lane (so the
// chain is throughput- not latency-bound with >= 2 waves per SIMD); the report is
// lane-operations per second for the whole chip.
// Every probe also reads the shader-cycle counter (s_memtime) and the constant-rate reference counter (s_memrealtime, 100 MHz) at its
// start and end, one lane per wave: the clock the chip granted THAT probe = cycles / reference ticks x 100 MHz, and the issue cost is
// reported in cycles of that clock (round 5; before, "cycles at a nominal 2.4 GHz" of probes that ran near 1.9 - VERDICT r4).
// Build: make -C tools ; run on the GPU box: tools/_build/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
constexpr int REPS = 4;   // instructions per accumulator per iteration

#define X8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define X4(M) M M M M

__device__ __forceinline__ void stamp(uint64_t &cyc, uint64_t &ref) {
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(cyc), "=s"(ref) : : "memory");
}
template <int KIND>
__global__ __launch_bounds__(256) void probe(uint64_t *out, uint32_t seed, uint64_t *clk) {
    uint64_t c0, r0, c1, r1;
    stamp(c0, r0);
    uint32_t a = threadIdx.x * 2654435761u + seed, b = (a ^ 0x9e3779b9u) | 1u;
    uint64_t q[8]; uint32_t w[8]; double d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { q[i] = a + i * 77u; w[i] = a * (i + 3); d[i] = 1.0 + 1e-9 * (double)((a + i) & 1023); }
    double da = 1.0000001, db = 1e-12;
    uint64_t c64 = ((uint64_t)b << 7) | 5;
    for (int it = 0; it < ITERS; ++it) {
        if (KIND == 0) {
#define M(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(w[i]), "v"(b) : "vcc");
            X4(X8(M))
#undef M
        } else if (KIND == 1) {
#define M(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(w[i]) : "v"(b));
            X4(X8(M))
#undef M
        } else if (KIND == 2) {
#define M(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(w[i]) : "v"(b));
            X4(X8(M))
#undef M
        } else if (KIND == 3) {
#define M(i) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(q[i]) : "v"(c64));
            X4(X8(M))
#undef M
        } else if (KIND == 4) {
#define M(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(w[i]) : "v"(b));
            X4(X8(M))
#undef M
        } else if (KIND == 5) {
#define M(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(da), "v"(db));
            X4(X8(M))
#undef M
        } else if (KIND == 6) {
#define M(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(w[i]) : "v"(b));
            X4(X8(M))
#undef M
        } else if (KIND == 7) {
#define M(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(w[i]) : "v"(b));
            X4(X8(M))
#undef M
        } else if (KIND == 8) {
            // 64-bit add as a carry pair; the gfx950 VCC hazard needs 2 wait states between them
#define M(i) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(w[i]), "+v"(w[(i + 1) & 7]) : "v"(b) : "vcc");
            X4(X8(M))
#undef M
        } else if (KIND == 9) {
#define M(i) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(w[i]));
            X4(X8(M))
#undef M
        } else if (KIND == 10) {
#define M(i) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(q[i]));
            X4(X8(M))
#undef M
        } else if (KIND == 11) {
#define M(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(da));
            X4(X8(M))
#undef M
        } else if (KIND == 12) {        // round 4: the rest of the field kernels' instruction mix (tools/alu_model.py prices them)
#define M(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(w[i]) : "v"(b));
            X4(X8(M))
#undef M
        } else if (KIND == 13) {
#define M(i) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(w[i]) : "v"(b));
            X4(X8(M))
#undef M
        } else if (KIND == 14) {
#define M(i) asm volatile("v_lshrrev_b32 %0, 28, %0" : "+v"(w[i]));
            X4(X8(M))
#undef M
        } else if (KIND == 15) {
#define M(i) asm volatile("v_lshrrev_b64 %0, 28, %0" : "+v"(q[i]));
            X4(X8(M))
#undef M
        } else if (KIND == 16) {
#define M(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(w[i]) : "v"(b));
            X4(X8(M))
#undef M
        } else if (KIND == 17) {
#define M(i) asm volatile("v_lshl_or_b32 %0, %0, 4, %1" : "+v"(w[i]) : "v"(b));
            X4(X8(M))
#undef M
        } else if (KIND == 18) {
#define M(i) asm volatile("v_bfe_u32 %0, %0, 4, 20" : "+v"(w[i]));
            X4(X8(M))
#undef M
        } else if (KIND == 19) {
#define M(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(w[i]) : "v"(b));
            X4(X8(M))
#undef M
        } else if (KIND == 20) {
#define M(i) asm volatile("v_mov_b32 %0, %1" : "+v"(w[i]) : "v"(w[(i + 1) & 7]));
            X4(X8(M))
#undef M
        } else if (KIND == 21) {
#define M(i) asm volatile("v_ashrrev_i32 %0, 28, %0" : "+v"(w[i]));
            X4(X8(M))
#undef M
        } else if (KIND == 22) {
#define M(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(w[i]) : "v"(b) : "vcc");
            X4(X8(M))
#undef M
        } else if (KIND == 23) {
#define M(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(w[i]) : "v"(b));
            X4(X8(M))
#undef M
        }
    }
    uint64_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r ^= q[i] ^ w[i] ^ (uint64_t)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    stamp(c1, r1);
    if ((threadIdx.x & 63) == 0 && clk) {               // per wave: shader cycles and reference ticks it lived
        const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        clk[2 * w] = c1 - c0;
        clk[2 * w + 1] = r1 - r0;
    }
}

static uint64_t *g_clk = nullptr;          // device: (cycles, reference ticks) per wave of the last launch
template <int KIND>
int run(const char *name, uint64_t *d_out) {
    const int blocks = 256 * 8, threads = 256, waves = blocks * threads / 64, LAUNCHES = 40;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), 0, 0, d_out, 1u, g_clk);   // the clock settles
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int rep = 0; rep < LAUNCHES; ++rep) hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), 0, 0, d_out, 2u + rep, g_clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    static uint64_t h_clk[2 * 256 * 8 * 4];
    CHECK(hipMemcpy(h_clk, g_clk, sizeof(uint64_t) * 2 * waves, hipMemcpyDeviceToHost));
    double cyc = 0, ref = 0;
    for (int w = 0; w < waves; ++w) { cyc += (double)h_clk[2 * w]; ref += (double)h_clk[2 * w + 1]; }
    const double ghz = ref > 0 ? cyc / ref * 0.1 : 0.0;          // reference counter: 100 MHz
    const double lane_ops = (double)LAUNCHES * blocks * threads * (double)ITERS * 8.0 * REPS;
    const double rate = lane_ops / (ms * 1e-3);
    // cycles per wave-instruction per SIMD: 1024 SIMDs * 64 lanes * clock / rate - at the probe's OWN clock, and at a nominal 2.4 GHz
    printf("%-22s %8.3f ms  %8.2f T lane-ops/s   ~%5.2f cyc/wave-instr @2.4GHz   own clock %.3f GHz  -> %5.2f cyc/wave-instr\n", name, ms / LAUNCHES,
           rate / 1e12, 1024.0 * 64.0 * 2.4e9 / rate, ghz, 1024.0 * 64.0 * ghz * 1e9 / rate);
    return 0;
}

int main() {
    uint64_t *d_out;
    CHECK(hipMalloc(&d_out, 256 * 8 * 256 * 8));
    CHECK(hipMalloc(&g_clk, sizeof(uint64_t) * 2 * 256 * 8 * 4));
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s  CUs %d  clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    run<4>("v_add_u32", d_out);
    run<9>("v_alignbit_b32", d_out);
    run<0>("v_mad_u64_u32", d_out);
    run<1>("v_mul_lo_u32", d_out);
    run<2>("v_mul_hi_u32", d_out);
    run<3>("v_lshl_add_u64", d_out);
    run<10>("v_lshlrev_b64", d_out);
    run<5>("v_fma_f64", d_out);
    run<11>("v_mul_f64", d_out);
    run<6>("v_mad_u32_u24", d_out);
    run<7>("v_mul_hi_u32_u24", d_out);
    run<8>("add_co+nop+addc (pair)", d_out);
    run<12>("v_and_b32", d_out);
    run<13>("v_sub_u32", d_out);
    run<14>("v_lshrrev_b32", d_out);
    run<15>("v_lshrrev_b64", d_out);
    run<16>("v_xor_b32", d_out);
    run<17>("v_lshl_or_b32", d_out);
    run<18>("v_bfe_u32", d_out);
    run<19>("v_add3_u32", d_out);
    run<20>("v_mov_b32", d_out);
    run<21>("v_ashrrev_i32", d_out);
    run<22>("v_cndmask_b32", d_out);
    run<23>("v_lshl_add_u32", d_out);
    return 0;
}
