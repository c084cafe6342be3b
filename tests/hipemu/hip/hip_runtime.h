// TEST INFRASTRUCTURE ONLY - a host stand-in for <hip/hip_runtime.h>, just wide enough to compile sandstorm_amd/csrc/*.hip for the
// CPU (tests/hipemu/build.sh -> tests/hipemu/_build/libsandstorm_hipemu.so) so that the DEVICE CODE of the library - every kernel,
// every launch geometry, the C ABI's host side around them - can be executed and held to the oracle without a GPU
// (tests/test_device_code_on_host.py).  It is not a fallback: nothing under sandstorm_amd/ knows it exists, the product loads
// sandstorm_amd/_build/libsandstorm_hip.so only and fails without an MI355X, and nothing here is measured.
//
// Execution model: "device" memory is host memory; a launch runs its workgroups one after the other (large grids: on a few OS threads,
// each with its own LDS and fibers); the lanes of a workgroup are fibers scheduled round-robin, each resumed once per pass, and __syncthreads() / __shfl_xor() hand control back to the
// scheduler - so a pass runs every live lane from one barrier to the next, which is exactly what a barrier guarantees.  LDS
// (`__shared__`) is thread_local storage of the OS thread that runs the workgroup.  There are no waves: code that relies on
// lock-step execution of 64 lanes without a barrier would not be emulated faithfully (the library has none; its one cross-lane
// exchange is __shfl_xor).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

// ---- language
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ thread_local
#define HIPEMU 1

#define __constant__
// vector types: plain structs WITHOUT the device's 16-byte alignment - on the host the tests' buffers are only as aligned as numpy
// makes them, and an over-aligned type would let the compiler use aligned vector moves on them
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
struct uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
struct ulonglong2 { unsigned long long x, y; };
static inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { return ulonglong2{x, y}; }

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct hipemu_uint3 { unsigned x, y, z; };
extern thread_local hipemu_uint3 threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

void hipemu_barrier();                                       // yield to the workgroup's scheduler until every live lane has arrived
#define __syncthreads() hipemu_barrier()
int hipemu_shfl_xor(int v, int mask, int width);
#define __shfl_xor(v, mask, width) hipemu_shfl_xor((v), (mask), (width))
// DPP quad permutes (ctrl < 0x100: lane 4j + i reads lane 4j + ((ctrl >> 2i) & 3)) - the only DPP form the kernels use
#define __builtin_amdgcn_mov_dpp(v, ctrl, row_mask, bank_mask, bound_ctrl) \
    hipemu_shfl_xor((v), (int)((threadIdx.x & 3u) ^ ((((unsigned)(ctrl)) >> (2u * (threadIdx.x & 3u))) & 3u)), 64)

static inline unsigned long long atomicMin(unsigned long long *p, unsigned long long v) {      // workgroups may run on several OS threads
    unsigned long long o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < o && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
// 32-bit atomics on "device" memory and LDS (trace.hip's histograms): workgroups may run on several OS threads
static inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline uint32_t atomicOr(uint32_t *p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline uint32_t atomicMin(uint32_t *p, uint32_t v) {
    uint32_t o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < o && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
static inline uint32_t atomicMax(uint32_t *p, uint32_t v) {
    uint32_t o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v > o && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
static inline uint32_t atomicCAS(uint32_t *p, uint32_t expect, uint32_t v) {
    __atomic_compare_exchange_n(p, &expect, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
    return expect;                                           // the value that was there
}
static inline uint32_t __brev(uint32_t x) { uint32_t r = 0; for (int i = 0; i < 32; ++i) r |= ((x >> i) & 1u) << (31 - i); return r; }
static inline uint64_t __brevll(uint64_t x) { uint64_t r = 0; for (int i = 0; i < 64; ++i) r |= ((x >> i) & 1ull) << (63 - i); return r; }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
static inline uint64_t __umul64hi(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }
// gfx950 builtins the kernels name: v_alignbit_b32 = low word of (hi:lo) >> (s & 31); v_bitop3_b32 = bitwise 3-input table, bit
// (a << 2 | b << 1 | c) of the table; scheduling barriers mean nothing here
static inline uint32_t hipemu_alignbit(uint32_t hi, uint32_t lo, uint32_t s) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (s & 31u)); }
static inline uint32_t hipemu_bitop3(uint32_t a, uint32_t b, uint32_t c, uint32_t tbl) {
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i)
        if ((tbl >> i) & 1u) r |= ((i & 4) ? a : ~a) & ((i & 2) ? b : ~b) & ((i & 1) ? c : ~c);
    return r;
}
#define __builtin_amdgcn_alignbit(hi, lo, s) hipemu_alignbit((hi), (lo), (s))
#define __builtin_amdgcn_bitop3_b32(a, b, c, t) hipemu_bitop3((a), (b), (c), (t))
#define __builtin_amdgcn_sched_barrier(m) ((void)0)

// ---- runtime
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101, hipErrorNotReady = 600 };
typedef struct hipemu_stream *hipStream_t;
typedef struct hipemu_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; int multiProcessorCount; size_t totalGlobalMem; };

static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "hipemu error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    memset(p, 0, sizeof(*p));
    strcpy(p->name, "host emulation of the device code (tests/hipemu)");
    strcpy(p->gcnArchName, "gfx950:hipemu");                // the arch the sources are written for; the suffix says what this is
    p->multiProcessorCount = 256;
    p->totalGlobalMem = (size_t)1 << 34;
    return hipSuccess;
}
static inline hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) { *free_b = (size_t)8 << 30; *total_b = (size_t)16 << 30; return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t bytes) { *p = malloc(bytes ? bytes : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T>
static inline hipError_t hipMalloc(T **p, size_t bytes) { return hipMalloc((void **)p, bytes); }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t = nullptr) {
    for (size_t r = 0; r < h; ++r) memmove((char *)d + r * dp, (const char *)s + r * sp, w);
    return hipSuccess;
}
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }          // launches run to completion where they are made
static inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = 0; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned flags, int) { return hipStreamCreateWithFlags(s, flags); }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
enum { hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocMapped = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }     // every "stream" of the emulation runs its work at once
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t bytes, unsigned = 0) { return hipMalloc(p, bytes); }
static inline hipError_t hipHostFree(void *p) { return hipFree(p); }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }

// one launch: every workgroup of the grid, its lanes as fibers
void hipemu_launch(dim3 grid, dim3 block, const std::function<void()> &lane_body);
template <class K, class... A>
static inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t /*lds: the tiles are statically sized here*/, hipStream_t, A... args) {
    hipemu_launch(grid, block, [&]() { kernel(args...); });
}
