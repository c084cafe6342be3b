// capi.hip — the C ABI of libsandstorm_hip.so (include/sandstorm_hip.h).
//
// Host-side planning only: twiddle plans, pass decomposition, level-by-level
// Merkle scheduling.  All arithmetic on proof data happens in the gfx950
// kernels of ntt.hip / hash.hip / pedersen.hip / fri.hip / deep.hip /
// quotient.hip; there is no CPU fallback — without a usable HIP device every
// entry point returns SS_ERR_NO_DEVICE / SS_ERR_HIP.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cstdarg>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/sandstorm_hip.h"
#include "fp252.h"
#include "fp252_host.h"
#include "kernels.h"
#include "quotient_gen.h"
#include "ext_scan.h"
#include "gl_ntt.h"

using namespace ss;

namespace {

thread_local std::string g_err;

ss_status fail(ss_status code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) return fail(SS_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct PlanKey {
    uint32_t log_n;
    int inverse;
    uint32_t off[8];
    uint32_t win_stages = 0, win_first = 0;      // the plan of a windowed top pass (ss_ntt_shard_fp252): its stages, the window's first q
    bool operator<(const PlanKey &o) const {
        if (log_n != o.log_n) return log_n < o.log_n;
        if (inverse != o.inverse) return inverse < o.inverse;
        if (win_stages != o.win_stages) return win_stages < o.win_stages;
        if (win_first != o.win_first) return win_first < o.win_first;
        return memcmp(off, o.off, sizeof off) < 0;
    }
};

Fp fp_from_limbs64(const uint64_t v[4]) {
    Fp a;
    for (int i = 0; i < 4; ++i) { a.v[2 * i] = (u32)v[i]; a.v[2 * i + 1] = (u32)(v[i] >> 32); }
    return a;
}

// 3^((p-1)/2^log_n), (p-1) = (2^59+17) * 2^192, and its inverse: both from a table made once per process (the 8 x 32-bit arithmetic of
// fp252.h is shaped for the device: on a host core the ~230 squarings of one root cost 35 us and a power-ladder inversion 67 us, and a proof
// asked for ~60 of them between its launches - 2 ms of idle device, profiles/r05_host_gaps.txt)
struct RootTable {
    Fp w[193], wi[193];
    RootTable() {
        Fp c = fp_pow_u64(fp_from_u64(3), (1ull << 59) + 17ull);
        for (int k = 192; k >= 0; --k) { w[k] = c; c = fp_sqr(c); }
        for (int k = 0; k <= 192; ++k) wi[k] = fp_inv_safegcd(w[k]);
    }
};
const RootTable &root_table() { static const RootTable t; return t; }
Fp root_of_unity(uint32_t log_n) { return root_table().w[log_n <= 192 ? log_n : 192]; }
Fp root_of_unity_inv(uint32_t log_n) { return root_table().wi[log_n <= 192 ? log_n : 192]; }

}  // namespace

// The shader clock of a profiled launch (ss_profile_enable(ctx, 2)).  s_memtime is a free-running shader-cycle counter, but one per
// compute unit (stamps of two waves on different CUs differ by their counters' offsets - measured: a first version that paired
// before / after stamps by XCD came out at "3.1 GHz"), while s_memrealtime is the chip's one constant-rate reference counter
// (100 MHz).  So ONE wave - the monitor, launched on a stream of its own when level 2 is switched on - samples its own
// (s_memtime, s_memrealtime) pair every few microseconds into a ring for as long as the measuring pass lasts, and every profiled
// scope stamps only the reference counter before and after its launches: the monitor's cycle count interpolated at those two
// reference times is the shader cycles that passed, cycles / reference ticks x 100 MHz the clock the chip granted the scope's
// kernels (it clocks to its power budget: the transforms run ~10 % below the constraint kernels).  One DVFS domain is assumed:
// the monitor's CU clocks as the others do.
struct ClockStamp { uint64_t cycles, ref; };
static constexpr uint32_t CLOCK_MONITOR_SLOTS = 1u << 17;          // 2 MB; a sample per ~20 us: 2.6 s of proof
__device__ __forceinline__ void clock_read(uint64_t &cycles, uint64_t &ref) {
#if defined(HIPEMU)
    cycles = 0; ref = 0;
#else
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(cycles), "=s"(ref) : : "memory");
#endif
}
__global__ void clock_probe_kernel(ClockStamp *out) {
    if (threadIdx.x == 0) { ClockStamp s; clock_read(s.cycles, s.ref); *out = s; }
}
// one wave: samples until `*stop` is set (the host writes it through a pinned mapping), the ring is full, or max_ticks reference
// ticks have passed - whichever comes first, so a lost stop flag ends in seconds, not never
__global__ __launch_bounds__(64) void clock_monitor_kernel(ClockStamp *ring, uint32_t slots, volatile uint32_t *stop, uint64_t max_ticks, uint32_t *count,
                                                           volatile uint32_t *started) {
#if defined(HIPEMU)
    if (threadIdx.x == 0) { *count = 0; *started = 1; }
#else
    if (threadIdx.x != 0) return;
    uint64_t c, r, r0, next;
    clock_read(c, r0);
    next = r0;
    uint32_t n = 0;
    while (n < slots) {
        clock_read(c, r);
        if (r >= next) {
            ring[n].cycles = c; ring[n].ref = r;
            if (n == 0) { __threadfence_system(); *started = 1; }  // the host lets scopes stamp from here on (monitor_start)
            ++n;
            next = r + 2000;                                     // 20 us
            if (*stop || r - r0 > max_ticks) break;
        }
        __builtin_amdgcn_s_sleep(32);
    }
    __threadfence_system();
    *count = n;
#endif
}

struct ss_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::map<PlanKey, Fp *> plans;          // device twiddle tables
    std::map<std::tuple<uint32_t, int, uint64_t>, uint64_t *> gl_plans;     // the 64-bit field's: (log_n, inverse, offset)
    // pinned staging for the small per-launch tables that go host -> scratch in front of a kernel (the compiled constraint
    // kernels' constants): the copy and the launch are queued, the call returns without waiting for the stream
    void *stage = nullptr;
    size_t stage_bytes = 0;
    hipEvent_t stage_event = nullptr;       // recorded behind the last copy out of `stage`
    ss_status stage_acquire(size_t bytes, void **out) {
        if (stage_event) HIP_TRY(hipEventSynchronize(stage_event));     // the previous copy out of the buffer (long done by now)
        else HIP_TRY(hipEventCreateWithFlags(&stage_event, hipEventDisableTiming));
        if (bytes > stage_bytes) {
            if (stage) HIP_TRY(hipHostFree(stage));
            stage = nullptr; stage_bytes = 0;
            HIP_TRY(hipHostMalloc(&stage, bytes, hipHostMallocDefault));
            stage_bytes = bytes;
        }
        *out = stage;
        return SS_OK;
    }
    PedersenTables *ped = nullptr;
    void *scratch = nullptr;                // grow-only device scratch
    size_t scratch_bytes = 0;
    uint64_t *d_small = nullptr;            // 64 x u64: PoW prefix/best etc.
    // Device-memory pool behind ss_dev_alloc/ss_dev_free.  A proof allocates and releases the
    // same multi-GiB LDE matrices every time; hipMalloc/hipFree of blocks that size costs
    // hundreds of ms (map/unmap) and hipFree synchronises the device.  Freed blocks are kept
    // (stream-ordered reuse is safe: one stream per context) up to pool_cap bytes.
    std::multimap<size_t, void *> pool_free;          // block size -> pointer
    std::map<void *, size_t> pool_live;               // handed-out pointer -> block size
    size_t pool_cached = 0, pool_cap = (size_t)96 << 30;
    void pool_trim() {
        for (auto &kv : pool_free) (void)hipFree(kv.second);
        pool_free.clear();
        pool_cached = 0;
    }
    // hipMalloc for what lives outside the pool (plans, scratch): out of memory gives the pool's cache back and tries once more
    hipError_t malloc_retry(void **p, size_t bytes) {
        hipError_t e = hipMalloc(p, bytes);
        if (e == hipErrorOutOfMemory && !pool_free.empty()) {
            (void)hipGetLastError();
            (void)hipStreamSynchronize(stream);
            pool_trim();
            e = hipMalloc(p, bytes);
        }
        return e;
    }

    // ss_upload_async: copies on a stream of their own, an event per copy (ticket = slot + 1); the only context state another
    // host thread may touch while the owner enqueues kernels (the trace generator's thread uploads a column the moment it is final)
    hipStream_t copy_stream = nullptr;
    std::mutex copy_mutex;
    std::vector<hipEvent_t> copy_events;               // nullptr: slot free

    bool prof_on = false, prof_clock = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events[SS_PROF_KINDS];
    double prof_ms[SS_PROF_KINDS] = {0};
    uint64_t prof_launches[SS_PROF_KINDS] = {0};
    // clock stamps (prof_clock): scope i of the current batch wrote d_stamps[2 i] before and [2 i + 1] after its launches
    static constexpr size_t CLOCK_SCOPES = 8192;
    ClockStamp *d_stamps = nullptr, *d_ring = nullptr;
    uint32_t *h_stop = nullptr, *d_count = nullptr;    // pinned, mapped: the monitor's stop flag; its sample count
    hipStream_t monitor_stream = nullptr;
    bool monitor_running = false;
    std::vector<ClockStamp> monitor_samples;           // of the measuring pass so far
    std::vector<int> stamp_kinds;                      // kind of scope i
    double prof_cycles[SS_PROF_KINDS] = {0}, prof_ref[SS_PROF_KINDS] = {0};

    // bracket one launch with events (only when profiling is on)
    struct Scope {
        ss_ctx *c; int kind; hipEvent_t e1 = nullptr; long stamp = -1;
        Scope(ss_ctx *c_, int k) : c(c_), kind(k) {
            if (!c->prof_on) return;
            hipEvent_t e0;
            if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { e1 = nullptr; return; }
            if (c->prof_clock && c->monitor_running && c->stamp_kinds.size() < CLOCK_SCOPES) {
                stamp = (long)c->stamp_kinds.size();
                c->stamp_kinds.push_back(kind);
                hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, c->stream, c->d_stamps + 2 * stamp);
            }
            (void)hipEventRecord(e0, c->stream);
            c->prof_events[kind].push_back({e0, e1});
        }
        ~Scope() {
            if (!e1) return;
            (void)hipEventRecord(e1, c->stream);
            if (stamp >= 0) hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, c->stream, c->d_stamps + 2 * stamp + 1);
        }
    };
    // The monitor wave runs until it is told to stop, so it must never sit in the hardware queue the measured stream maps to (the runtime
    // spreads a process's streams over a few hardware queues; in a process that had made enough streams before, the monitor's landed
    // on the context's: the measured launches waited for the monitor's ring to fill and every scope stamped outside its samples).
    // Its stream is created at the highest priority - priority classes have queues of their own - and, belt and braces, a probe on the
    // measured stream must complete while the monitor runs: if it does not, the monitor is stopped and started on another stream.
    hipError_t monitor_stream_create() {
        int least = 0, greatest = 0;
        if (getenv("SS_CLOCK_MONITOR_NORMAL_PRIORITY") || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || greatest == least)
            return hipStreamCreateWithFlags(&monitor_stream, hipStreamNonBlocking);
        return hipStreamCreateWithPriority(&monitor_stream, hipStreamNonBlocking, greatest);
    }
    hipError_t monitor_start() {
        if (monitor_running) return hipSuccess;
        hipError_t e = hipSuccess;
        if (!d_stamps) {
            void *p = nullptr;
            if ((e = malloc_retry(&p, (2 * CLOCK_SCOPES + 1) * sizeof(ClockStamp))) != hipSuccess) return e;
            d_stamps = (ClockStamp *)p;
            if ((e = malloc_retry(&p, CLOCK_MONITOR_SLOTS * sizeof(ClockStamp))) != hipSuccess) return e;
            d_ring = (ClockStamp *)p;
            if ((e = hipHostMalloc((void **)&h_stop, 4 * sizeof(uint32_t), hipHostMallocMapped)) != hipSuccess) return e;
            d_count = h_stop + 1;
            if ((e = monitor_stream_create()) != hipSuccess) return e;
        }
        for (int attempt = 0; attempt < 6; ++attempt) {
            h_stop[0] = 0; h_stop[1] = 0; h_stop[2] = 0;
            monitor_samples.clear();
            hipLaunchKernelGGL(clock_monitor_kernel, dim3(1), dim3(64), 0, monitor_stream, d_ring, CLOCK_MONITOR_SLOTS, (volatile uint32_t *)h_stop, (uint64_t)300000000ull /* 3 s */, d_count,
                               (volatile uint32_t *)(h_stop + 2));
            if ((e = hipGetLastError()) != hipSuccess) return e;
            monitor_running = true;
            // a scope that stamps before the monitor's first sample has no cycle count to be read against: wait for the wave to be
            // there (tens of microseconds), then for a probe on the measured stream (idle: the callers synchronise it first)
            const auto t0 = std::chrono::steady_clock::now();
            while (!((volatile uint32_t *)h_stop)[2] && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(2)) {}
            hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, stream, d_stamps + 2 * CLOCK_SCOPES);
            bool through = false;
            const auto t1 = std::chrono::steady_clock::now();
            while (std::chrono::steady_clock::now() - t1 < std::chrono::milliseconds(100))
                if (hipStreamQuery(stream) == hipSuccess) { through = true; break; }
            if (through && ((volatile uint32_t *)h_stop)[2]) return hipSuccess;
            // the measured stream is stuck behind the monitor (or the monitor never started): stop it, take another stream
            if (getenv("SS_CLOCK_MONITOR_DEBUG"))
                fprintf(stderr, "clock monitor: attempt %d: started=%u, the measured stream %s\n", attempt, ((volatile uint32_t *)h_stop)[2], through ? "ran" : "did not run");
            monitor_stop();
            (void)hipStreamSynchronize(stream);
            (void)hipStreamDestroy(monitor_stream);
            monitor_stream = nullptr;
            if ((e = hipStreamCreateWithFlags(&monitor_stream, hipStreamNonBlocking)) != hipSuccess) return e;
        }
        return hipErrorNotReady;                       // no stream that leaves the measured one alone: no clocks, the caller says so
    }
    void monitor_stop() {                               // -> monitor_samples
        if (!monitor_running) return;
        *(volatile uint32_t *)h_stop = 1;
        (void)hipStreamSynchronize(monitor_stream);
        monitor_running = false;
        const uint32_t n = ((volatile uint32_t *)h_stop)[1];
        monitor_samples.resize(n);
        if (n && hipMemcpy(monitor_samples.data(), d_ring, n * sizeof(ClockStamp), hipMemcpyDeviceToHost) != hipSuccess) monitor_samples.clear();
    }
    // the monitor's cycle count at reference time `ref`, linearly between its two samples around it (-1: outside the samples)
    double monitor_cycles_at(uint64_t ref) const {
        const std::vector<ClockStamp> &m = monitor_samples;
        if (m.size() < 2 || ref < m.front().ref || ref > m.back().ref) return -1.0;
        size_t lo = 0, hi = m.size() - 1;
        while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (m[mid].ref <= ref) lo = mid; else hi = mid; }
        const double span = (double)(m[hi].ref - m[lo].ref);
        return (double)(m[lo].cycles - m.front().cycles) + (span > 0 ? (double)(m[hi].cycles - m[lo].cycles) * (double)(ref - m[lo].ref) / span : 0.0);
    }
    void prof_collect() {                              // (the stream is synchronised)
        for (int k = 0; k < SS_PROF_KINDS; ++k) {
            for (auto &pr : prof_events[k]) {
                float ms = 0;
                if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { prof_ms[k] += ms; prof_launches[k] += 1; }
                (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second);
            }
            prof_events[k].clear();
        }
        if (stamp_kinds.empty()) return;
        // the scopes' reference stamps against the monitor's samples: the monitor is stopped (and restarted if the pass goes on)
        const bool again = monitor_running;
        monitor_stop();
        std::vector<ClockStamp> st(2 * stamp_kinds.size());
        if (hipMemcpy(st.data(), d_stamps, st.size() * sizeof(ClockStamp), hipMemcpyDeviceToHost) == hipSuccess)
            for (size_t i = 0; i < stamp_kinds.size(); ++i) {
                const double c0 = monitor_cycles_at(st[2 * i].ref), c1 = monitor_cycles_at(st[2 * i + 1].ref);
                if (c0 < 0 || c1 <= c0 || st[2 * i + 1].ref <= st[2 * i].ref) continue;
                prof_cycles[stamp_kinds[i]] += c1 - c0;
                prof_ref[stamp_kinds[i]] += (double)(st[2 * i + 1].ref - st[2 * i].ref);
            }
        stamp_kinds.clear();
        if (again) (void)monitor_start();
    }

    ss_status ensure_scratch(size_t bytes) {
        if (bytes <= scratch_bytes) return SS_OK;
        if (scratch) { HIP_TRY(hipStreamSynchronize(stream)); HIP_TRY(hipFree(scratch)); scratch = nullptr; scratch_bytes = 0; }
        HIP_TRY(malloc_retry(&scratch, bytes));
        scratch_bytes = bytes;
        return SS_OK;
    }

    // Twiddle plan for the stage network of size 2^log_n: T_s[k] = h^(n/2^(s+1)) * r^(k n/2^(s+1)),
    // r = w (forward) or w^-1 (inverse), h = offset or offset^-1.  Written into d_tw (n-1 felts).
    ss_status build_plan(uint32_t log_n, bool inverse, const Fp &offset, Fp *d_tw, bool bitrev_levels = false, const NttWindow *win = nullptr,
                         uint32_t win_stages = 0) {
        const uint64_t n = 1ull << log_n;
        Fp r = root_of_unity(log_n), h = offset;
        if (inverse) { r = root_of_unity_inv(log_n); h = fp_inv_safegcd(h); }
        const bool h_is_one = fp_eq(h, fp_one());
        const uint64_t half = n / 2 ? n / 2 : 1;
        const uint64_t n_lo = half < 4096 ? half : 4096, n_hi = half < 4096 ? 1 : half / 4096;
        std::vector<Fp> host(n_lo + n_hi + log_n + 1);
        Fp *lo = host.data(), *hi = lo + n_lo, *hp = hi + n_hi;
        lo[0] = fp_one();
        for (uint64_t i = 1; i < n_lo; ++i) lo[i] = fp_mul(lo[i - 1], r);
        Fp r4096 = n_lo == 4096 ? fp_mul(lo[4095], r) : fp_one();
        hi[0] = fp_one();
        for (uint64_t i = 1; i < n_hi; ++i) hi[i] = fp_mul(hi[i - 1], r4096);
        if (log_n > 0) {
            hp[log_n - 1] = h;
            for (uint32_t s = log_n - 1; s-- > 0;) hp[s] = fp_sqr(hp[s + 1]);
        }
        Fp *d_tabs = nullptr;
        HIP_TRY(hipMalloc(&d_tabs, host.size() * sizeof(Fp)));
        HIP_TRY(hipMemcpyAsync(d_tabs, host.data(), host.size() * sizeof(Fp), hipMemcpyHostToDevice, stream));
        if (log_n > 0)
            HIP_TRY(launch_twiddles(stream, d_tw, d_tabs, d_tabs + n_lo, d_tabs + n_lo + n_hi, log_n, h_is_one, bitrev_levels, win, win_stages));
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipFree(d_tabs));
        return SS_OK;
    }
    // cached plans: the per-proof-invariant ones (trace / LDE / FRI domains)
    // bitrev_levels: every level's entries in bit-reversed order (PLAN_BITREV: what the CTI network of ntt_pass_kernel reads)
    // win: only what the windowed top pass of `win_stages` stages reads (ntt.hip PassParams.tw_entries): a transform spread over
    // R ranks costs a rank (R - 1) / R^2 of the whole plan's memory, not all of it
    ss_status get_plan(uint32_t log_n, bool inverse, const Fp &offset, const Fp **out, bool bitrev_levels = false, const NttWindow *win = nullptr,
                       uint32_t win_stages = 0) {
        PlanKey key;
        key.log_n = log_n; key.inverse = (inverse ? 1 : 0) | (bitrev_levels ? 2 : 0);
        if (win) { key.win_stages = win_stages; key.win_first = win->first; }
        memcpy(key.off, offset.v, sizeof key.off);
        auto it = plans.find(key);
        if (it != plans.end()) { *out = it->second; return SS_OK; }
        Fp *d_tw = nullptr;
        const uint64_t entries = win ? ntt_window_plan_entries(*win, win_stages) : ((1ull << log_n) > 1 ? (1ull << log_n) - 1 : 1);
        HIP_TRY(malloc_retry((void **)&d_tw, entries * NTT_PLAN_ENTRY_BYTES + 64));
        ss_status st = build_plan(log_n, inverse, offset, d_tw, bitrev_levels, win, win_stages);
        if (st != SS_OK) { (void)hipFree(d_tw); return st; }
        plans[key] = d_tw;
        *out = d_tw;
        return SS_OK;
    }
    // one reusable slot for plans whose offset changes every proof (the OOD point z)
    Fp *transient_tw = nullptr;
    size_t transient_elems = 0;
    ss_status get_transient_plan(uint32_t log_n, bool inverse, const Fp &offset, const Fp **out) {
        const size_t need = (size_t)1 << log_n;
        if (need > transient_elems) {
            if (transient_tw) { HIP_TRY(hipStreamSynchronize(stream)); HIP_TRY(hipFree(transient_tw)); transient_tw = nullptr; }
            HIP_TRY(malloc_retry((void **)&transient_tw, need * NTT_PLAN_ENTRY_BYTES + 64));
            transient_elems = need;
        }
        ss_status st = build_plan(log_n, inverse, offset, transient_tw);
        if (st != SS_OK) return st;
        *out = transient_tw;
        return SS_OK;
    }
    // FRI: the powers w_len^-k a layer's rows divide by, as two small tables per layer size (launch_fri_pow_table; domain only)
    std::map<uint32_t, Fp *> fri_pow_tabs;
    // DEEP's two denominator tables made AHEAD of ss_deep_compose (ss_deep_prepare): they depend on the out-of-domain point only,
    // so the device builds them while the host hashes the out-of-domain values into the coin
    Fp *deep_tab = nullptr;
    size_t deep_tab_elems = 0;
    struct DeepKey { bool valid = false; uint32_t log_n = 0, ncomp = 0; Fp off, z; } deep_key;
    // second grow-only scratch (tables that must coexist with `scratch`)
    void *scratch2 = nullptr;
    size_t scratch2_bytes = 0;
    ss_status ensure_scratch2(size_t bytes) {
        if (bytes <= scratch2_bytes) return SS_OK;
        if (scratch2) { HIP_TRY(hipStreamSynchronize(stream)); HIP_TRY(hipFree(scratch2)); scratch2 = nullptr; scratch2_bytes = 0; }
        HIP_TRY(malloc_retry(&scratch2, bytes));
        scratch2_bytes = bytes;
        return SS_OK;
    }
};

namespace {

// stages -> passes: pass 0 is the contiguous one (up to 2^11 adjacent
// elements), the rest take <= 7 stages each so a 2048-element tile still holds
// >= 16 adjacent elements per row (>= 512-byte global runs).
struct Pass { uint32_t s0, r; };
std::vector<Pass> plan_passes(uint32_t log_n) {
    std::vector<Pass> v;
    const uint32_t lt = (uint32_t)ntt_log_tile_max();
    const uint32_t r0 = log_n < lt ? log_n : lt;
    v.push_back({0, r0});
    uint32_t rem = log_n - r0;
    if (rem) {
        const uint32_t rmax = lt - 4;                  // rows of >= 16 adjacent elements (>= 512-byte runs)
        const uint32_t k = (rem + rmax - 1) / rmax;
        uint32_t s0 = r0;
        for (uint32_t i = 0; i < k; ++i) {
            uint32_t r = rem / (k - i) + ((rem % (k - i)) ? 1 : 0);
            v.push_back({s0, r});
            s0 += r; rem -= r;
        }
    }
    return v;
}

// forward: bit-reversed (optionally 2^-log_expand sub-sampled) src -> natural dst
ss_status run_forward(ss_ctx *ctx, const ColPtrs &cols, uint32_t ncols, uint32_t log_n, const Fp *tw,
                      uint32_t log_expand, bool canonical_out = true) {
    const uint32_t lt = (uint32_t)ntt_log_tile_max();
    const uint32_t log_tile = log_n < lt ? log_n : lt;
    std::vector<Pass> passes = plan_passes(log_n);
    if (log_expand >= log_n && log_n > 0) return fail(SS_ERR_INVALID, "log_blowup %u too large", log_expand);
    // The first log_expand stages of the network see a zero in every second input: they replicate (element i = source
    // i >> log_expand) and are skipped - whole passes of them too (a polynomial of 2^8 coefficients on 2^24 points: the 11-stage
    // pass and five stages of the next; the first pass that runs reads the source at the shifted index, whatever its stride).
    size_t start = 0;
    uint32_t skip = log_expand;
    while (start + 1 < passes.size() && skip >= passes[start].r) { skip -= passes[start].r; ++start; }
    if (skip >= passes[start].r && log_n > 0) return fail(SS_ERR_INVALID, "log_blowup %u too large", log_expand);
    ColPtrs inplace = cols;
    for (uint32_t c = 0; c < ncols; ++c) inplace.src[c] = cols.dst[c];
    for (size_t i = start; i < passes.size(); ++i) {
        const bool first = i == start;
        ss_ctx::Scope prof(ctx, SS_PROF_NTT_PASS);
        HIP_TRY(launch_ntt_pass(ctx->stream, NTT_MODE_DIT, first ? cols : inplace, ncols, tw, log_n, passes[i].s0,
                                passes[i].r, log_tile, first ? skip : 0, first ? log_expand : 0, 0, canonical_out && i + 1 == passes.size()));
    }
    return SS_OK;
}
// The inverse transform's plan and stage network: over the subgroup itself (offset 1: every trace column) the Cooley-Tukey
// network on a bit-reversed plan (NTT_MODE_CTI, ntt.hip), over a coset the Gentleman-Sande one (it folds h^-j into its twiddles).
// SS_NTT_INVERSE_DIF=1: always the latter (A/B, and the parity tests run both).
ss_status get_inverse_plan(ss_ctx *ctx, uint32_t log_n, const Fp &offset, const Fp **tw, int *mode) {
    static const bool force_dif = getenv("SS_NTT_INVERSE_DIF") != nullptr;
    const bool cti = fp_eq(offset, fp_one()) && !force_dif;
    *mode = cti ? NTT_MODE_CTI : NTT_MODE_DIF;
    return ctx->get_plan(log_n, true, offset, tw, cti);
}
// inverse: natural src -> bit-reversed dst, scaled by 1/n (2^-scale_log: the local part of a transform spread over several ranks
// scales by the whole transform's size); cti_trivial: the CTI plan's offset is one
ss_status run_inverse(ss_ctx *ctx, const ColPtrs &cols, uint32_t ncols, uint32_t log_n, const Fp *tw, int mode,
                      uint32_t scale_log = ~0u, bool cti_trivial = true) {
    if (scale_log == ~0u) scale_log = log_n;
    const uint32_t lt = (uint32_t)ntt_log_tile_max();
    const uint32_t log_tile = log_n < lt ? log_n : lt;
    std::vector<Pass> passes = plan_passes(log_n);
    ColPtrs inplace = cols;
    for (uint32_t c = 0; c < ncols; ++c) inplace.src[c] = cols.dst[c];
    for (size_t i = passes.size(); i-- > 0;) {
        const bool first = i == passes.size() - 1;
        const bool last = i == 0;
        ss_ctx::Scope prof(ctx, SS_PROF_NTT_PASS);
        HIP_TRY(launch_ntt_pass(ctx->stream, mode, first ? cols : inplace, ncols, tw, log_n, passes[i].s0,
                                passes[i].r, log_tile, 0, 0, last ? scale_log : 0, last, cti_trivial));
    }
    return SS_OK;
}

bool valid_log(uint32_t log_n) { return log_n >= 1 && log_n <= 30; }

}  // namespace

extern "C" {

const char *ss_last_error(void) { return g_err.c_str(); }
uint32_t ss_abi_version(void) { return SS_ABI_VERSION; }   // 2: ss_ctx_trim, *_ex; 3: ss_permutation_product, ss_diluted_aggregate, ss_dev_zero; 4: the row-block forms ss_eval_quotient_rows, ss_deep_compose_rows, ss_deep_extend; 5: the 64-bit field: ss_ntt_gl64, ss_lde_gl64, ss_fri_fold_gl64x3; 6: its DEEP, constraint program and row hashing: ss_ood_eval_gl64x3, ss_deep_compose_gl64x3, ss_eval_quotient_gl64x3, ss_hash_rows_gl64, ss_gather_rows_gl64, ss_running_product_gl64x3; 7: the sharded driver's data movement: ss_dev_copy, ss_dev_copy_2d, ss_bitrev_permute32, ss_comm_* (RCCL); 8: one transform / one FRI layer spread over the ranks: ss_ntt_shard_fp252, ss_fri_fold_rows; 9: ss_profile_enable(ctx, 2) + ss_profile_read_clock (shader-clock stamps around profiled launches), ss_ntt_shard_fp252 takes any number of columns, ss_upload_async / ss_wait_upload (uploads on a copy stream, ordered into the context's stream by ticket), SS_HASH_SHA256 / SS_TREE_SHA256 for the 64-bit field's rows and trees; 10: ss_gather_batch (the query phase's gathers in one round trip); 11: ss_trace_* (the base trace made on the device from trace.bin / memory.bin); 12: one extension scan over the row blocks of several devices: ss_scale_strided, ss_diluted_aggregate_block, ss_affine_apply; ss_deep_prepare (DEEP's denominator tables queued while the host hashes the out-of-domain values)

ss_status ss_ctx_create(int device, ss_ctx **out) {
    if (!out) return fail(SS_ERR_INVALID, "out is NULL");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(SS_ERR_NO_DEVICE, "no HIP device: %s", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(SS_ERR_INVALID, "device %d out of range (%d)", device, count);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SS_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    ss_ctx *ctx = new ss_ctx;
    ctx->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    HIP_TRY(hipMalloc(&ctx->d_small, 64 * sizeof(uint64_t)));
    HIP_TRY(ntt_set_func_attributes());
    HIP_TRY(gl_set_func_attributes());
    *out = ctx;
    return SS_OK;
}

void ss_ctx_destroy(ss_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->plans) hipFree(kv.second);
    ctx->pool_trim();
    for (auto &kv : ctx->pool_live) hipFree(kv.first);     // leaked by the caller
    pedersen_tables_destroy(ctx->ped);
    ctx->monitor_stop();
    for (hipEvent_t ev : ctx->copy_events) if (ev) (void)hipEventDestroy(ev);
    if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
    if (ctx->d_stamps) hipFree(ctx->d_stamps);
    if (ctx->d_ring) hipFree(ctx->d_ring);
    if (ctx->h_stop) hipHostFree(ctx->h_stop);
    if (ctx->monitor_stream) hipStreamDestroy(ctx->monitor_stream);
    if (ctx->scratch) hipFree(ctx->scratch);
    if (ctx->scratch2) hipFree(ctx->scratch2);
    if (ctx->deep_tab) hipFree(ctx->deep_tab);
    for (auto &kv : ctx->fri_pow_tabs) hipFree(kv.second);
    if (ctx->transient_tw) hipFree(ctx->transient_tw);
    if (ctx->d_small) hipFree(ctx->d_small);
    if (ctx->stage) hipHostFree(ctx->stage);
    if (ctx->stage_event) hipEventDestroy(ctx->stage_event);
    if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

ss_status ss_ctx_set_stream(ss_ctx *ctx, void *hip_stream) {
    if (!ctx) return fail(SS_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return SS_OK;
}
ss_status ss_ctx_sync(ss_ctx *ctx) {
    if (!ctx) return fail(SS_ERR_INVALID, "ctx is NULL");
    {   // the uploads of ss_upload_async too: a caller that gives up half way must not free pinned memory a copy still reads
        std::lock_guard<std::mutex> lock(ctx->copy_mutex);
        if (ctx->copy_stream) HIP_TRY(hipStreamSynchronize(ctx->copy_stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SS_OK;
}
ss_status ss_dev_alloc(ss_ctx *ctx, size_t bytes, void **d_out) {
    if (!ctx || !d_out) return fail(SS_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t want = ((bytes ? bytes : 1) + 255) & ~(size_t)255;
    // best fit among cached blocks, wasting at most a quarter of the block
    auto it = ctx->pool_free.lower_bound(want);
    if (it != ctx->pool_free.end() && it->first - want <= it->first / 4) {
        *d_out = it->second;
        ctx->pool_live[it->second] = it->first;
        ctx->pool_cached -= it->first;
        ctx->pool_free.erase(it);
        return SS_OK;
    }
    hipError_t e = hipMalloc(d_out, want);
    if (e == hipErrorOutOfMemory && !ctx->pool_free.empty()) {     // give the cache back and retry
        (void)hipGetLastError();
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->pool_trim();
        e = hipMalloc(d_out, want);
    }
    HIP_TRY(e);
    ctx->pool_live[*d_out] = want;
    return SS_OK;
}
ss_status ss_dev_free(ss_ctx *ctx, void *d_ptr) {
    if (!ctx) return fail(SS_ERR_INVALID, "ctx is NULL");
    if (!d_ptr) return SS_OK;
    auto it = ctx->pool_live.find(d_ptr);
    if (it == ctx->pool_live.end()) return fail(SS_ERR_INVALID, "ss_dev_free: pointer was not returned by ss_dev_alloc");
    const size_t sz = it->second;
    ctx->pool_live.erase(it);
    if (ctx->pool_cached + sz > ctx->pool_cap) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipFree(d_ptr));
        return SS_OK;
    }
    ctx->pool_free.emplace(sz, d_ptr);
    ctx->pool_cached += sz;
    return SS_OK;
}
ss_status ss_ctx_trim(ss_ctx *ctx) {
    if (!ctx) return fail(SS_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->pool_trim();
    if (ctx->deep_tab) { (void)hipFree(ctx->deep_tab); ctx->deep_tab = nullptr; ctx->deep_tab_elems = 0; ctx->deep_key.valid = false; }
    pedersen_tables_trim();                      // the window tables of contexts that are gone (up to 23.6 GB per device)
    return SS_OK;
}
ss_status ss_upload(ss_ctx *ctx, void *d_dst, const void *src, size_t bytes) {
    if (!ctx || (!d_dst && bytes) || (!src && bytes)) return fail(SS_ERR_INVALID, "NULL argument");
    HIP_TRY(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SS_OK;
}
ss_status ss_upload_async(ss_ctx *ctx, void *d_dst, const void *src, size_t bytes, uint64_t *ticket) {
    if (!ctx || !ticket || (!d_dst && bytes) || (!src && bytes)) return fail(SS_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lock(ctx->copy_mutex);
    HIP_TRY(hipSetDevice(ctx->device));                // (the caller may be a thread that has never touched the device)
    if (!ctx->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    hipEvent_t ev;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(ev, ctx->copy_stream);
    if (e != hipSuccess) { (void)hipEventDestroy(ev); return fail(SS_ERR_HIP, "%s", hipGetErrorString(e)); }
    size_t slot = 0;
    while (slot < ctx->copy_events.size() && ctx->copy_events[slot]) ++slot;
    if (slot == ctx->copy_events.size()) ctx->copy_events.push_back(nullptr);
    ctx->copy_events[slot] = ev;
    *ticket = slot + 1;
    return SS_OK;
}
ss_status ss_wait_upload(ss_ctx *ctx, uint64_t ticket) {
    if (!ctx) return fail(SS_ERR_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->copy_mutex);
    if (ticket == 0 || ticket > ctx->copy_events.size() || !ctx->copy_events[ticket - 1]) return fail(SS_ERR_INVALID, "no such upload ticket (a ticket is waited for once)");
    hipEvent_t ev = ctx->copy_events[ticket - 1];
    ctx->copy_events[ticket - 1] = nullptr;
    const hipError_t e = hipStreamWaitEvent(ctx->stream, ev, 0);
    (void)hipEventDestroy(ev);                          // (the wait keeps what it needs of the event)
    if (e != hipSuccess) return fail(SS_ERR_HIP, "%s", hipGetErrorString(e));
    return SS_OK;
}
ss_status ss_dev_zero(ss_ctx *ctx, void *d_ptr, size_t bytes) {
    if (!ctx || (!d_ptr && bytes)) return fail(SS_ERR_INVALID, "NULL argument");
    if (bytes) HIP_TRY(hipMemsetAsync(d_ptr, 0, bytes, ctx->stream));
    return SS_OK;
}
// ---- data movement of the sharded driver (ABI 7)
ss_status ss_dev_copy(ss_ctx *ctx, void *d_dst, const void *d_src, size_t bytes) {
    if (!ctx || (bytes && (!d_dst || !d_src))) return fail(SS_ERR_INVALID, "NULL argument");
    if (bytes) HIP_TRY(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return SS_OK;
}
// rows of `wpr` 64-bit words each, pitches in words: word i of the flattened rows goes from src[(i / wpr) * sp + i % wpr] to the same slot of dst
__global__ void __launch_bounds__(256) copy_rows_u64_kernel(uint64_t *__restrict__ dst, uint64_t dp, const uint64_t *__restrict__ src, uint64_t sp, uint32_t wpr, uint64_t words) {
    const uint64_t base = (uint64_t)blockIdx.x * (256 * 8) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint64_t i = base + (uint64_t)k * 256;
        if (i < words) {
            const uint64_t r = wpr == 1 ? i : i / wpr, w = wpr == 1 ? 0 : i - r * wpr;
            dst[r * dp + w] = src[r * sp + w];
        }
    }
}
ss_status ss_dev_copy_2d(ss_ctx *ctx, void *d_dst, size_t dst_pitch, const void *d_src, size_t src_pitch, size_t width, size_t rows) {
    if (!ctx || (width && rows && (!d_dst || !d_src))) return fail(SS_ERR_INVALID, "NULL argument");
    if (dst_pitch < width || src_pitch < width) return fail(SS_ERR_INVALID, "pitch smaller than the row width");
    if (!width || !rows) return SS_OK;
    // narrow rows of whole 64-bit words (a column out of interleaved coordinates, a column spread over every other slot, the comb of a leaf
    // block): one thread per word.  The runtime's rectangular copy moves such shapes at ~0.8 TB/s (profiles/r05_gl64_cpp_host.txt)
    if (width <= 64 && !(width & 7) && !(dst_pitch & 7) && !(src_pitch & 7) && !(((uintptr_t)d_dst | (uintptr_t)d_src) & 7) && rows * (width / 8) < (1ull << 40)) {
        const uint32_t wpr = (uint32_t)(width / 8);
        const uint64_t words = rows * wpr;
        const uint32_t per_block = 256 * 8;
        hipLaunchKernelGGL(copy_rows_u64_kernel, dim3((unsigned)((words + per_block - 1) / per_block)), dim3(256), 0, ctx->stream, (uint64_t *)d_dst, dst_pitch / 8,
                           (const uint64_t *)d_src, src_pitch / 8, wpr, words);
        HIP_TRY(hipGetLastError());
        return SS_OK;
    }
    HIP_TRY(hipMemcpy2DAsync(d_dst, dst_pitch, d_src, src_pitch, width, rows, hipMemcpyDeviceToDevice, ctx->stream));
    return SS_OK;
}
ss_status ss_bitrev_permute32(ss_ctx *ctx, const void *d_src, uint32_t log_n, void *d_dst) {
    if (!ctx || !d_src || !d_dst) return fail(SS_ERR_INVALID, "NULL argument");
    if (log_n > 40) return fail(SS_ERR_INVALID, "log_n out of range");
    if (d_src == d_dst) return fail(SS_ERR_INVALID, "ss_bitrev_permute32 is not in place");
    HIP_TRY(launch_bitrev_copy(ctx->stream, (const Fp *)d_src, log_n, (Fp *)d_dst));
    return SS_OK;
}

ss_status ss_download(ss_ctx *ctx, void *dst, const void *d_src, size_t bytes) {
    if (!ctx || (!dst && bytes) || (!d_src && bytes)) return fail(SS_ERR_INVALID, "NULL argument");
    HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SS_OK;
}

// ---- RCCL, loaded at the first use (dlopen: a process that never shards one proof over several GPUs carries no dependency)
struct ss_comm {
    ss_ctx *ctx;
    void *nccl;              // ncclComm_t
    uint32_t rank, world;
};
namespace {
struct Rccl {
    struct Id { char b[128]; };                 // ncclUniqueId (passed by value)
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, Id, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string error;
    std::once_flag once;
    bool load() {                               // two threads may create their communicators at once: one of them loads
        std::call_once(once, [this] { load_once(); });
        return lib != nullptr;
    }
    void load_once() {
        // one node, one process per GPU: the communicator's bootstrap runs over the loopback interface (on a box without any
        // other interface RCCL otherwise spends minutes probing); a deployment that wants something else sets the variable itself
        setenv("NCCL_SOCKET_IFNAME", "lo", 0);
        // the copy the process already holds (PyTorch loads the librccl.so it bundles), else the ROCm installation's
        for (const char *name : {"librccl.so", "librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (lib) break;
        }
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (lib) break;
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!lib) { error = std::string("RCCL not found: ") + dlerror(); return; }
        auto sym = [&](const char *n) { void *p = dlsym(lib, n); if (!p) error = std::string("RCCL lacks ") + n; return p; };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        AllGather = (decltype(AllGather))sym("ncclAllGather");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !AllGather) lib = nullptr;
    }
} g_rccl;
constexpr int NCCL_INT8 = 0;                    // ncclInt8 / ncclChar
}  // namespace
#define RCCL_TRY(expr) do { const int r_ = (expr); if (r_ != 0) return fail(SS_ERR_HIP, "RCCL: %s (%s)", g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "error", #expr); } while (0)

ss_status ss_comm_unique_id(uint8_t id_out[128]) {
    if (!id_out) return fail(SS_ERR_INVALID, "NULL argument");
    if (!g_rccl.load()) return fail(SS_ERR_UNSUPPORTED, "%s", g_rccl.error.c_str());
    RCCL_TRY(g_rccl.GetUniqueId(id_out));
    return SS_OK;
}
ss_status ss_comm_create(ss_ctx *ctx, const uint8_t id[128], uint32_t rank, uint32_t world, ss_comm **out) {
    if (!ctx || !id || !out) return fail(SS_ERR_INVALID, "NULL argument");
    if (world == 0 || rank >= world) return fail(SS_ERR_INVALID, "rank %u of %u", rank, world);
    if (!g_rccl.load()) return fail(SS_ERR_UNSUPPORTED, "%s", g_rccl.error.c_str());
    HIP_TRY(hipSetDevice(ctx->device));
    Rccl::Id uid;
    memcpy(uid.b, id, 128);
    void *c = nullptr;
    RCCL_TRY(g_rccl.CommInitRank(&c, (int)world, uid, (int)rank));
    ss_comm *cm = new ss_comm;
    cm->ctx = ctx; cm->nccl = c; cm->rank = rank; cm->world = world;
    *out = cm;
    return SS_OK;
}
void ss_comm_destroy(ss_comm *comm) {
    if (!comm) return;
    if (comm->nccl && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(comm->nccl);
    delete comm;
}
ss_status ss_comm_exchange(ss_comm *comm, uint32_t nsend, const uint32_t *send_peer, const void *const *d_send, const uint64_t *send_bytes,
                           uint32_t nrecv, const uint32_t *recv_peer, void *const *d_recv, const uint64_t *recv_bytes) {
    if (!comm || (nsend && (!send_peer || !d_send || !send_bytes)) || (nrecv && (!recv_peer || !d_recv || !recv_bytes)))
        return fail(SS_ERR_INVALID, "NULL argument");
    hipStream_t s = comm->ctx->stream;
    const uint32_t r = comm->rank;
    for (uint32_t i = 0; i < nsend; ++i)
        if (send_peer[i] >= comm->world || (send_bytes[i] && !d_send[i])) return fail(SS_ERR_INVALID, "send %u: bad peer or NULL buffer", i);
    for (uint32_t j = 0; j < nrecv; ++j)
        if (recv_peer[j] >= comm->world || (recv_bytes[j] && !d_recv[j])) return fail(SS_ERR_INVALID, "receive %u: bad peer or NULL buffer", j);
    // messages to the own rank: matched in order, device copies
    uint32_t j = 0;
    for (uint32_t i = 0; i < nsend; ++i) {
        if (send_peer[i] != r) continue;
        while (j < nrecv && recv_peer[j] != r) ++j;
        if (j == nrecv || recv_bytes[j] != send_bytes[i]) return fail(SS_ERR_INVALID, "a message to the own rank has no matching receive");
        if (send_bytes[i]) HIP_TRY(hipMemcpyAsync(d_recv[j], d_send[i], send_bytes[i], hipMemcpyDeviceToDevice, s));
        ++j;
    }
    if (comm->world == 1) return SS_OK;
    // the group is closed on every path: a Send / Recv that fails inside an open group would leave this thread's later RCCL calls
    // undefined; the first error is the one reported
    RCCL_TRY(g_rccl.GroupStart());
    int first_err = 0;
    const char *what = "";
    for (uint32_t i = 0; i < nsend && !first_err; ++i)
        if (send_peer[i] != r && send_bytes[i]) { first_err = g_rccl.Send(d_send[i], send_bytes[i], NCCL_INT8, (int)send_peer[i], comm->nccl, s); what = "ncclSend"; }
    for (uint32_t k = 0; k < nrecv && !first_err; ++k)
        if (recv_peer[k] != r && recv_bytes[k]) { first_err = g_rccl.Recv(d_recv[k], recv_bytes[k], NCCL_INT8, (int)recv_peer[k], comm->nccl, s); what = "ncclRecv"; }
    const int end_err = g_rccl.GroupEnd();
    if (first_err) return fail(SS_ERR_HIP, "RCCL: %s (%s)", g_rccl.GetErrorString ? g_rccl.GetErrorString(first_err) : "error", what);
    if (end_err) return fail(SS_ERR_HIP, "RCCL: %s (ncclGroupEnd)", g_rccl.GetErrorString ? g_rccl.GetErrorString(end_err) : "error");
    return SS_OK;
}
ss_status ss_comm_all_gather(ss_comm *comm, const void *d_send, uint64_t bytes, void *d_recv) {
    if (!comm || (bytes && (!d_send || !d_recv))) return fail(SS_ERR_INVALID, "NULL argument");
    if (!bytes) return SS_OK;
    if (comm->world == 1) {
        if (d_send != d_recv) HIP_TRY(hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, comm->ctx->stream));
        return SS_OK;
    }
    RCCL_TRY(g_rccl.AllGather(d_send, d_recv, bytes, NCCL_INT8, comm->nccl, comm->ctx->stream));
    return SS_OK;
}

ss_status ss_profile_enable(ss_ctx *ctx, int on) {
    if (!ctx) return fail(SS_ERR_INVALID, "ctx is NULL");
    if (ctx->prof_clock && on != 2) { HIP_TRY(hipStreamSynchronize(ctx->stream)); ctx->prof_collect(); ctx->monitor_stop(); }
    if (on == 2 && !ctx->prof_clock) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        const hipError_t me = ctx->monitor_start();
        if (me == hipErrorNotReady) return fail(SS_ERR_UNSUPPORTED, "no stream for the clock monitor that leaves the context's stream running (levels 0 and 1 are available)");
        HIP_TRY(me);
    }
    ctx->prof_on = on != 0;
    ctx->prof_clock = on == 2;
    return SS_OK;
}
ss_status ss_profile_reset(ss_ctx *ctx) {
    if (!ctx) return fail(SS_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->prof_collect();
    for (int k = 0; k < SS_PROF_KINDS; ++k) { ctx->prof_ms[k] = 0; ctx->prof_launches[k] = 0; ctx->prof_cycles[k] = 0; ctx->prof_ref[k] = 0; }
    return SS_OK;
}
ss_status ss_profile_read_clock(ss_ctx *ctx, int kind, double *shader_cycles, double *ref_ticks) {
    if (!ctx || kind < 0 || kind >= SS_PROF_KINDS) return fail(SS_ERR_INVALID, "bad argument");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->prof_collect();
    if (shader_cycles) *shader_cycles = ctx->prof_cycles[kind];
    if (ref_ticks) *ref_ticks = ctx->prof_ref[kind];
    return SS_OK;
}
ss_status ss_profile_read(ss_ctx *ctx, int kind, double *total_ms, uint64_t *launches) {
    if (!ctx || kind < 0 || kind >= SS_PROF_KINDS) return fail(SS_ERR_INVALID, "bad argument");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->prof_collect();
    if (total_ms) *total_ms = ctx->prof_ms[kind];
    if (launches) *launches = ctx->prof_launches[kind];
    return SS_OK;
}

// every entry of a caller's column-pointer table names memory: a NULL one would only show as a fault inside a kernel
static bool has_null(const void *const *cols, uint32_t n) {
    for (uint32_t c = 0; c < n; ++c) if (!cols[c]) return true;
    return false;
}
static bool valid_order(int order) { return order == SS_ORDER_NATURAL || order == SS_ORDER_BITREV; }

// ------------------------------------------------------------------- NTT
ss_status ss_ntt_fp252(ss_ctx *ctx, uint64_t *const *d_cols, uint32_t ncols, uint32_t log_n, int direction,
                       const uint64_t offset[4], int in_order, int out_order) {
    if (!ctx || !d_cols) return fail(SS_ERR_INVALID, "NULL argument");
    if (!valid_log(log_n)) return fail(SS_ERR_INVALID, "log_n %u out of range [1,30]", log_n);
    if (direction != SS_NTT_FORWARD && direction != SS_NTT_INVERSE) return fail(SS_ERR_INVALID, "bad direction");
    if (!valid_order(in_order) || !valid_order(out_order)) return fail(SS_ERR_INVALID, "bad element order %d / %d", in_order, out_order);
    if (has_null((const void *const *)d_cols, ncols)) return fail(SS_ERR_INVALID, "NULL column");
    const Fp off = offset ? fp_from_limbs64(offset) : fp_one();
    const bool inverse = direction == SS_NTT_INVERSE;
    const Fp *tw = nullptr;
    int inv_mode = NTT_MODE_DIF;
    ss_status st = inverse ? get_inverse_plan(ctx, log_n, off, &tw, &inv_mode) : ctx->get_plan(log_n, false, off, &tw);
    if (st != SS_OK) return st;
    for (uint32_t base = 0; base < ncols; base += MAX_COLS) {
        const uint32_t nc = ncols - base < (uint32_t)MAX_COLS ? ncols - base : (uint32_t)MAX_COLS;
        ColPtrs cols;
        memset(&cols, 0, sizeof cols);
        for (uint32_t c = 0; c < nc; ++c) { cols.src[c] = d_cols[base + c]; cols.dst[c] = d_cols[base + c]; }
        if (!inverse) {
            if (in_order == SS_ORDER_NATURAL)
                for (uint32_t c = 0; c < nc; ++c) HIP_TRY(launch_bitrev(ctx->stream, (Fp *)cols.dst[c], log_n));
            st = run_forward(ctx, cols, nc, log_n, tw, 0);
            if (st != SS_OK) return st;
            if (out_order == SS_ORDER_BITREV)
                for (uint32_t c = 0; c < nc; ++c) HIP_TRY(launch_bitrev(ctx->stream, (Fp *)cols.dst[c], log_n));
        } else {
            if (in_order == SS_ORDER_BITREV)
                for (uint32_t c = 0; c < nc; ++c) HIP_TRY(launch_bitrev(ctx->stream, (Fp *)cols.dst[c], log_n));
            st = run_inverse(ctx, cols, nc, log_n, tw, inv_mode);
            if (st != SS_OK) return st;
            if (out_order == SS_ORDER_NATURAL)
                for (uint32_t c = 0; c < nc; ++c) HIP_TRY(launch_bitrev(ctx->stream, (Fp *)cols.dst[c], log_n));
        }
    }
    return SS_OK;
}

ss_status ss_lde_fp252(ss_ctx *ctx, const uint64_t *const *d_in, uint32_t ncols, uint32_t log_n,
                       uint32_t log_blowup, const uint64_t offset[4], uint64_t *const *d_evals,
                       uint64_t *const *d_coeffs) {
    if (!ctx || !d_in || !d_evals) return fail(SS_ERR_INVALID, "NULL argument");
    if (!valid_log(log_n) || !valid_log(log_n + log_blowup)) return fail(SS_ERR_INVALID, "size out of range");
    if (has_null((const void *const *)d_in, ncols) || has_null((const void *const *)d_evals, ncols) || (d_coeffs && has_null((const void *const *)d_coeffs, ncols)))
        return fail(SS_ERR_INVALID, "NULL column");
    const Fp off = offset ? fp_from_limbs64(offset) : fp_one();
    const Fp *tw_inv = nullptr, *tw_fwd = nullptr;
    int inv_mode = NTT_MODE_DIF;
    ss_status st = get_inverse_plan(ctx, log_n, fp_one(), &tw_inv, &inv_mode);
    if (st != SS_OK) return st;
    st = ctx->get_plan(log_n + log_blowup, false, off, &tw_fwd);
    if (st != SS_OK) return st;
    const size_t col_bytes = sizeof(Fp) << log_n;
    for (uint32_t base = 0; base < ncols; base += MAX_COLS) {
        const uint32_t nc = ncols - base < (uint32_t)MAX_COLS ? ncols - base : (uint32_t)MAX_COLS;
        if (!d_coeffs) { st = ctx->ensure_scratch(col_bytes * nc); if (st != SS_OK) return st; }
        ColPtrs inv, fwd;
        memset(&inv, 0, sizeof inv); memset(&fwd, 0, sizeof fwd);
        for (uint32_t c = 0; c < nc; ++c) {
            void *co = d_coeffs ? (void *)d_coeffs[base + c] : (void *)((char *)ctx->scratch + col_bytes * c);
            inv.src[c] = d_in[base + c]; inv.dst[c] = co;
            fwd.src[c] = co; fwd.dst[c] = d_evals[base + c];
        }
        st = run_inverse(ctx, inv, nc, log_n, tw_inv, inv_mode);
        if (st != SS_OK) return st;
        st = run_forward(ctx, fwd, nc, log_n + log_blowup, tw_fwd, log_blowup);
        if (st != SS_OK) return st;
    }
    return SS_OK;
}

ss_status ss_evaluate_fp252(ss_ctx *ctx, const uint64_t *const *d_coeffs, uint32_t ncols, uint32_t log_n,
                            uint32_t log_blowup, const uint64_t offset[4], uint64_t *const *d_evals) {
    if (!ctx || !d_coeffs || !d_evals) return fail(SS_ERR_INVALID, "NULL argument");
    if (!valid_log(log_n) || !valid_log(log_n + log_blowup)) return fail(SS_ERR_INVALID, "size out of range");
    if (has_null((const void *const *)d_coeffs, ncols) || has_null((const void *const *)d_evals, ncols)) return fail(SS_ERR_INVALID, "NULL column");
    const Fp off = offset ? fp_from_limbs64(offset) : fp_one();
    const Fp *tw = nullptr;
    ss_status st = ctx->get_plan(log_n + log_blowup, false, off, &tw);
    if (st != SS_OK) return st;
    for (uint32_t base = 0; base < ncols; base += MAX_COLS) {
        const uint32_t nc = ncols - base < (uint32_t)MAX_COLS ? ncols - base : (uint32_t)MAX_COLS;
        ColPtrs fwd;
        memset(&fwd, 0, sizeof fwd);
        for (uint32_t c = 0; c < nc; ++c) { fwd.src[c] = d_coeffs[base + c]; fwd.dst[c] = d_evals[base + c]; }
        st = run_forward(ctx, fwd, nc, log_n + log_blowup, tw, log_blowup);
        if (st != SS_OK) return st;
    }
    return SS_OK;
}

// One rank's share of ONE transform spread over R = 2^log_ranks ranks (include/sandstorm_hip.h; host/sharded.cpp).
ss_status ss_ntt_shard_fp252(ss_ctx *ctx, uint64_t *const *d_cols, uint32_t ncols, uint32_t log_n, uint32_t log_ranks, uint32_t rank,
                             int direction, const uint64_t offset[4], int part, uint32_t log_expand, uint64_t *const *d_out) {
    if (!ctx || !d_cols) return fail(SS_ERR_INVALID, "NULL argument");
    if (direction != SS_NTT_FORWARD && direction != SS_NTT_INVERSE) return fail(SS_ERR_INVALID, "bad direction");
    if (part != SS_NTT_PART_LOCAL && part != SS_NTT_PART_CROSS) return fail(SS_ERR_INVALID, "bad part");
    if (!valid_log(log_n) || log_ranks == 0 || 2 * log_ranks > log_n || log_ranks > 7) return fail(SS_ERR_INVALID, "a transform of 2^%u points does not split over 2^%u ranks (needs n >= R^2)", log_n, log_ranks);
    if (rank >> log_ranks) return fail(SS_ERR_INVALID, "rank %u of %u", rank, 1u << log_ranks);
    if (ncols == 0) return fail(SS_ERR_INVALID, "no columns");
    if (has_null((const void *const *)d_cols, ncols) || (d_out && has_null((const void *const *)d_out, ncols))) return fail(SS_ERR_INVALID, "NULL column");
    for (uint32_t c0 = 0; ncols > (uint32_t)MAX_COLS && c0 < ncols; c0 += MAX_COLS) {        // any number of columns, MAX_COLS per launch (as ss_ntt_fp252)
        const uint32_t k = ncols - c0 < (uint32_t)MAX_COLS ? ncols - c0 : (uint32_t)MAX_COLS;
        const ss_status st = ss_ntt_shard_fp252(ctx, d_cols + c0, k, log_n, log_ranks, rank, direction, offset, part, log_expand, d_out ? d_out + c0 : nullptr);
        if (st != SS_OK || c0 + k == ncols) return st;
    }
    const bool inverse = direction == SS_NTT_INVERSE, local = part == SS_NTT_PART_LOCAL;
    if (log_expand && (inverse || !local)) return fail(SS_ERR_INVALID, "log_expand belongs to the local part of a forward transform");
    if (log_expand >= log_n - log_ranks && log_expand) return fail(SS_ERR_INVALID, "log_expand %u too large", log_expand);
    const Fp off = offset ? fp_from_limbs64(offset) : fp_one();
    const uint32_t log_local = log_n - log_ranks;
    ColPtrs cols;
    memset(&cols, 0, sizeof cols);
    for (uint32_t c = 0; c < ncols; ++c) { cols.src[c] = d_cols[c]; cols.dst[c] = d_out ? d_out[c] : d_cols[c]; }
    static const bool force_dif = getenv("SS_NTT_INVERSE_DIF") != nullptr;
    const bool cti = inverse && fp_eq(off, fp_one()) && !force_dif;
    const Fp *tw = nullptr;
    if (local) {
        // Stage s < log_local of the whole network pairs elements of ONE block and its twiddle only reads the index bits below s:
        // on a block the local stages ARE a whole transform of 2^log_local points - DIT / DIF: over the coset offset^R
        // (T_s[k] = offset^(n / 2^(s+1)) w_n^(k n / 2^(s+1)) either way); CTI: the remainder tree below node `rank` of level
        // log_ranks, whose modulus is x^(n/R) - c^(n/R) with c = w_n^-bitrev(rank): the smaller tree's twiddles times c^(2^s),
        // which is what a bit-reversed plan built for the offset c^-1 holds.
        if (!inverse) {
            Fp h = off;
            for (uint32_t i = 0; i < log_ranks; ++i) h = fp_sqr(h);
            ss_status st = ctx->get_plan(log_local, false, h, &tw);
            if (st != SS_OK) return st;
            return run_forward(ctx, cols, ncols, log_local, tw, log_expand, false);
        }
        if (cti) {
            uint32_t br = 0;
            for (uint32_t i = 0; i < log_ranks; ++i) br |= ((rank >> i) & 1u) << (log_ranks - 1 - i);
            const Fp c_inv = fp_pow_u64(root_of_unity(log_n), br);
            ss_status st = ctx->get_plan(log_local, true, c_inv, &tw, true);
            if (st != SS_OK) return st;
            return run_inverse(ctx, cols, ncols, log_local, tw, NTT_MODE_CTI, log_n, br == 0);
        }
        Fp h = off;
        for (uint32_t i = 0; i < log_ranks; ++i) h = fp_sqr(h);
        ss_status st = ctx->get_plan(log_local, true, h, &tw);
        if (st != SS_OK) return st;
        return run_inverse(ctx, cols, ncols, log_local, tw, NTT_MODE_DIF, log_n);
    }
    // the top log_ranks stages: one windowed pass over the rank's share of every row (ntt.hip PassParams) with a plan that holds
    // what this window's butterflies read - DIT / DIF: (2^log_ranks - 1) window lengths of the top stages' entries; CTI: its top
    // levels ARE the plan of a transform of 2^log_ranks points
    NttWindow win;
    win.log_len = log_local - log_ranks;
    win.first = rank << win.log_len;
    uint64_t tw_entries;
    ss_status st;
    if (cti) { tw_entries = (1ull << log_ranks) - 1ull; st = ctx->get_plan(log_ranks, true, fp_one(), &tw, true); }
    else { tw_entries = ntt_window_plan_entries(win, log_ranks); st = ctx->get_plan(log_n, inverse, off, &tw, false, &win, log_ranks); }
    if (st != SS_OK) return st;
    const uint32_t lt = (uint32_t)ntt_log_tile_max();
    const uint32_t log_t = win.log_len < lt - log_ranks ? win.log_len : lt - log_ranks;
    ss_ctx::Scope prof(ctx, SS_PROF_NTT_PASS);
    HIP_TRY(launch_ntt_pass(ctx->stream, inverse ? (cti ? NTT_MODE_CTI : NTT_MODE_DIF) : NTT_MODE_DIT, cols, ncols, tw, log_n, log_local, log_ranks,
                            log_ranks + log_t, 0, 0, 0, !inverse, true, &win, tw_entries));
    return SS_OK;
}

// --------------------------------------------------------------- hashing
ss_status ss_hash_rows(ss_ctx *ctx, int hash_kind, const uint64_t *const *d_cols, uint32_t ncols,
                       uint64_t nrows, uint8_t *d_digests) {
    return ss_hash_rows_ex(ctx, hash_kind, d_cols, ncols, nrows, SS_ORDER_NATURAL, d_digests);
}
ss_status ss_hash_rows_ex(ss_ctx *ctx, int hash_kind, const uint64_t *const *d_cols, uint32_t ncols,
                          uint64_t nrows, int row_order, uint8_t *d_digests) {
    if (!ctx || !d_cols || !d_digests) return fail(SS_ERR_INVALID, "NULL argument");
    uint32_t brev_bits = 0;
    if (row_order == SS_ORDER_BITREV) {
        if (nrows == 0 || (nrows & (nrows - 1))) return fail(SS_ERR_INVALID, "bit-reversed row order needs a power-of-two row count");
        while ((1ull << brev_bits) < nrows) ++brev_bits;
    } else if (row_order != SS_ORDER_NATURAL) {
        return fail(SS_ERR_INVALID, "bad row order %d", row_order);
    }
    if (hash_kind < 0 || hash_kind > 3) return fail(SS_ERR_INVALID, "bad hash kind %d", hash_kind);
    if (ncols == 0 || ncols > (uint32_t)MAX_COLS) return fail(SS_ERR_UNSUPPORTED, "ncols %u not in [1,%d]", ncols, MAX_COLS);
    if (has_null((const void *const *)d_cols, ncols)) return fail(SS_ERR_INVALID, "NULL column");
    ConstColPtrs cols;
    memset(&cols, 0, sizeof cols);
    for (uint32_t c = 0; c < ncols; ++c) cols.p[c] = d_cols[c];
    ss_ctx::Scope prof(ctx, SS_PROF_HASH_ROWS);
    HIP_TRY(launch_hash_rows(ctx->stream, hash_kind, cols, ncols, nrows, brev_bits, d_digests));
    return SS_OK;
}

ss_status ss_merkle_build(ss_ctx *ctx, int tree_kind, uint32_t n_friendly_layers, int leaf_kind,
                          const void *d_leaves, uint64_t n, uint8_t *d_nodes, uint8_t *d_tags,
                          uint8_t root_out[33]) {
    return ss_merkle_build_ex(ctx, tree_kind, n_friendly_layers, leaf_kind, d_leaves, n, SS_ORDER_NATURAL, d_nodes, d_tags, root_out);
}
ss_status ss_merkle_build_ex(ss_ctx *ctx, int tree_kind, uint32_t n_friendly_layers, int leaf_kind,
                             const void *d_leaves, uint64_t n, int leaf_order, uint8_t *d_nodes, uint8_t *d_tags,
                             uint8_t root_out[33]) {
    if (!ctx || !d_leaves || !d_nodes) return fail(SS_ERR_INVALID, "NULL argument");
    if (n < 2 || (n & (n - 1))) return fail(SS_ERR_INVALID, "n must be a power of two >= 2");
    if (tree_kind < 0 || tree_kind > SS_TREE_SHA256) return fail(SS_ERR_INVALID, "bad tree kind");
    if (tree_kind == SS_TREE_SHA256 && leaf_kind != SS_LEAF_DIGEST) return fail(SS_ERR_UNSUPPORTED, "SHA-256 trees take digests as leaves");
    if (leaf_kind != SS_LEAF_DIGEST && leaf_kind != SS_LEAF_FELT) return fail(SS_ERR_INVALID, "bad leaf kind %d", leaf_kind);
    if (leaf_order != SS_ORDER_NATURAL && leaf_order != SS_ORDER_BITREV) return fail(SS_ERR_INVALID, "bad leaf order %d", leaf_order);
    uint32_t log_n = 0;
    while ((1ull << log_n) < n) ++log_n;
    if (leaf_kind == SS_LEAF_FELT && leaf_order == SS_ORDER_BITREV) {
        // single-column matrix committed in bit-reversed order: its permuted image is the leaf array
        ss_status pst = ctx->ensure_scratch2(n * sizeof(Fp));
        if (pst != SS_OK) return pst;
        HIP_TRY(launch_bitrev_copy(ctx->stream, (const Fp *)d_leaves, log_n, (Fp *)ctx->scratch2));
        d_leaves = ctx->scratch2;
    }
    hipStream_t s = ctx->stream;
    const int hk = tree_kind == SS_TREE_KECCAK ? SS_HASH_KECCAK : tree_kind == SS_TREE_KECCAK_M20 ? SS_HASH_KECCAK_M20
                 : tree_kind == SS_TREE_BLAKE2S ? SS_HASH_BLAKE2S : tree_kind == SS_TREE_SHA256 ? SS_HASH_SHA256 : SS_HASH_BLAKE2S_M20;
    if (tree_kind == SS_TREE_FRIENDLY && !ctx->ped) HIP_TRY(pedersen_tables_create(s, &ctx->ped));
    Fp *ped_tmp = nullptr;
    if (tree_kind == SS_TREE_FRIENDLY) {
        ss_status st = ctx->ensure_scratch(PEDERSEN_TMP_FELTS_PER_HASH * (n / 2) * sizeof(Fp));
        if (st != SS_OK) return st;
        ped_tmp = (Fp *)ctx->scratch;
    }
    HIP_TRY(hipMemsetAsync(d_nodes, 0, 64, s));
    if (d_tags) HIP_TRY(hipMemsetAsync(d_tags, 0, 2 * n, s));
    // leaf slots
    if (leaf_kind == SS_LEAF_FELT) {
        HIP_TRY(launch_felts_to_be(s, (const Fp *)d_leaves, n, d_nodes + 32 * n));
    } else {
        HIP_TRY(hipMemcpyAsync(d_nodes + 32 * n, d_leaves, 32 * n, hipMemcpyDeviceToDevice, s));
        if (d_tags && tree_kind == SS_TREE_FRIENDLY) HIP_TRY(hipMemsetAsync(d_tags + n, 1, n, s));
    }
    // level d holds nodes [2^d, 2^(d+1)); `depth` of an output node = its level, root = 0
    ss_ctx::Scope prof(ctx, SS_PROF_MERKLE);
    for (uint32_t d = log_n; d-- > 0;) {
        const uint64_t count = 1ull << d;
        const uint8_t *in = d_nodes + 32 * (2 * count);
        uint8_t *out = d_nodes + 32 * count;
        const bool leaf_level = d == log_n - 1;
        if (tree_kind == SS_TREE_FRIENDLY) {
            const bool pedersen = leaf_kind == SS_LEAF_FELT || d < n_friendly_layers;
            if (leaf_level && leaf_kind == SS_LEAF_FELT) {
                HIP_TRY(launch_pedersen_felt_pairs(s, ctx->ped, (const Fp *)d_leaves, count, out, ped_tmp));
            } else if (pedersen) {
                HIP_TRY(launch_pedersen_pairs(s, ctx->ped, in, count, out, ped_tmp));
            } else {
                HIP_TRY(launch_hash_pairs(s, SS_HASH_BLAKE2S_M20, in, count, out));
                if (d_tags) HIP_TRY(hipMemsetAsync(d_tags + count, 1, count, s));
            }
        } else if (leaf_level && leaf_kind == SS_LEAF_FELT) {
            HIP_TRY(launch_hash_felt_pairs(s, hk, (const Fp *)d_leaves, count, out));
        } else {
            HIP_TRY(launch_hash_pairs(s, hk, in, count, out));
        }
    }
    if (root_out) {
        HIP_TRY(hipMemcpyAsync(root_out, d_nodes + 32, 32, hipMemcpyDeviceToHost, s));
        root_out[32] = 0;
        if (d_tags) HIP_TRY(hipMemcpyAsync(root_out + 32, d_tags + 1, 1, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (!d_tags && tree_kind == SS_TREE_FRIENDLY)
            root_out[32] = (leaf_kind == SS_LEAF_FELT || n_friendly_layers > 0) ? 0 : 1;
    }
    return SS_OK;
}

ss_status ss_merkle_open(ss_ctx *ctx, const uint8_t *d_nodes, const uint8_t *d_tags, uint64_t n,
                         const uint64_t *idx, uint32_t nidx, uint8_t *out, uint8_t *out_tags) {
    if (!ctx || !d_nodes || (!idx && nidx) || (!out && nidx)) return fail(SS_ERR_INVALID, "NULL argument");
    if (n < 2 || (n & (n - 1))) return fail(SS_ERR_INVALID, "n must be a power of two >= 2");
    uint32_t log_n = 0;
    while ((1ull << log_n) < n) ++log_n;
    const uint64_t total = (uint64_t)nidx * log_n;
    if (total == 0) return SS_OK;
    std::vector<uint64_t> sib(total);
    for (uint32_t q = 0; q < nidx; ++q) {
        if (idx[q] >= n) return fail(SS_ERR_INVALID, "leaf index %llu out of range", (unsigned long long)idx[q]);
        uint64_t k = n + idx[q];
        for (uint32_t l = 0; l < log_n; ++l) { sib[(uint64_t)q * log_n + l] = k ^ 1ull; k >>= 1; }
    }
    const size_t need = total * 8 + total * 32 + total;
    ss_status st = ctx->ensure_scratch(need);
    if (st != SS_OK) return st;
    uint64_t *d_idx = (uint64_t *)ctx->scratch;
    uint8_t *d_out = (uint8_t *)ctx->scratch + total * 8;
    uint8_t *d_otags = d_out + total * 32;
    HIP_TRY(hipMemcpyAsync(d_idx, sib.data(), total * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(launch_gather32(ctx->stream, d_nodes, d_idx, total, d_out));
    HIP_TRY(hipMemcpyAsync(out, d_out, total * 32, hipMemcpyDeviceToHost, ctx->stream));
    if (out_tags) {
        if (d_tags) {
            HIP_TRY(launch_gather8(ctx->stream, d_tags, d_idx, total, d_otags));
            HIP_TRY(hipMemcpyAsync(out_tags, d_otags, total, hipMemcpyDeviceToHost, ctx->stream));
        } else {
            memset(out_tags, 0, total);
        }
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SS_OK;
}

ss_status ss_gather_rows(ss_ctx *ctx, const uint64_t *const *d_cols, uint32_t ncols, const uint64_t *idx,
                         uint32_t nidx, uint64_t *out) {
    if (!ctx || !d_cols || (!idx && nidx) || (!out && nidx)) return fail(SS_ERR_INVALID, "NULL argument");
    if (nidx == 0 || ncols == 0) return SS_OK;
    if (has_null((const void *const *)d_cols, ncols)) return fail(SS_ERR_INVALID, "NULL column");
    const size_t idx_bytes = ((size_t)nidx * 8 + 31) & ~(size_t)31;      // the rows behind the indices on a 32-byte boundary (uint4 stores)
    const size_t need = idx_bytes + (size_t)nidx * ncols * 32;
    ss_status st = ctx->ensure_scratch(need);
    if (st != SS_OK) return st;
    uint64_t *d_idx = (uint64_t *)ctx->scratch;
    uint8_t *d_out = (uint8_t *)ctx->scratch + idx_bytes;
    HIP_TRY(hipMemcpyAsync(d_idx, idx, (size_t)nidx * 8, hipMemcpyHostToDevice, ctx->stream));
    // one launch per MAX_COLS columns, written row after row: the download is the caller's array
    for (uint32_t c0 = 0; c0 < ncols; c0 += (uint32_t)MAX_COLS)
        HIP_TRY(launch_gather32_cols(ctx->stream, (const void *const *)d_cols + c0, std::min<uint32_t>((uint32_t)MAX_COLS, ncols - c0), c0, ncols, d_idx, nidx, d_out));
    HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)nidx * ncols * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SS_OK;
}

ss_status ss_gather_batch(ss_ctx *ctx, const ss_gather_job *jobs, uint32_t njobs) {
    if (!ctx || (!jobs && njobs)) return fail(SS_ERR_INVALID, "NULL argument");
    size_t n_idx = 0, out_bytes = 0;
    for (uint32_t j = 0; j < njobs; ++j) {
        const ss_gather_job &g = jobs[j];
        if (!g.nidx || !g.ncols) continue;
        if (!g.d_cols || !g.idx || !g.out) return fail(SS_ERR_INVALID, "job %u: NULL argument", j);
        if (g.entry_bytes != 32 && !(g.entry_bytes == 1 && g.ncols == 1)) return fail(SS_ERR_INVALID, "job %u: entries of 32 bytes, or of 1 byte from one array", j);
        if (has_null(g.d_cols, g.ncols)) return fail(SS_ERR_INVALID, "job %u: NULL column", j);
        n_idx += g.nidx;
        out_bytes += (((size_t)g.nidx * g.ncols * g.entry_bytes) + 31) & ~(size_t)31;      // every job's output 32-byte aligned
    }
    if (!n_idx) return SS_OK;
    const size_t idx_bytes = (n_idx * 8 + 31) & ~(size_t)31;             // the rows behind the indices on a 32-byte boundary (uint4 stores)
    ss_status st = ctx->ensure_scratch(idx_bytes + out_bytes);
    if (st != SS_OK) return st;
    std::vector<uint64_t> all_idx;
    all_idx.reserve(n_idx);
    for (uint32_t j = 0; j < njobs; ++j)
        if (jobs[j].nidx && jobs[j].ncols) all_idx.insert(all_idx.end(), jobs[j].idx, jobs[j].idx + jobs[j].nidx);
    uint64_t *d_idx = (uint64_t *)ctx->scratch;
    uint8_t *d_out = (uint8_t *)ctx->scratch + idx_bytes;
    HIP_TRY(hipMemcpyAsync(d_idx, all_idx.data(), n_idx * 8, hipMemcpyHostToDevice, ctx->stream));
    size_t io = 0, oo = 0;
    for (uint32_t j = 0; j < njobs; ++j) {
        const ss_gather_job &g = jobs[j];
        if (!g.nidx || !g.ncols) continue;
        if (g.entry_bytes == 1) HIP_TRY(launch_gather8(ctx->stream, (const uint8_t *)g.d_cols[0], d_idx + io, g.nidx, d_out + oo));
        else
            for (uint32_t c0 = 0; c0 < g.ncols; c0 += (uint32_t)MAX_COLS)
                HIP_TRY(launch_gather32_cols(ctx->stream, g.d_cols + c0, std::min<uint32_t>((uint32_t)MAX_COLS, g.ncols - c0), c0, g.ncols, d_idx + io, g.nidx, d_out + oo));
        io += g.nidx;
        oo += (((size_t)g.nidx * g.ncols * g.entry_bytes) + 31) & ~(size_t)31;
    }
    std::vector<uint8_t> host(out_bytes);
    HIP_TRY(hipMemcpyAsync(host.data(), d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    oo = 0;
    for (uint32_t j = 0; j < njobs; ++j) {
        const ss_gather_job &g = jobs[j];
        if (!g.nidx || !g.ncols) continue;
        const size_t bytes = (size_t)g.nidx * g.ncols * g.entry_bytes;
        memcpy(g.out, host.data() + oo, bytes);
        oo += (bytes + 31) & ~(size_t)31;
    }
    return SS_OK;
}

// ------------------------------------------------------------------- FRI
ss_status ss_fri_fold(ss_ctx *ctx, const uint64_t *d_evals, uint32_t log_len, uint32_t fold,
                      const uint64_t alpha[4], const uint64_t domain_offset[4], uint64_t *d_out) {
    return ss_fri_fold_ex(ctx, d_evals, log_len, fold, alpha, domain_offset, 0, d_out);
}
static ss_status fri_fold_impl(ss_ctx *ctx, const uint64_t *d_evals, uint32_t log_len, uint32_t fold, const uint64_t alpha[4],
                               const uint64_t domain_offset[4], uint32_t flags, bool whole, uint64_t row0, uint64_t count, uint64_t *d_out) {
    if (!ctx || !d_evals || !alpha || !d_out) return fail(SS_ERR_INVALID, "NULL argument");
    if (flags & ~(uint32_t)(SS_FRI_BITREV_ROWS | SS_FRI_UNNORMALISED)) return fail(SS_ERR_INVALID, "unknown FRI flags %u", flags);
    uint32_t log_fold = 0;
    while ((1u << log_fold) < fold) ++log_fold;
    if ((1u << log_fold) != fold || log_fold < 1 || log_fold > 4) return fail(SS_ERR_INVALID, "fold must be 2, 4, 8 or 16");
    if (!valid_log(log_len) || log_len < log_fold) return fail(SS_ERR_INVALID, "log_len out of range");
    const uint64_t rows = (1ull << log_len) >> log_fold;
    if (whole) { row0 = 0; count = rows; }
    else {
        if (flags & SS_FRI_BITREV_ROWS) return fail(SS_ERR_UNSUPPORTED, "a row range of a layer in bit-reversed order");
        if (row0 > rows || count > rows - row0) return fail(SS_ERR_INVALID, "rows %llu .. +%llu of %llu", (unsigned long long)row0, (unsigned long long)count, (unsigned long long)rows);
        if (!count) return SS_OK;
    }
    const Fp off = domain_offset ? fp_from_limbs64(domain_offset) : fp_one();
    const Fp w_inv = root_of_unity_inv(log_len);
    const Fp wf_inv = root_of_unity_inv(log_fold);
    Fp tw[8];
    tw[0] = fp_one();
    for (int k = 1; k < 8; ++k) tw[k] = fp_mul(tw[k - 1], wf_inv);
    // 1 / x_j = offset^-1 w_len^-e: e = (hi << lo_bits) | lo read from two tables of this layer size (kept by the context)
    const uint32_t row_bits = log_len - log_fold, lo_bits = (row_bits + 1) / 2, hi_bits = row_bits - lo_bits;
    const Fp *pow_tab = nullptr;
    if (row_bits >= 8) {
        const uint32_t key = log_len | (log_fold << 8);          // (the split of the exponent depends on the fold too)
        auto it = ctx->fri_pow_tabs.find(key);
        if (it == ctx->fri_pow_tabs.end()) {
            Fp *tab = nullptr;
            HIP_TRY(ctx->malloc_retry((void **)&tab, ((1ull << lo_bits) + (1ull << hi_bits)) * sizeof(Fp)));
            hipError_t e = launch_fri_pow_table(ctx->stream, tab, w_inv, lo_bits, hi_bits);
            if (e != hipSuccess) { (void)hipFree(tab); return fail(SS_ERR_HIP, "%s", hipGetErrorString(e)); }
            it = ctx->fri_pow_tabs.emplace(key, tab).first;
        }
        pow_tab = it->second;
    }
    ss_ctx::Scope prof(ctx, SS_PROF_FRI);
    HIP_TRY(launch_fri_fold(ctx->stream, (const Fp *)d_evals, log_len, log_fold, fp_from_limbs64(alpha),
                            fp_inv_safegcd(off), w_inv, tw, flags, (Fp *)d_out, row0, count, pow_tab, lo_bits));
    return SS_OK;
}
ss_status ss_fri_fold_ex(ss_ctx *ctx, const uint64_t *d_evals, uint32_t log_len, uint32_t fold,
                         const uint64_t alpha[4], const uint64_t domain_offset[4], uint32_t flags, uint64_t *d_out) {
    return fri_fold_impl(ctx, d_evals, log_len, fold, alpha, domain_offset, flags, true, 0, 0, d_out);
}
ss_status ss_fri_fold_rows(ss_ctx *ctx, const uint64_t *d_evals, uint32_t log_len, uint32_t fold, const uint64_t alpha[4],
                           const uint64_t domain_offset[4], uint32_t flags, uint64_t row0, uint64_t count, uint64_t *d_out) {
    return fri_fold_impl(ctx, d_evals, log_len, fold, alpha, domain_offset, flags, false, row0, count, d_out);
}

// ------------------------------------------------------------------- PoW
ss_status ss_pow_grind(ss_ctx *ctx, int coin_kind, const uint8_t digest[32], uint32_t bits, uint64_t *nonce_out) {
    if (!ctx || !digest || !nonce_out) return fail(SS_ERR_INVALID, "NULL argument");
    if (coin_kind != SS_COIN_SOLIDITY && coin_kind != SS_COIN_CAIRO) return fail(SS_ERR_INVALID, "bad coin kind");
    if (bits > 48) return fail(SS_ERR_UNSUPPORTED, "proof-of-work bits %u > 48", bits);
    uint64_t *d_prefix = ctx->d_small;
    unsigned long long *d_best = (unsigned long long *)(ctx->d_small + 8);
    HIP_TRY(launch_pow_prefix(ctx->stream, coin_kind, digest, bits, d_prefix));
    const uint64_t window = 1ull << 22;
    for (uint64_t start = 1;; start += window) {
        unsigned long long best = ~0ull;
        HIP_TRY(hipMemcpyAsync(d_best, &best, 8, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(launch_pow_grind(ctx->stream, coin_kind, d_prefix, bits, start, window, d_best));
        HIP_TRY(hipMemcpyAsync(&best, d_best, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (best != ~0ull) { *nonce_out = best; return SS_OK; }
        if (start > (1ull << 62)) return fail(SS_ERR_INVALID, "no nonce found");
    }
}

// ------------------------------------------------- extension-trace scans
static bool perm_operand_ok(const ss_perm_operand *o) {
    return o && o->d_data && o->stride && o->addr_offset < o->stride && (o->value_offset < 0 || (uint64_t)o->value_offset < o->stride);
}
ss_status ss_permutation_product(ss_ctx *ctx, const ss_perm_operand *num, const ss_perm_operand *den, uint64_t count,
                                 const uint64_t z[4], const uint64_t alpha[4], uint64_t *d_out, uint64_t out_stride,
                                 uint64_t out_offset, uint64_t last_out[4]) {
    if (!ctx || !z || !d_out) return fail(SS_ERR_INVALID, "NULL argument");
    if (!perm_operand_ok(num) || !perm_operand_ok(den)) return fail(SS_ERR_INVALID, "bad permutation operand (NULL column, zero stride or offset >= stride)");
    if ((num->value_offset >= 0 || den->value_offset >= 0) && !alpha) return fail(SS_ERR_INVALID, "alpha is required for (address, value) terms");
    if (count == 0 || count > (1ull << 32)) return fail(SS_ERR_INVALID, "count out of range");
    if (!out_stride || out_offset >= out_stride) return fail(SS_ERR_INVALID, "bad output stride/offset");
    ss_status st = ctx->ensure_scratch(permutation_product_scratch_felts(count) * sizeof(Fp));
    if (st != SS_OK) return st;
    ss_ctx::Scope prof(ctx, SS_PROF_EXT);
    const PermOperand n{(const Fp *)num->d_data, num->stride, num->addr_offset, num->value_offset};
    const PermOperand d{(const Fp *)den->d_data, den->stride, den->addr_offset, den->value_offset};
    HIP_TRY(launch_permutation_product(ctx->stream, n, d, count, fp_from_limbs64(z), alpha ? fp_from_limbs64(alpha) : fp_zero(),
                                       (Fp *)d_out, out_stride, out_offset, (Fp *)ctx->scratch));
    if (last_out) {
        HIP_TRY(hipMemcpyAsync(last_out, (const Fp *)d_out + (count - 1) * out_stride + out_offset, sizeof(Fp), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return SS_OK;
}

ss_status ss_diluted_aggregate(ss_ctx *ctx, const uint64_t *d_ordered, uint64_t stride, uint64_t offset, uint64_t count,
                               const uint64_t z[4], const uint64_t alpha[4], uint64_t *d_out, uint64_t out_stride,
                               uint64_t out_offset) {
    if (!ctx || !d_ordered || !z || !alpha || !d_out) return fail(SS_ERR_INVALID, "NULL argument");
    if (!stride || offset >= stride || !out_stride || out_offset >= out_stride) return fail(SS_ERR_INVALID, "bad stride/offset");
    if (count == 0 || count > (1ull << 32)) return fail(SS_ERR_INVALID, "count out of range");
    ss_status st = ctx->ensure_scratch(diluted_aggregate_scratch_felts(count) * sizeof(Fp));
    if (st != SS_OK) return st;
    ss_ctx::Scope prof(ctx, SS_PROF_EXT);
    HIP_TRY(launch_diluted_aggregate(ctx->stream, (const Fp *)d_ordered, stride, offset, count, fp_from_limbs64(z),
                                     fp_from_limbs64(alpha), (Fp *)d_out, out_stride, out_offset, (Fp *)ctx->scratch));
    return SS_OK;
}

// ---- the same scans over the row blocks of several devices (ABI 12; host/extension.cpp build_extension_blocks)
ss_status ss_scale_strided(ss_ctx *ctx, uint64_t *d_data, uint64_t stride, uint64_t offset, uint64_t count, const uint64_t factor[4]) {
    if (!ctx || !d_data || !factor) return fail(SS_ERR_INVALID, "NULL argument");
    if (!stride || offset >= stride) return fail(SS_ERR_INVALID, "bad stride/offset");
    if (count == 0 || count > (1ull << 32)) return fail(SS_ERR_INVALID, "count out of range");
    ss_ctx::Scope prof(ctx, SS_PROF_EXT);
    HIP_TRY(launch_scale_strided(ctx->stream, (Fp *)d_data, stride, offset, count, fp_from_limbs64(factor)));
    return SS_OK;
}
ss_status ss_diluted_aggregate_block(ss_ctx *ctx, const uint64_t *d_ordered, uint64_t stride, uint64_t offset, uint64_t count,
                                     int starts_column, const uint64_t z[4], const uint64_t alpha[4], uint64_t *d_maps,
                                     uint64_t total_out[8]) {
    if (!ctx || !d_ordered || !z || !alpha || !d_maps) return fail(SS_ERR_INVALID, "NULL argument");
    if (!stride || offset >= stride) return fail(SS_ERR_INVALID, "bad stride/offset");
    if (count == 0 || count > (1ull << 32)) return fail(SS_ERR_INVALID, "count out of range");
    ss_status st = ctx->ensure_scratch(scan_agg_felts(count, 2) * sizeof(Fp));
    if (st != SS_OK) return st;
    ss_ctx::Scope prof(ctx, SS_PROF_EXT);
    HIP_TRY(launch_diluted_aggregate_maps(ctx->stream, (const Fp *)d_ordered, stride, offset, count, starts_column != 0, fp_from_limbs64(z),
                                          fp_from_limbs64(alpha), (Fp *)d_maps, (Fp *)ctx->scratch));
    if (total_out) {
        HIP_TRY(hipMemcpyAsync(total_out, (const Fp *)d_maps + 2 * (count - 1), 2 * sizeof(Fp), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return SS_OK;
}
ss_status ss_affine_apply(ss_ctx *ctx, const uint64_t *d_maps, uint64_t count, const uint64_t start[4], uint64_t *d_out,
                          uint64_t out_stride, uint64_t out_offset) {
    if (!ctx || !d_maps || !start || !d_out) return fail(SS_ERR_INVALID, "NULL argument");
    if (!out_stride || out_offset >= out_stride) return fail(SS_ERR_INVALID, "bad output stride/offset");
    if (count == 0 || count > (1ull << 32)) return fail(SS_ERR_INVALID, "count out of range");
    ss_ctx::Scope prof(ctx, SS_PROF_EXT);
    HIP_TRY(launch_affine_apply(ctx->stream, (const Fp *)d_maps, count, fp_from_limbs64(start), (Fp *)d_out, out_stride, out_offset));
    return SS_OK;
}

// -------------------------------------------------------------- the base trace on the device (ABI 11; trace.hip)
namespace {
static_assert(SS_TRACE_ERR_MISSING_CELL == TRACE_ERR_MISSING_CELL && SS_TRACE_ERR_FILL == TRACE_ERR_FILL && SS_TRACE_ERR_NOT_CONTINUOUS == TRACE_ERR_NOT_CONTINUOUS &&
              SS_TRACE_ERR_PUBLIC_CELLS == TRACE_ERR_PUBLIC_CELLS && SS_TRACE_STATUS_WORDS == TRACE_ST_WORDS && SS_TRACE_NPC_OP1 == TRACE_NPC_OP1 &&
              SS_TRACE_RC_OFF_OP1 == TRACE_RC_OFF_OP1 && SS_TRACE_AUX_RES == TRACE_AUX_RES && SS_TRACE_CELL_ADDRESS == TRACE_TILE_ADDRESS,
              "the header's constants are the kernels'");
bool trace_layout_ok(const ss_trace_layout *l) {
    if (!l) return false;
    for (int j = 0; j < 8; ++j) if (l->npc_pair[j] > SS_TRACE_NPC_OP1) return false;
    for (int o = 0; o < 16; ++o) if (l->rc_cell[o] > SS_TRACE_RC_OFF_OP1 || l->aux_cell[o] > SS_TRACE_AUX_RES) return false;
    return true;
}
TraceRcPlan rc_plan_of(const ss_trace_rc_plan *p) {
    TraceRcPlan q;
    q.n_slots = p->n_slots; q.n_given = p->n_given; q.slot_rows = p->slot_rows; q.addr_begin = p->addr_begin; q.n_padding = p->n_padding; q.pad0 = p->pad0;
    q.part_stride = p->part_stride; q.part_off = p->part_off; q.pair_off = p->pair_off; q.rc_lo = p->rc_lo; q.rc_hi = p->rc_hi;
    q.ordered_step = p->ordered_step; q.ordered_off = p->ordered_off; q.unused_off = p->unused_off;
    return q;
}
}  // namespace
ss_status ss_trace_memory_image(ss_ctx *ctx, const uint64_t *d_records, uint64_t n_records, uint64_t *d_image, uint64_t cells) {
    if (!ctx || !d_image || (n_records && !d_records)) return fail(SS_ERR_INVALID, "NULL argument");
    if (!cells || cells > (1ull << 33)) return fail(SS_ERR_INVALID, "memory image of %llu cells", (unsigned long long)cells);
    ss_ctx::Scope prof(ctx, SS_PROF_TRACE);
    HIP_TRY(launch_trace_memory_image(ctx->stream, d_records, n_records, d_image, cells));
    return SS_OK;
}
ss_status ss_trace_cpu_cells(ss_ctx *ctx, const ss_trace_layout *layout, const uint64_t *d_states, uint64_t num_cycles, const uint64_t *d_image,
                             uint64_t cells, const uint64_t pad_value[4], uint64_t rc_fill, uint64_t *d_flags, uint64_t *d_pool, uint64_t *d_range_check,
                             uint64_t *d_auxiliary, uint32_t *d_pool_addr, uint32_t *d_status) {
    if (!ctx || !d_states || !d_image || !pad_value || !d_flags || !d_pool || !d_range_check || !d_auxiliary || !d_pool_addr || !d_status)
        return fail(SS_ERR_INVALID, "NULL argument");
    if (!trace_layout_ok(layout)) return fail(SS_ERR_INVALID, "bad trace layout (a cell kind out of range)");
    if (!num_cycles || num_cycles > (1ull << 28)) return fail(SS_ERR_INVALID, "num_cycles out of range");
    TraceLayout L;
    memcpy(L.npc_pair, layout->npc_pair, 8); memcpy(L.rc_cell, layout->rc_cell, 16); memcpy(L.aux_cell, layout->aux_cell, 16);
    ss_ctx::Scope prof(ctx, SS_PROF_TRACE);
    HIP_TRY(launch_trace_cpu(ctx->stream, L, d_states, num_cycles, d_image, cells, fp_from_limbs64(pad_value), rc_fill, (Fp *)d_flags, (Fp *)d_pool,
                             (Fp *)d_range_check, (Fp *)d_auxiliary, d_pool_addr, d_status));
    return SS_OK;
}
ss_status ss_trace_builtin(ss_ctx *ctx, uint64_t *const *d_cols, uint32_t ncols, const ss_trace_cell *d_cells, uint32_t n_cells, const uint64_t *d_values,
                           uint32_t n_templates, const uint32_t *d_template_of_block, uint64_t n_blocks, uint64_t block_rows, uint64_t addr_begin,
                           uint64_t addr_per_block, uint32_t *d_pool_addr) {
    if (!ctx || !d_cols || !d_cells || !d_values || !d_pool_addr) return fail(SS_ERR_INVALID, "NULL argument");
    if (!ncols || ncols > (uint32_t)MAX_COLS) return fail(SS_ERR_INVALID, "ncols out of range");
    if (!n_templates || !block_rows) return fail(SS_ERR_INVALID, "no template / empty blocks");
    if (n_blocks > (1ull << 32) || (double)n_blocks * n_cells > 1e12) return fail(SS_ERR_INVALID, "too many cells");
    ColPtrs cp{};
    for (uint32_t c = 0; c < ncols; ++c) { if (!d_cols[c]) return fail(SS_ERR_INVALID, "NULL column"); cp.dst[c] = d_cols[c]; }
    static_assert(sizeof(ss_trace_cell) == sizeof(TraceTileEntry), "ss_trace_cell is TraceTileEntry");
    ss_ctx::Scope prof(ctx, SS_PROF_TRACE);
    HIP_TRY(launch_trace_tile(ctx->stream, cp, ncols, (const TraceTileEntry *)d_cells, n_cells, (const Fp *)d_values, n_templates, d_template_of_block, n_blocks,
                              block_rows, addr_begin, addr_per_block, d_pool_addr));
    return SS_OK;
}
namespace {
bool rc_plan_ok(const ss_trace_rc_plan *p) {
    return p && p->rc_lo <= p->rc_hi && p->rc_hi < 65536 && (p->ordered_step == 1 || p->ordered_step == 2 || p->ordered_step == 4 || p->ordered_step == 8 || p->ordered_step == 16) &&
           p->ordered_off < p->ordered_step && p->unused_off < 16 && p->n_given <= p->n_slots && p->n_padding <= 65536 &&
           (!p->n_slots || (p->part_off + 7ull * p->part_stride < p->slot_rows && p->pair_off + 1ull < p->slot_rows));
}
}  // namespace
ss_status ss_trace_rc_pool(ss_ctx *ctx, const ss_trace_rc_plan *plan, const uint32_t *d_first, const uint16_t *d_padding, uint64_t num_cycles,
                           uint64_t *d_range_check) {
    if (!ctx || !d_first || !d_range_check || (plan && plan->n_padding && !d_padding)) return fail(SS_ERR_INVALID, "NULL argument");
    if (!rc_plan_ok(plan)) return fail(SS_ERR_INVALID, "bad range-check plan");
    if (!num_cycles || num_cycles > (1ull << 28)) return fail(SS_ERR_INVALID, "num_cycles out of range");
    ss_ctx::Scope prof(ctx, SS_PROF_TRACE);
    HIP_TRY(launch_trace_rc_pool(ctx->stream, rc_plan_of(plan), d_first, d_padding, num_cycles, (Fp *)d_range_check));
    return SS_OK;
}
ss_status ss_trace_rc_builtin(ss_ctx *ctx, const ss_trace_rc_plan *plan, const uint64_t *d_given, const uint16_t *d_padding, uint64_t *d_range_check,
                              uint64_t *d_pool, uint32_t *d_pool_addr) {
    if (!ctx || !d_range_check || !d_pool || !d_pool_addr || (plan && plan->n_given && !d_given) || (plan && plan->n_padding && !d_padding))
        return fail(SS_ERR_INVALID, "NULL argument");
    if (!rc_plan_ok(plan)) return fail(SS_ERR_INVALID, "bad range-check plan");
    ss_ctx::Scope prof(ctx, SS_PROF_TRACE);
    HIP_TRY(launch_trace_rc_builtin(ctx->stream, rc_plan_of(plan), d_given, d_padding, (Fp *)d_range_check, (Fp *)d_pool, d_pool_addr));
    return SS_OK;
}
ss_status ss_trace_ordered_runs(ss_ctx *ctx, uint64_t *d_col, uint64_t stride, uint64_t offset, uint64_t slots, const uint32_t *d_first,
                                uint32_t n_values, uint32_t lo, int diluted) {
    if (!ctx || !d_col || !d_first) return fail(SS_ERR_INVALID, "NULL argument");
    if (!stride || offset >= stride || !n_values || (uint64_t)lo + n_values > (1ull << 32)) return fail(SS_ERR_INVALID, "bad stride / offset / value range");
    if (diluted && (uint64_t)lo + n_values > 65536) return fail(SS_ERR_INVALID, "diluted values have 16 bits");
    ss_ctx::Scope prof(ctx, SS_PROF_TRACE);
    HIP_TRY(launch_trace_runs(ctx->stream, (Fp *)d_col, stride, offset, slots, d_first, n_values, lo, diluted != 0));
    return SS_OK;
}
ss_status ss_trace_patch(ss_ctx *ctx, uint64_t *d_col, uint64_t col_rows, const uint64_t *d_rows, const uint64_t *d_values, uint64_t count) {
    if (!ctx || !d_col || (count && (!d_rows || !d_values))) return fail(SS_ERR_INVALID, "NULL argument");
    ss_ctx::Scope prof(ctx, SS_PROF_TRACE);
    HIP_TRY(launch_trace_patch(ctx->stream, (Fp *)d_col, col_rows, d_rows, d_values, count));
    return SS_OK;
}
ss_status ss_trace_ordered_memory(ss_ctx *ctx, uint64_t n, uint64_t *d_pool, uint64_t *d_memory, uint32_t *d_pool_addr, const uint32_t *d_public_addr,
                                  const uint64_t *d_public_value, uint32_t n_public, uint64_t public_cells, const uint64_t pad_value[4],
                                  uint32_t unused_off, uint32_t *d_status) {
    if (!ctx || !d_pool || !d_memory || !d_pool_addr || !pad_value || !d_status || (n_public && (!d_public_addr || !d_public_value)))
        return fail(SS_ERR_INVALID, "NULL argument");
    if (n < 16 || (n & (n - 1)) || n > (1ull << 32)) return fail(SS_ERR_INVALID, "n must be a power of two in [16, 2^32]");
    if (unused_off >= 16 || (unused_off & 1)) return fail(SS_ERR_INVALID, "the unused pair starts at an even row of a cycle");
    if (n_public > public_cells) return fail(SS_ERR_INVALID, "public memory does not fit its cells");
    ss_status st = ctx->ensure_scratch(trace_memory_scratch_words(n / 2) * sizeof(uint32_t));
    if (st != SS_OK) return st;
    TraceMemoryArgs m;
    m.n = n; m.npc = (Fp *)d_pool; m.memory = (Fp *)d_memory; m.d_pool_addr = d_pool_addr; m.d_public_addr = d_public_addr;
    m.d_public_value = (const Fp *)d_public_value; m.n_public = n_public; m.public_cells = public_cells; m.pad_value = fp_from_limbs64(pad_value);
    m.unused_off = unused_off; m.d_status = d_status;
    ss_ctx::Scope prof(ctx, SS_PROF_TRACE);
    HIP_TRY(launch_trace_ordered_memory(ctx->stream, m, (uint32_t *)ctx->scratch));
    return SS_OK;
}
ss_status ss_trace_status(ss_ctx *ctx, const uint32_t *d_status, uint32_t *status_out) {
    if (!ctx || !d_status || !status_out) return fail(SS_ERR_INVALID, "NULL argument");
    HIP_TRY(hipMemcpyAsync(status_out, d_status, SS_TRACE_STATUS_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SS_OK;
}

// -------------------------------------------------------------- Pedersen
ss_status ss_pedersen_hash(ss_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, uint64_t n, uint64_t *d_out) {
    if (!ctx || !d_a || !d_b || !d_out) return fail(SS_ERR_INVALID, "NULL argument");
    if (!ctx->ped) HIP_TRY(pedersen_tables_create(ctx->stream, &ctx->ped));
    ss_status st = ctx->ensure_scratch(PEDERSEN_TMP_FELTS_PER_HASH * n * sizeof(Fp));
    if (st != SS_OK) return st;
    HIP_TRY(launch_pedersen_felts(ctx->stream, ctx->ped, (const Fp *)d_a, (const Fp *)d_b, n, (Fp *)d_out, (Fp *)ctx->scratch));
    return SS_OK;
}

// host Keccak-256 (Keccak team padding 0x01), for the Fiat-Shamir coin only
namespace {
const uint64_t HOST_KRC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull,
    0x000000000000808bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
    0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
    0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull,
    0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
const int HOST_KPILN[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
const int HOST_KROTC[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
void host_keccak_f(uint64_t st[25]) {
    for (int r = 0; r < 24; ++r) {
        uint64_t bc[5];
        for (int i = 0; i < 5; ++i) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; ++i) {
            const uint64_t t = bc[(i + 4) % 5] ^ ((bc[(i + 1) % 5] << 1) | (bc[(i + 1) % 5] >> 63));
            for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
        }
        uint64_t t = st[1];
        for (int i = 0; i < 24; ++i) {
            const int j = HOST_KPILN[i];
            const uint64_t b = st[j];
            st[j] = (t << HOST_KROTC[i]) | (t >> (64 - HOST_KROTC[i]));
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; ++i) bc[i] = st[j + i];
            for (int i = 0; i < 5; ++i) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= HOST_KRC[r];
    }
}
}  // namespace

ss_status ss_keccak256_host(const uint8_t *msg, size_t len, uint8_t out[32]) {
    if ((!msg && len) || !out) return fail(SS_ERR_INVALID, "NULL argument");
    uint64_t st[25] = {0};
    const size_t rate = 136;
    while (len >= rate) {
        for (int i = 0; i < 17; ++i) { uint64_t w; memcpy(&w, msg + 8 * i, 8); st[i] ^= w; }
        host_keccak_f(st);
        msg += rate; len -= rate;
    }
    uint8_t blk[136] = {0};
    if (len) memcpy(blk, msg, len);
    blk[len] ^= 0x01;
    blk[135] ^= 0x80;
    for (int i = 0; i < 17; ++i) { uint64_t w; memcpy(&w, blk + 8 * i, 8); st[i] ^= w; }
    host_keccak_f(st);
    memcpy(out, st, 32);
    return SS_OK;
}

ss_status ss_pedersen_hash_host(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
    if (!a || !b || !out) return fail(SS_ERR_INVALID, "NULL argument");
    const Fp r = pedersen_hash_host(fp_from_limbs64(a), fp_from_limbs64(b));
    for (int i = 0; i < 4; ++i) out[i] = (uint64_t)r.v[2 * i] | ((uint64_t)r.v[2 * i + 1] << 32);
    return SS_OK;
}

ss_status ss_fp252_mul_bench(ss_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, uint64_t n, uint32_t reps,
                             uint64_t *d_out) {
    if (!ctx || !d_a || !d_b || !d_out) return fail(SS_ERR_INVALID, "NULL argument");
    HIP_TRY(launch_mul_bench(ctx->stream, (const Fp *)d_a, (const Fp *)d_b, (Fp *)d_out, n, reps));
    return SS_OK;
}

// ------------------------------------------------------------------- D1
ss_status ss_poly_eval(ss_ctx *ctx, const uint64_t *const *d_coeffs, uint32_t ncols, uint32_t log_n,
                       const uint64_t x[4], uint64_t *out) {
    if (!ctx || !d_coeffs || !x || !out) return fail(SS_ERR_INVALID, "NULL argument");
    if (!valid_log(log_n)) return fail(SS_ERR_INVALID, "log_n out of range");
    if (ncols == 0) return SS_OK;
    if (ncols > (uint32_t)MAX_COLS) return fail(SS_ERR_UNSUPPORTED, "ncols %u > %d", ncols, MAX_COLS);
    if (has_null((const void *const *)d_coeffs, ncols)) return fail(SS_ERR_INVALID, "NULL column");
    // x^(2^j), j < log_n
    std::vector<Fp> xp(log_n);
    xp[0] = fp_from_limbs64(x);
    for (uint32_t j = 1; j < log_n; ++j) xp[j] = fp_sqr(xp[j - 1]);
    const uint64_t n = 1ull << log_n;
    const size_t per_col = (size_t)((n >> 1) + (n >> 2) + 2) * sizeof(Fp);      // two ping-pong regions
    ss_status st = ctx->ensure_scratch(per_col * ncols);
    if (st != SS_OK) return st;
    const void *in[MAX_COLS];
    void *bufa[MAX_COLS], *bufb[MAX_COLS];
    for (uint32_t c = 0; c < ncols; ++c) {
        in[c] = d_coeffs[c];
        bufa[c] = (char *)ctx->scratch + per_col * c;
        bufb[c] = (char *)bufa[c] + ((n >> 1) + 1) * sizeof(Fp);
    }
    uint32_t lc = log_n;
    bool to_a = true;
    const void *const *src = in;
    while (lc > 0) {
        const uint32_t lv = lc >= 3 ? 3 : lc;
        Fp mult[3];
        for (uint32_t k = 0; k < lv; ++k) mult[k] = xp[lc - 1 - k];
        void *const *dst = to_a ? bufa : bufb;
        HIP_TRY(launch_poly_reduce(ctx->stream, src, dst, ncols, 1ull << lc, lv, mult));
        src = (const void *const *)dst;
        to_a = !to_a;
        lc -= lv;
    }
    for (uint32_t c = 0; c < ncols; ++c)
        HIP_TRY(hipMemcpyAsync(out + 4 * c, src[c], sizeof(Fp), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SS_OK;
}

static ss_status ood_eval_sparse(ss_ctx *ctx, const uint64_t *const *d_coeffs, uint32_t ncols, uint32_t log_n,
                                 const uint32_t *mask_col, const uint32_t *mask_off, uint32_t nmask, const uint64_t z[4], uint64_t *out);

ss_status ss_ood_eval(ss_ctx *ctx, const uint64_t *const *d_coeffs, uint32_t ncols, uint32_t log_n,
                      const uint32_t *mask_col, const uint32_t *mask_off, uint32_t nmask, const uint64_t z[4],
                      uint64_t *out) {
    if (!ctx || !d_coeffs || !z || (nmask && (!mask_col || !mask_off || !out))) return fail(SS_ERR_INVALID, "NULL argument");
    if (!valid_log(log_n)) return fail(SS_ERR_INVALID, "log_n out of range");
    if (ncols == 0 || ncols > (uint32_t)MAX_COLS) return fail(SS_ERR_UNSUPPORTED, "ncols %u not in [1,%d]", ncols, MAX_COLS);
    if (nmask == 0) return SS_OK;
    if (has_null((const void *const *)d_coeffs, ncols)) return fail(SS_ERR_INVALID, "NULL column");
    for (uint32_t j = 0; j < nmask; ++j)
        if (mask_col[j] >= ncols) return fail(SS_ERR_INVALID, "mask cell %u names column %u", j, mask_col[j]);
    // A mask names few points per column (269 cells over 10 columns for starknet): large columns are evaluated point by point
    // (deep.hip: blocks transformed in registers + fused Horner trees), a fifth of the arithmetic of the transforms below - from
    // 2^20 coefficients on: below, the descriptors and level multipliers of every point (host work, ~0.3 ms per call in 64-bit
    // limbs) and the fold launches cost what ten small transforms do (measured at 2^18 and 2^20, gpurun_out/r03_call16).
    // SS_OOD_SPARSE_MIN_LOG moves the threshold (the parity tests run the point-by-point path from 2^12), SS_OOD_TRANSFORM=1
    // keeps the transforms.
    uint32_t sparse_min_log = 20;
    if (const char *e = getenv("SS_OOD_SPARSE_MIN_LOG")) sparse_min_log = (uint32_t)strtoul(e, nullptr, 10);
    if (sparse_min_log < 12) sparse_min_log = 12;
    if (log_n >= sparse_min_log && !getenv("SS_OOD_TRANSFORM")) return ood_eval_sparse(ctx, d_coeffs, ncols, log_n, mask_col, mask_off, nmask, z, out);
    // T_c(z w^k) for every k at once: one coset NTT (offset z) of each coefficient column
    const uint64_t n = 1ull << log_n;
    const size_t col_bytes = sizeof(Fp) << log_n;
    const size_t small = (size_t)nmask * (4 + 8 + 32);
    ss_status st = ctx->ensure_scratch(col_bytes * ncols + small);
    if (st != SS_OK) return st;
    const Fp *tw = nullptr;
    st = ctx->get_transient_plan(log_n, false, fp_from_limbs64(z), &tw);
    if (st != SS_OK) return st;
    ColPtrs cols;
    memset(&cols, 0, sizeof cols);
    const void *evs[MAX_COLS];
    for (uint32_t c = 0; c < ncols; ++c) {
        cols.src[c] = d_coeffs[c];
        cols.dst[c] = (char *)ctx->scratch + col_bytes * c;
        evs[c] = cols.dst[c];
    }
    st = run_forward(ctx, cols, ncols, log_n, tw, 0);
    if (st != SS_OK) return st;
    char *sm = (char *)ctx->scratch + col_bytes * ncols;
    uint64_t *d_idx = (uint64_t *)sm;
    Fp *d_out = (Fp *)(sm + (size_t)nmask * 8);
    uint32_t *d_col = (uint32_t *)(sm + (size_t)nmask * 40);
    std::vector<uint64_t> idx(nmask);
    for (uint32_t j = 0; j < nmask; ++j) idx[j] = mask_off[j] & (n - 1);
    HIP_TRY(hipMemcpyAsync(d_idx, idx.data(), (size_t)nmask * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_col, mask_col, (size_t)nmask * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(launch_gather_cells(ctx->stream, evs, ncols, d_col, d_idx, nmask, d_out));
    HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)nmask * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SS_OK;
}


// T_col(z w_n^k) for the distinct (column, k) of the mask, without a transform of the whole column (deep.hip, "out-of-domain
// evaluation at FEW points").  Per column S = 0 / 2 / 4 stages of the forward network run on blocks of 2^S coefficients -
// more cells, more shared work - then every point folds the R = n >> S block values it needs with a bit-reversed Horner tree.
static ss_status ood_eval_sparse(ss_ctx *ctx, const uint64_t *const *d_coeffs, uint32_t ncols, uint32_t log_n,
                                 const uint32_t *mask_col, const uint32_t *mask_off, uint32_t nmask, const uint64_t z[4], uint64_t *out) {
    const uint64_t n = 1ull << log_n;
    const Fp zf = fp_from_limbs64(z), wn = root_of_unity(log_n);
    // the host's share - level multipliers of every point, ~300 products each - in 64-bit limbs (fp252_host.h)
    Fp two24 = fp_zero(); two24.v[0] = 1u << 24;
    const Fp to_r280 = fp_to_mont(two24);                      // x R -> x 2^280: a product with the image of 2^24, as fl_to_r280
    // ---- the distinct points, column by column
    struct Point { uint32_t col, k, S; Fp x; std::vector<Fp> xpow; Fp *buf[2]; };
    std::vector<Point> pts;
    std::map<std::pair<uint32_t, uint32_t>, uint32_t> index;
    std::vector<std::vector<uint32_t>> of_col(ncols);
    for (uint32_t j = 0; j < nmask; ++j) {
        const uint32_t k = mask_off[j] & (uint32_t)(n - 1);
        if (index.emplace(std::make_pair(mask_col[j], k), (uint32_t)pts.size()).second) {
            Point p;
            p.col = mask_col[j]; p.k = k; p.S = 0; p.x = fph_mul(zf, fph_pow_u64(wn, k)); p.buf[0] = p.buf[1] = nullptr;
            of_col[p.col].push_back((uint32_t)pts.size());
            pts.push_back(p);
        }
    }
    std::vector<uint32_t> S_of(ncols, 0);
    for (uint32_t c = 0; c < ncols; ++c) {
        const size_t cnt = of_col[c].size();
        S_of[c] = cnt <= 3 ? 0u : cnt <= 12 ? 2u : 4u;
        if (const char *e = getenv("SS_OOD_BLOCK_LOG")) S_of[c] = cnt ? (uint32_t)strtoul(e, nullptr, 10) : 0u;    // A/B: 0 .. 4 for every column
        if (S_of[c] > 4) S_of[c] = 4;
        for (uint32_t pi : of_col[c]) pts[pi].S = S_of[c];
    }
    for (Point &p : pts) {                       // x^(2^j), j < log_n - S: the Horner tree's level multipliers
        p.xpow.resize(log_n - p.S);
        p.xpow[0] = p.x;
        for (size_t j = 1; j < p.xpow.size(); ++j) p.xpow[j] = fph_sqr(p.xpow[j - 1]);
    }
    std::vector<Fp> zsq(log_n);                  // z^(2^i)
    zsq[0] = zf;
    for (uint32_t i = 1; i < log_n; ++i) zsq[i] = fph_sqr(zsq[i - 1]);
    // ---- device memory: block outputs, two fold buffers per point, the final values, the descriptors of every launch
    size_t felts = pts.size() + 8;
    std::vector<size_t> y_off(ncols, 0);
    std::vector<uint32_t> res_mask(ncols, 0);
    for (uint32_t c = 0; c < ncols; ++c) {
        if (!S_of[c]) continue;
        for (uint32_t pi : of_col[c]) res_mask[c] |= 1u << (pts[pi].k & ((1u << S_of[c]) - 1u));
        y_off[c] = felts;
        felts += (size_t)__builtin_popcount(res_mask[c]) * (n >> S_of[c]);
    }
    auto first_levels = [&](uint32_t lc) { return lc >= 3 ? 3u : lc; };      // 8 values per lane and launch (16 would not fit 256 registers)
    for (Point &p : pts) {
        const uint32_t lc = log_n - p.S, g1 = first_levels(lc), g2 = first_levels(lc - g1);
        const size_t a = (size_t)1 << (lc - g1), b = (size_t)1 << (lc - g1 - g2);
        p.buf[0] = (Fp *)(uintptr_t)felts; felts += a;                          // (offsets for now: the scratch may move)
        p.buf[1] = (Fp *)(uintptr_t)felts; felts += b;
    }
    const size_t desc_bytes = (size_t)(log_n / 3 + 2) * 3 * (pts.size() * (sizeof(OodFoldPoint) + sizeof(OodFoldArray)) + 64);
    ss_status st = ctx->ensure_scratch(felts * sizeof(Fp) + desc_bytes + 256);
    if (st != SS_OK) return st;
    Fp *base = (Fp *)ctx->scratch, *d_final = base;
    char *d_desc = (char *)(base + felts);
    for (Point &p : pts) { p.buf[0] = base + (uintptr_t)p.buf[0]; p.buf[1] = base + (uintptr_t)p.buf[1]; }
    hipStream_t s = ctx->stream;
    // ---- (A) the blocks
    for (uint32_t c = 0; c < ncols; ++c) {
        const uint32_t S = S_of[c];
        if (!S || of_col[c].empty()) continue;
        OodBlockArgs a;
        memset(&a, 0, sizeof a);
        a.coeffs = (const Fp *)d_coeffs[c]; a.out = base + y_off[c]; a.blocks = n >> S; a.res_mask = res_mask[c];
        for (uint32_t u = 0; u < S; ++u) {       // stage u of the size-n forward network: z^(n / 2^(u+1)) w_n^(j n / 2^(u+1)), j < 2^u
            const Fp h = zsq[log_n - u - 1], r = root_of_unity(u + 1);
            Fp rj = fp_one();
            for (uint32_t j = 0; j < (1u << u); ++j) {
                const Fl t = fl_from_fp(fph_mul(fph_mul(h, rj), to_r280));
                for (int i = 0; i < 9; ++i) a.tw[(1u << u) - 1 + j][i] = t.l[i];
                rj = fph_mul(rj, r);
            }
        }
        ss_ctx::Scope prof(ctx, SS_PROF_DEEP);
        HIP_TRY(launch_ood_blocks(s, (int)S, a));
    }
    // ---- (B) the folds, one group of columns (same S: same lengths) after the other; every launch's descriptors are laid out
    //      first and uploaded in one copy
    std::vector<char> host_desc;
    struct Launch { size_t arrays_at, points_at; uint32_t narrays, max_points; uint64_t in_len; uint32_t levels; bool last; };
    std::vector<Launch> launches;
    for (uint32_t S = 0; S <= 4; ++S) {
        std::vector<uint32_t> group;
        for (uint32_t pi = 0; pi < pts.size(); ++pi) if (pts[pi].S == S) group.push_back(pi);
        if (group.empty()) continue;
        uint32_t lc = log_n - S;
        int round = 0;
        while (lc > 0) {
            const uint32_t g = first_levels(lc);
            const bool last = lc == g;
            std::vector<OodFoldArray> arrays;
            std::vector<OodFoldPoint> points;
            auto add_point = [&](uint32_t pi) {
                const Point &p = pts[pi];
                OodFoldPoint fp_;
                memset(&fp_, 0, sizeof fp_);
                fp_.out = last ? d_final + pi : p.buf[round & 1];
                for (uint32_t t = 0; t < (1u << g); ++t) {       // element t = sum t_i 2^i of a group: weight prod_i (x^(2^(lc-1-i)))^(t_i)
                    Fp wgt = to_r280;                              // weights are multipliers of the fold: R280 form (fl252.h)
                    for (uint32_t i = 0; i < g; ++i) if ((t >> i) & 1u) wgt = fph_mul(wgt, p.xpow[lc - 1 - i]);
                    const Fl l = fl_from_fp(wgt);
                    for (int q = 0; q < 9; ++q) fp_.coef[t][q] = l.l[q];
                }
                points.push_back(fp_);
            };
            if (round == 0) {                    // inputs: the coefficient column (S = 0) or one block-output array per residue
                for (uint32_t c = 0; c < ncols; ++c) {
                    if (S_of[c] != S || of_col[c].empty()) continue;
                    if (S == 0) {
                        arrays.push_back({(const Fp *)d_coeffs[c], (uint32_t)points.size(), (uint32_t)of_col[c].size()});
                        for (uint32_t pi : of_col[c]) add_point(pi);
                        continue;
                    }
                    uint32_t slot = 0;
                    for (uint32_t res = 0; res < (1u << S); ++res) {
                        if (!((res_mask[c] >> res) & 1u)) continue;
                        const uint32_t first = (uint32_t)points.size();
                        for (uint32_t pi : of_col[c]) if ((pts[pi].k & ((1u << S) - 1u)) == res) add_point(pi);
                        arrays.push_back({base + y_off[c] + (size_t)slot * (n >> S), first, (uint32_t)points.size() - first});
                        ++slot;
                    }
                }
            } else {
                for (uint32_t pi : group) {
                    arrays.push_back({pts[pi].buf[(round - 1) & 1], (uint32_t)points.size(), 1u});
                    add_point(pi);
                }
            }
            Launch L;
            L.narrays = (uint32_t)arrays.size(); L.in_len = 1ull << lc; L.levels = g; L.last = last;
            L.max_points = 1;
            for (const OodFoldArray &ar : arrays) if (ar.npoints > L.max_points) L.max_points = ar.npoints;
            host_desc.resize((host_desc.size() + 63) / 64 * 64);
            L.arrays_at = host_desc.size();
            host_desc.insert(host_desc.end(), (const char *)arrays.data(), (const char *)(arrays.data() + arrays.size()));
            host_desc.resize((host_desc.size() + 63) / 64 * 64);
            L.points_at = host_desc.size();
            host_desc.insert(host_desc.end(), (const char *)points.data(), (const char *)(points.data() + points.size()));
            launches.push_back(L);
            lc -= g;
            ++round;
        }
    }
    if (host_desc.size() > desc_bytes) return fail(SS_ERR_INVALID, "out-of-domain descriptors: %zu bytes for %zu reserved", host_desc.size(), desc_bytes);
    HIP_TRY(hipMemcpyAsync(d_desc, host_desc.data(), host_desc.size(), hipMemcpyHostToDevice, s));
    for (const Launch &L : launches) {
        ss_ctx::Scope prof(ctx, SS_PROF_DEEP);
        HIP_TRY(launch_ood_fold(s, (const OodFoldArray *)(d_desc + L.arrays_at), L.narrays, L.max_points, (const OodFoldPoint *)(d_desc + L.points_at), L.in_len,
                                L.levels, L.last));
    }
    std::vector<Fp> vals(pts.size());
    HIP_TRY(hipMemcpyAsync(vals.data(), d_final, pts.size() * sizeof(Fp), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));            // (also keeps host_desc alive until its copy is done)
    for (uint32_t j = 0; j < nmask; ++j) {
        const Fp &v = vals[index[std::make_pair(mask_col[j], mask_off[j] & (uint32_t)(n - 1))]];
        for (int q = 0; q < 4; ++q) out[4 * j + q] = (uint64_t)v.v[2 * q] | ((uint64_t)v.v[2 * q + 1] << 32);
    }
    return SS_OK;
}

}  // extern "C"

namespace {
// The DEEP polynomial's values at the sub-coset points offset * w_n^m, m0 <= m < m0 + count, into d_sub[0 .. count).
// The column pointers name LDE row (m0 << log_blowup): whole columns with m0 = 0 (ss_deep_compose) or a rank's row block
// (ss_deep_compose_rows).  block != 0: the inverse table is built for this range only (with the rows the group shifts
// reach in front of it) instead of over the whole sub-coset.
ss_status deep_subcoset(ss_ctx *ctx, const uint64_t *const *d_trace_lde, uint32_t ntrace_cols,
                        const uint64_t *const *d_comp_lde, uint32_t ncomp, uint32_t log_n, uint32_t log_blowup,
                        const uint64_t offset[4], const uint32_t *mask_col, const uint32_t *mask_off,
                        uint32_t nmask, const uint64_t *ood_trace, const uint64_t *coeff_trace,
                        const uint64_t *ood_comp, const uint64_t *coeff_comp, const uint64_t z[4],
                        uint64_t m0, uint64_t count, bool block, Fp *d_sub /* block: where the values go */,
                        Fp **sub_out /* !block: the n sub-coset values, in scratch2 */) {
    const uint64_t n = 1ull << log_n;
    const Fp off = offset ? fp_from_limbs64(offset) : fp_one();
    const Fp zf = fp_from_limbs64(z);
    const Fp wn_inv = root_of_unity_inv(log_n);
    // Taps of the denominator table, column by column (deep.hip): cell (col, off) reads D at shift `off` with coefficient
    // c' = coeff * w_n^-off (R280 form: times 2^24); one more "column" of constants carries -K_off = -sum_{cells at off} c' * ood.
    Fp two24 = fp_zero(); two24.v[0] = 1u << 24;
    const Fp r280_factor = fp_to_mont(two24);
    std::vector<uint32_t> order(nmask);
    for (uint32_t j = 0; j < nmask; ++j) {
        if (mask_col[j] >= ntrace_cols) return fail(SS_ERR_INVALID, "mask cell %u names column %u", j, mask_col[j]);
        order[j] = j;
    }
    auto off_of = [&](uint32_t j) { return mask_off[j] & (uint32_t)(n - 1); };
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        return mask_col[x] != mask_col[y] ? mask_col[x] < mask_col[y] : off_of(x) < off_of(y);
    });
    // ---- which cells go the rational way (deep.hip, "the mask's LARGE columns as rational functions"): whole-domain calls from
    // 2^20 points on (below, five pruned transforms cost more than the taps they replace), the up to DEEP_RATIONAL_MAX_COLS
    // columns with >= 12 cells, and with them the constants' column; SS_DEEP_RATIONAL_MIN_LOG moves the threshold (the parity
    // tests run this path from 2^10), SS_DEEP_TAPS=1 keeps every cell a tap.  The row-block form evaluates the polynomials on the
    // whole sub-coset too (a transform has no cheap R-th) and uses its range of them: five pruned transforms still cost a rank
    // less than its share of 412 taps, on 8 ranks as on one.
    std::vector<uint32_t> cells_of(ntrace_cols, 0);
    std::map<uint32_t, uint32_t> off_index;                    // distinct offset -> its index (ascending)
    for (uint32_t j = 0; j < nmask; ++j) { cells_of[mask_col[j]] += 1; off_index.emplace(off_of(j), 0u); }
    { uint32_t k = 0; for (auto &kv : off_index) kv.second = k++; }
    uint32_t rat_min_log = 20;
    if (const char *e = getenv("SS_DEEP_RATIONAL_MIN_LOG")) rat_min_log = (uint32_t)strtoul(e, nullptr, 10);
    std::vector<int> rat_slot(ntrace_cols, -1);                // column -> its slot among the rational columns
    std::vector<uint32_t> rat_cols;
    uint32_t poly_log = 0;                                     // the polynomials' coefficient arrays: 2^poly_log entries
    bool rational = nmask && log_n >= rat_min_log && !getenv("SS_DEEP_TAPS");
    if (rational) {
        std::vector<uint32_t> by_size;
        uint32_t min_cells = 12;                              // a polynomial costs what ~ 10 taps do (measured: gpurun_out/r03_call23)
        if (const char *e = getenv("SS_DEEP_RATIONAL_MIN_CELLS")) min_cells = (uint32_t)strtoul(e, nullptr, 10);
        for (uint32_t c = 0; c < ntrace_cols; ++c) if (cells_of[c] >= min_cells) by_size.push_back(c);
        std::stable_sort(by_size.begin(), by_size.end(), [&](uint32_t x, uint32_t y) { return cells_of[x] > cells_of[y]; });
        if (by_size.size() > DEEP_RATIONAL_MAX_COLS) by_size.resize(DEEP_RATIONAL_MAX_COLS);
        uint32_t moved = (uint32_t)off_index.size();
        for (uint32_t c : by_size) moved += cells_of[c];
        while ((1ull << poly_log) < off_index.size() + 1) ++poly_log;
        if (poly_log == 0) poly_log = 1;
        // worth it when the taps moved outnumber what (columns + 2) transforms and the point-wise pass cost
        if (poly_log >= log_n || moved < min_cells * ((uint32_t)by_size.size() + 3u)) rational = false;
        else { rat_cols = by_size; std::sort(rat_cols.begin(), rat_cols.end()); for (uint32_t k = 0; k < rat_cols.size(); ++k) rat_slot[rat_cols[k]] = (int)k; }
    }
    std::map<uint32_t, Fp> wk_of, k_of;                       // per distinct offset: w_n^-off, K_off
    std::vector<uint32_t> tap_shift, cdesc;
    std::vector<Fp> tap_coef;
    uint32_t max_shift = 0;
    for (uint32_t t = 0; t < nmask;) {
        const uint32_t col = mask_col[order[t]], first = (uint32_t)tap_shift.size();
        for (; t < nmask && mask_col[order[t]] == col; ++t) {
            const uint32_t j = order[t], offv = off_of(j);
            auto it = wk_of.find(offv);
            if (it == wk_of.end()) { it = wk_of.emplace(offv, fp_pow_u64(wn_inv, offv)).first; k_of.emplace(offv, fp_zero()); }
            const Fp cprime = fp_mul(fp_from_limbs64(coeff_trace + 4 * j), it->second);
            k_of[offv] = fp_add(k_of[offv], fp_mul(cprime, fp_from_limbs64(ood_trace + 4 * j)));
            if (rat_slot[col] >= 0) continue;
            tap_shift.push_back(offv);
            tap_coef.push_back(fp_mul(cprime, r280_factor));
            if (offv > max_shift) max_shift = offv;
        }
        if ((uint32_t)tap_shift.size() == first) continue;
        cdesc.push_back(col); cdesc.push_back(first); cdesc.push_back((uint32_t)tap_shift.size() - first);
    }
    if (nmask && !rational) {
        const uint32_t first = (uint32_t)tap_shift.size();
        for (auto &kv : k_of) { tap_shift.push_back(kv.first); tap_coef.push_back(fp_neg(kv.second)); }
        cdesc.push_back(0xffffffffu); cdesc.push_back(first); cdesc.push_back((uint32_t)tap_shift.size() - first);
    }
    // ---- the rational part's polynomials, host arithmetic in 64-bit limbs (fp252_host.h): B = prod_o (x - z_o), z_o = z w_n^o;
    // Q_o = B / (x - z_o) by synthetic division; A_c = sum_{cells (c, o)} coeff Q_o, A_K = sum_o K_o Q_o with
    // K_o = sum_{cells at o} coeff * ood (the plain coefficients here: the w_n^-o above belongs to the shifted-table form)
    const uint32_t npoly = rational ? (uint32_t)rat_cols.size() + 2u : 0u;       // A_c ..., A_K, B
    std::vector<Fp> poly_coef((size_t)npoly << poly_log, fp_zero());
    auto build_polynomials = [&]() {             // (~ 1 ms of host arithmetic: run while the tables and the taps are on the GPU)
        const uint32_t d = (uint32_t)off_index.size();
        const size_t plen = (size_t)1 << poly_log;
        std::vector<H4> zo(d), B(d + 1), Q(d);
        std::vector<std::vector<H4>> A(npoly - 1, std::vector<H4>(d, H4{{0, 0, 0, 0}}));
        const H4 zero = H4{{0, 0, 0, 0}}, one = h4_from_fp(fp_one()), zh = h4_from_fp(zf), wn_h = h4_from_fp(root_of_unity(log_n));
        for (auto &kv : off_index) zo[kv.second] = h4_mul(zh, h4_from_fp(fph_pow_u64(h4_to_fp(wn_h), kv.first)));
        B[0] = one;
        for (uint32_t k = 1; k <= d; ++k) B[k] = zero;
        for (uint32_t o = 0; o < d; ++o)                       // B <- B (x - z_o), degree o -> o + 1
            for (uint32_t k = o + 1; k-- > 0;) {
                B[k + 1] = h4_add(B[k + 1], B[k]);
                B[k] = h4_sub(zero, h4_mul(zo[o], B[k]));
            }
        // per offset: the cells' coefficients by rational slot, and K_o
        std::vector<std::vector<H4>> coef_at(d, std::vector<H4>(npoly - 1, zero));
        for (uint32_t j = 0; j < nmask; ++j) {
            const uint32_t o = off_index[off_of(j)];
            const H4 cj = h4_from_fp(fp_from_limbs64(coeff_trace + 4 * j));
            coef_at[o][npoly - 2] = h4_add(coef_at[o][npoly - 2], h4_mul(cj, h4_from_fp(fp_from_limbs64(ood_trace + 4 * j))));
            if (rat_slot[mask_col[j]] >= 0) coef_at[o][rat_slot[mask_col[j]]] = h4_add(coef_at[o][rat_slot[mask_col[j]]], cj);
        }
        for (uint32_t o = 0; o < d; ++o) {
            Q[d - 1] = B[d];
            for (uint32_t k = d - 1; k > 0; --k) Q[k - 1] = h4_add(B[k], h4_mul(zo[o], Q[k]));
            for (uint32_t pidx = 0; pidx + 1 < npoly; ++pidx) {
                const H4 cf = coef_at[o][pidx];
                if (h4_is_zero(cf)) continue;
                for (uint32_t k = 0; k < d; ++k) A[pidx][k] = h4_add(A[pidx][k], h4_mul(cf, Q[k]));
            }
        }
        // multipliers leave the transform / the inversion in R280 form: A_c times 2^24, B times 2^-24 (A_K as it is)
        const H4 up = h4_from_fp(r280_factor), down = h4_from_fp(fp_inv_safegcd(r280_factor));
        auto brev = [&](uint32_t k) { uint32_t r = 0; for (uint32_t b = 0; b < poly_log; ++b) r |= ((k >> b) & 1u) << (poly_log - 1 - b); return r; };
        for (uint32_t pidx = 0; pidx + 2 < npoly; ++pidx)
            for (uint32_t k = 0; k < d; ++k) poly_coef[plen * pidx + brev(k)] = h4_to_fp(h4_mul(A[pidx][k], up));
        for (uint32_t k = 0; k < d; ++k) poly_coef[plen * (npoly - 2) + brev(k)] = h4_to_fp(A[npoly - 2][k]);
        for (uint32_t k = 0; k <= d; ++k) poly_coef[plen * (npoly - 1) + brev(k)] = h4_to_fp(h4_mul(B[k], down));
    };
    const uint32_t ntaps = (uint32_t)tap_shift.size(), ncoldesc = (uint32_t)(cdesc.size() / 3);
    std::vector<Fp> comp_coef(ncomp ? ncomp : 1);
    Fp comp_k = fp_zero(), zc = fp_one();
    for (uint32_t k = 0; k < ncomp; ++k) {
        comp_coef[k] = fp_from_limbs64(coeff_comp + 4 * k);
        comp_k = fp_add(comp_k, fp_mul(comp_coef[k], fp_from_limbs64(ood_comp + 4 * k)));
        zc = fp_mul(zc, zf);
    }
    for (uint32_t k = 0; k < ncomp; ++k) comp_coef[k] = fp_mul(comp_coef[k], r280_factor);
    // inverse tables: the whole sub-coset, or this range with the rows the shifts reach in front of it
    const uint64_t chunk = 1ull << BATCH_INVERSE_RANGE_LOG_CHUNK;
    const uint64_t pre = block ? ((uint64_t)max_shift + chunk - 1) / chunk * chunk : 0;
    const uint64_t d_len = block ? (pre + count + chunk - 1) / chunk * chunk : n, dc_len = block ? (count + chunk - 1) / chunk * chunk : n;
    ss_status st = SS_OK;
    Fp *D, *Dc, *poly_vals = nullptr;                          // poly_vals: npoly columns of n values + n of scratch for the inversion
    if (block) {
        st = ctx->ensure_scratch2((d_len + dc_len + (npoly ? (size_t)npoly * n + count : 0)) * sizeof(Fp));
        if (st != SS_OK) return st;
        D = (Fp *)ctx->scratch2; Dc = D + d_len; poly_vals = Dc + dc_len;
    } else {
        // D, Dc (sub-coset tables), the sub-coset values and the rational part's polynomial values live in scratch2
        st = ctx->ensure_scratch2((size_t)(3 + (npoly ? npoly + 1 : 0)) * n * sizeof(Fp));
        if (st != SS_OK) return st;
        D = (Fp *)ctx->scratch2; Dc = D + n; d_sub = D + 2 * n; poly_vals = D + 3 * n;
        if (sub_out) *sub_out = d_sub;
    }
    // the tables ss_deep_prepare left for this very point (queued on this stream before anything this call queues)
    const ss_ctx::DeepKey &dk = ctx->deep_key;
    const bool prepared = !block && dk.valid && dk.log_n == log_n && dk.ncomp == ncomp && fp_eq(dk.off, off) && fp_eq(dk.z, zf);
    ctx->deep_key.valid = false;
    if (prepared) { D = ctx->deep_tab; Dc = D + n; }
    const size_t small = (size_t)(ntaps + 1) * (4 + 32) + (size_t)(ncoldesc + 1) * 12 + (size_t)(ncomp + 1) * 32 + 256 + poly_coef.size() * 32;
    st = ctx->ensure_scratch(small);
    if (st != SS_OK) return st;
    char *p = (char *)ctx->scratch;
    Fp *d_tap_coef = (Fp *)p; p += (size_t)(ntaps + 1) * 32;
    Fp *d_comp_coef = (Fp *)p; p += (size_t)(ncomp + 1) * 32;
    Fp *d_poly_coef = (Fp *)p; p += poly_coef.size() * 32;
    uint32_t *d_tap_shift = (uint32_t *)p; p += (size_t)(ntaps + 1) * 4;
    uint32_t *d_cdesc = (uint32_t *)p;
    hipStream_t s = ctx->stream;

    if (ntaps) {
        HIP_TRY(hipMemcpyAsync(d_tap_coef, tap_coef.data(), (size_t)ntaps * 32, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_tap_shift, tap_shift.data(), (size_t)ntaps * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_cdesc, cdesc.data(), (size_t)ncoldesc * 12, hipMemcpyHostToDevice, s));
    }
    if (ncomp) HIP_TRY(hipMemcpyAsync(d_comp_coef, comp_coef.data(), (size_t)ncomp * 32, hipMemcpyHostToDevice, s));
    const Fp wn = root_of_unity(log_n);
    {
        ss_ctx::Scope prof(ctx, SS_PROF_DEEP);
        if (block) {
            // D[j] <-> sub-coset point m0 - pre + j (indices modulo n: the powers of w_n wrap by themselves)
            const Fp x_d = fp_mul(off, fp_pow_u64(wn, (m0 + n - pre % n) % n)), x_c = fp_mul(off, fp_pow_u64(wn, m0 % n));
            HIP_TRY(launch_batch_inverse_range(s, D, d_len, x_d, wn, root_of_unity_inv(log_n), zf, true));
            if (ncomp) HIP_TRY(launch_batch_inverse_range(s, Dc, dc_len, x_c, wn, root_of_unity_inv(log_n), zc, true));
        } else if (!prepared) {
            HIP_TRY(launch_batch_inverse(s, D, log_n, off, wn, root_of_unity_inv(log_n), zf, true));
            if (ncomp) HIP_TRY(launch_batch_inverse(s, Dc, log_n, off, wn, root_of_unity_inv(log_n), zc, true));
        }
        HIP_TRY(launch_deep(s, (const void *const *)d_trace_lde, ntrace_cols, (const void *const *)d_comp_lde, ncomp, D, Dc,
                            d_tap_shift, d_tap_coef, d_cdesc, ncoldesc, d_comp_coef, comp_k, count,
                            block ? (uint32_t)pre : 0u, block ? 0xffffffffu : (uint32_t)(n - 1), log_blowup, d_sub));
    }
    if (npoly) {
        build_polynomials();
        HIP_TRY(hipMemcpyAsync(d_poly_coef, poly_coef.data(), poly_coef.size() * 32, hipMemcpyHostToDevice, s));
        // the polynomials on the sub-coset: one pruned forward transform for all of them (the transforms are booked as transforms)
        const Fp *tw = nullptr;
        st = ctx->get_plan(log_n, false, off, &tw);
        if (st != SS_OK) return st;
        ColPtrs cols;
        memset(&cols, 0, sizeof cols);
        const Fp *a_vals[DEEP_RATIONAL_MAX_COLS];
        const void *t_cols[DEEP_RATIONAL_MAX_COLS];
        for (uint32_t k = 0; k < npoly; ++k) { cols.src[k] = d_poly_coef + ((size_t)k << poly_log); cols.dst[k] = poly_vals + (size_t)k * n; }
        for (uint32_t k = 0; k + 2 < npoly; ++k) { a_vals[k] = poly_vals + (size_t)k * n; t_cols[k] = d_trace_lde[rat_cols[k]]; }
        st = run_forward(ctx, cols, npoly, log_n, tw, log_n - poly_log);
        if (st != SS_OK) return st;
        ss_ctx::Scope prof(ctx, SS_PROF_DEEP);
        // this call's points m0 .. m0 + count of them (the whole sub-coset, or a rank's range; the trace columns are its blocks)
        Fp *b_vals = poly_vals + (size_t)(npoly - 1) * n + m0;
        for (uint32_t k = 0; k + 2 < npoly; ++k) a_vals[k] += m0;
        HIP_TRY(launch_batch_inverse_values(s, b_vals, poly_vals + (size_t)npoly * n, count));
        HIP_TRY(launch_deep_rational(s, t_cols, a_vals, npoly - 2, poly_vals + (size_t)(npoly - 2) * n + m0, b_vals, count, log_blowup, d_sub));
    }
    HIP_TRY(hipStreamSynchronize(s));     // host staging vectors go out of scope
    return SS_OK;
}

// sub-coset values of a polynomial of degree < n (on offset * <w_n>, natural order) -> its evaluations over the LDE domain
ss_status deep_extend(ss_ctx *ctx, Fp *d_sub, uint32_t log_n, uint32_t log_blowup, const Fp &off, uint64_t *d_out) {
    // The values on off * <w_n> are those of Q(y) = P(off y) on <w_n>, and P(off w_2n^k) = Q(w_2n^k): interpolate and evaluate Q
    // over the subgroups themselves (offset 1 on both sides - the inverse then runs the cheaper CTI network, and the
    // coefficients in between, c_j off^j, are nobody else's).
    (void)off;
    const Fp *tw_inv = nullptr, *tw_fwd = nullptr;
    int inv_mode = NTT_MODE_DIF;
    ss_status st = get_inverse_plan(ctx, log_n, fp_one(), &tw_inv, &inv_mode);
    if (st != SS_OK) return st;
    st = ctx->get_plan(log_n + log_blowup, false, fp_one(), &tw_fwd);
    if (st != SS_OK) return st;
    ColPtrs inv, fwd;
    memset(&inv, 0, sizeof inv); memset(&fwd, 0, sizeof fwd);
    inv.src[0] = d_sub; inv.dst[0] = d_sub;
    fwd.src[0] = d_sub; fwd.dst[0] = d_out;
    st = run_inverse(ctx, inv, 1, log_n, tw_inv, inv_mode);
    if (st != SS_OK) return st;
    return run_forward(ctx, fwd, 1, log_n + log_blowup, tw_fwd, log_blowup);
}

ss_status deep_check_args(ss_ctx *ctx, const uint64_t *const *d_trace_lde, uint32_t ntrace_cols, const uint64_t *const *d_comp_lde,
                          uint32_t ncomp, uint32_t log_n, uint32_t log_blowup, const uint32_t *mask_col, const uint32_t *mask_off,
                          uint32_t nmask, const uint64_t *ood_trace, const uint64_t *coeff_trace, const uint64_t *ood_comp,
                          const uint64_t *coeff_comp, const uint64_t z[4], const void *d_out) {
    if (!ctx || !d_trace_lde || !z || !d_out) return fail(SS_ERR_INVALID, "NULL argument");
    if (nmask && (!mask_col || !mask_off || !ood_trace || !coeff_trace)) return fail(SS_ERR_INVALID, "NULL mask argument");
    if (ncomp && (!d_comp_lde || !ood_comp || !coeff_comp)) return fail(SS_ERR_INVALID, "NULL composition argument");
    if (!valid_log(log_n) || !valid_log(log_n + log_blowup)) return fail(SS_ERR_INVALID, "size out of range");
    if (ntrace_cols > (uint32_t)MAX_COLS || ncomp > 4) return fail(SS_ERR_UNSUPPORTED, "too many columns");
    return SS_OK;
}
}  // namespace

extern "C" {

ss_status ss_deep_prepare(ss_ctx *ctx, uint32_t ncomp, uint32_t log_n, const uint64_t offset[4], const uint64_t z[4]) {
    if (!ctx || !z) return fail(SS_ERR_INVALID, "NULL argument");
    if (!valid_log(log_n) || ncomp > 4) return fail(SS_ERR_INVALID, "size out of range");
    const uint64_t n = 1ull << log_n;
    ctx->deep_key.valid = false;
    if (ctx->deep_tab_elems < 2 * n) {
        if (ctx->deep_tab) { HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(hipFree(ctx->deep_tab)); ctx->deep_tab = nullptr; ctx->deep_tab_elems = 0; }
        HIP_TRY(ctx->malloc_retry((void **)&ctx->deep_tab, 2 * n * sizeof(Fp)));
        ctx->deep_tab_elems = 2 * n;
    }
    const Fp off = offset ? fp_from_limbs64(offset) : fp_one(), zf = fp_from_limbs64(z), wn = root_of_unity(log_n);
    Fp zc = fp_one();
    for (uint32_t k = 0; k < ncomp; ++k) zc = fp_mul(zc, zf);
    {
        ss_ctx::Scope prof(ctx, SS_PROF_DEEP);
        HIP_TRY(launch_batch_inverse(ctx->stream, ctx->deep_tab, log_n, off, wn, root_of_unity_inv(log_n), zf, true));
        if (ncomp) HIP_TRY(launch_batch_inverse(ctx->stream, ctx->deep_tab + n, log_n, off, wn, root_of_unity_inv(log_n), zc, true));
    }
    ctx->deep_key.log_n = log_n; ctx->deep_key.ncomp = ncomp; ctx->deep_key.off = off; ctx->deep_key.z = zf;
    ctx->deep_key.valid = true;
    return SS_OK;                                // nothing waited for: the caller's host work runs beside the two kernels
}

ss_status ss_deep_compose(ss_ctx *ctx, const uint64_t *const *d_trace_lde, uint32_t ntrace_cols,
                          const uint64_t *const *d_comp_lde, uint32_t ncomp, uint32_t log_n, uint32_t log_blowup,
                          const uint64_t offset[4], const uint32_t *mask_col, const uint32_t *mask_off,
                          uint32_t nmask, const uint64_t *ood_trace, const uint64_t *coeff_trace,
                          const uint64_t *ood_comp, const uint64_t *coeff_comp, const uint64_t z[4],
                          uint64_t *d_out) {
    ss_status st = deep_check_args(ctx, d_trace_lde, ntrace_cols, d_comp_lde, ncomp, log_n, log_blowup, mask_col, mask_off, nmask,
                                   ood_trace, coeff_trace, ood_comp, coeff_comp, z, d_out);
    if (st != SS_OK) return st;
    const uint64_t n = 1ull << log_n;
    // the DEEP polynomial has degree < n: compose it on the sub-coset offset*<w_n> (LDE rows m * blowup), interpolate,
    // and expand back to the LDE domain
    Fp *sub = nullptr;
    st = deep_subcoset(ctx, d_trace_lde, ntrace_cols, d_comp_lde, ncomp, log_n, log_blowup, offset, mask_col, mask_off, nmask,
                       ood_trace, coeff_trace, ood_comp, coeff_comp, z, 0, n, false, nullptr, &sub);
    if (st != SS_OK) return st;
    st = deep_extend(ctx, sub, log_n, log_blowup, offset ? fp_from_limbs64(offset) : fp_one(), d_out);
    if (st != SS_OK) return st;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SS_OK;
}

ss_status ss_deep_compose_rows(ss_ctx *ctx, const uint64_t *const *d_trace_blocks, uint32_t ntrace_cols,
                               const uint64_t *const *d_comp_blocks, uint32_t ncomp, uint32_t log_n, uint32_t log_blowup,
                               const uint64_t offset[4], const uint32_t *mask_col, const uint32_t *mask_off,
                               uint32_t nmask, const uint64_t *ood_trace, const uint64_t *coeff_trace,
                               const uint64_t *ood_comp, const uint64_t *coeff_comp, const uint64_t z[4],
                               uint64_t m0, uint64_t count, uint64_t *d_out_subcoset) {
    ss_status st = deep_check_args(ctx, d_trace_blocks, ntrace_cols, d_comp_blocks, ncomp, log_n, log_blowup, mask_col, mask_off, nmask,
                                   ood_trace, coeff_trace, ood_comp, coeff_comp, z, d_out_subcoset);
    if (st != SS_OK) return st;
    const uint64_t n = 1ull << log_n;
    if (count == 0 || count > n || m0 >= n) return fail(SS_ERR_INVALID, "sub-coset range out of bounds");
    return deep_subcoset(ctx, d_trace_blocks, ntrace_cols, d_comp_blocks, ncomp, log_n, log_blowup, offset, mask_col, mask_off, nmask,
                         ood_trace, coeff_trace, ood_comp, coeff_comp, z, m0, count, true, (Fp *)d_out_subcoset, nullptr);
}

ss_status ss_deep_extend(ss_ctx *ctx, uint64_t *d_subcoset, uint32_t log_n, uint32_t log_blowup, const uint64_t offset[4],
                         uint64_t *d_out) {
    if (!ctx || !d_subcoset || !d_out) return fail(SS_ERR_INVALID, "NULL argument");
    if (!valid_log(log_n) || !valid_log(log_n + log_blowup)) return fail(SS_ERR_INVALID, "size out of range");
    ss_status st = deep_extend(ctx, (Fp *)d_subcoset, log_n, log_blowup, offset ? fp_from_limbs64(offset) : fp_one(), d_out);
    if (st != SS_OK) return st;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SS_OK;
}

// ------------------------------------------------------------------- Q1
ss_status ss_inverse_table(ss_ctx *ctx, uint32_t log_N, const uint64_t offset[4], const uint64_t cval[4],
                           uint64_t *d_out) {
    if (!ctx || !cval || !d_out) return fail(SS_ERR_INVALID, "NULL argument");
    if (!valid_log(log_N)) return fail(SS_ERR_INVALID, "log_N out of range");
    const Fp off = offset ? fp_from_limbs64(offset) : fp_one();
    const Fp w = root_of_unity(log_N);
    ss_ctx::Scope prof(ctx, SS_PROF_QUOTIENT);
    HIP_TRY(launch_batch_inverse(ctx->stream, (Fp *)d_out, log_N, off, w, root_of_unity_inv(log_N), fp_from_limbs64(cval), false));
    return SS_OK;
}

}  // extern "C"

namespace {
// the compiled kernels, by program
const QGenKernel *quotient_gen_find(const uint32_t *code, uint32_t n_instr, bool *variant_missing) {
    *variant_missing = false;
    uint64_t h = 0xcbf29ce484222325ull;                                  // FNV-1a over the code words (tools/gen_quotient.py)
    for (size_t k = 0; k < 2 * (size_t)n_instr; ++k)
        for (int b = 0; b < 4; ++b) h = (h ^ ((code[k] >> (8 * b)) & 0xffu)) * 0x100000001b3ull;
    // quotient_gen_variants.inc: the default kernels; with make QG_AB=1 also the A/B variants of tools/gen_quotient.py --all-variants
#define QG_VARIANT(entry) &entry(),
    const QGenKernel *all[] = {
#include "quotient_gen_variants.inc"
    };
#undef QG_VARIANT
    // A/B runs: SS_QG_VARIANT=k (tools/gen_quotient.py VARIANTS), or per layout SS_QG_VARIANT_STARKNET / SS_QG_VARIANT_RECURSIVE
    for (const QGenKernel *k : all) {
        if (k->code_hash != h || k->n_instr != n_instr) continue;
        uint32_t variant = 0;
        if (const char *e = getenv("SS_QG_VARIANT")) variant = (uint32_t)strtoul(e, nullptr, 10);
        if (const char *e = getenv(strcmp(k->layout, "starknet") == 0 ? "SS_QG_VARIANT_STARKNET" : "SS_QG_VARIANT_RECURSIVE")) variant = (uint32_t)strtoul(e, nullptr, 10);
        if (k->variant == variant) return k;
        *variant_missing = true;              // the program has compiled kernels, the variant asked for is not in this build
    }
    return nullptr;
}

// row0 / npoints / block: the whole domain (0, N, false) or a row block whose columns carry the rows behind it
ss_status eval_quotient_compiled(ss_ctx *ctx, const QGenKernel &gen, const ss_air_program *prog, const uint64_t *const *d_lde_cols,
                                 uint32_t ncols, uint32_t log_N, uint32_t log_blowup, const uint64_t offset[4], uint64_t *d_out,
                                 uint64_t row0, uint64_t npoints, bool block) {
    const uint64_t N = npoints;
    // constants in limb form: 9 x 28-bit limbs of the interchange image (add / sub / mov) and of the R280 form (fl_mul_r280)
    const uint32_t nc = prog->n_consts ? prog->n_consts : 1u;
    // [constants | table descriptors, then those of the scaled copies | one step w^(lanes of the grid) per part, 32-byte aligned]
    const size_t n_desc = (size_t)(prog->n_tables ? prog->n_tables : 1u) + gen.n_scaled;
    const size_t wstep_at = (((size_t)nc * QG_CONST_STRIDE + 2 * n_desc) + 7) / 8 * 8;
    std::vector<uint32_t> host(wstep_at + 8 * (size_t)QG_MAX_PARTS, 0u);
    for (uint32_t k = 0; k < prog->n_consts; ++k) {
        const Fp c = fp_from_limbs64(prog->consts + 4 * (size_t)k);
        Fp two24 = fp_zero(); two24.v[0] = 1u << 24;
        const Fp up = fp_mul(c, fp_to_mont(two24));              // c 2^24: takes a constraint's own wide sum (g C 2^-24) back, quotient_gen.h
        const Fl a = fl_from_fp(c), r = fl_to_r280(c), ru = fl_to_r280(up), rn = fl_to_r280(fp_neg(up));
        for (int j = 0; j < 9; ++j) {
            uint32_t *e = &host[(size_t)k * QG_CONST_STRIDE];
            e[j] = a.l[j]; e[12 + j] = r.l[j]; e[24 + j] = ru.l[j]; e[36 + j] = rn.l[j];
        }
    }
    uint32_t *tdesc = host.data() + (size_t)nc * QG_CONST_STRIDE;
    for (uint32_t t = 0; t < prog->n_tables; ++t) {
        tdesc[2 * t] = prog->table_desc[2 * t];
        tdesc[2 * t + 1] = (uint32_t)((1ull << prog->table_desc[2 * t + 1]) - 1ull);
    }
    // the multiplier-only tables times 2^24, one after the other behind the staged words (quotient_gen.h QG_TABLE_SCALED_RAW)
    QgScaleArgs sc;
    uint64_t scaled_felts = 0;
    if (gen.n_scaled > (uint32_t)QG_MAX_SCALED) return fail(SS_ERR_UNSUPPORTED, "%u scaled tables > %d", gen.n_scaled, QG_MAX_SCALED);
    for (uint32_t j = 0; j < gen.n_scaled; ++j) {
        const uint32_t t = gen.scaled[j];
        if (t >= prog->n_tables) return fail(SS_ERR_INVALID, "scaled table %u of %u", t, prog->n_tables);
        const uint64_t len = 1ull << prog->table_desc[2 * t + 1];
        sc.src[j] = prog->table_desc[2 * t]; sc.dst[j] = (uint32_t)scaled_felts; sc.len[j] = (uint32_t)len;
        tdesc[2 * (prog->n_tables + j)] = (uint32_t)scaled_felts;
        tdesc[2 * (prog->n_tables + j) + 1] = (uint32_t)(len - 1ull);
        scaled_felts += len;
    }
    sc.n = gen.n_scaled;
    const Fp w_dom = root_of_unity(log_N);
    std::vector<uint64_t> part_blocks(gen.n_parts);
    for (uint32_t p = 0; p < gen.n_parts; ++p) {
        // one workgroup per CU and SIMD slot the part's register / LDS budget allows; SS_QG_BLOCKS overrides (experiments)
        uint64_t blocks = 256ull * gen.parts[p].wgs_per_cu;
        if (const char *e = getenv("SS_QG_BLOCKS")) blocks = strtoull(e, nullptr, 10);
        if (blocks * gen.parts[p].threads > N) blocks = N / gen.parts[p].threads;
        if (blocks == 0) blocks = 1;
        part_blocks[p] = blocks;
        const Fp ws = fp_pow_u64(w_dom, blocks * gen.parts[p].threads);
        for (int j = 0; j < 8; ++j) host[wstep_at + 8 * (size_t)p + j] = ws.v[j];
    }
    const size_t scaled_at = ((host.size() * 4 + 256 + 63) / 64) * 64;
    ss_status st = ctx->ensure_scratch(scaled_at + scaled_felts * sizeof(Fp));
    if (st != SS_OK) return st;
    hipStream_t s = ctx->stream;
    void *pinned = nullptr;                      // no host round trip: the copy leaves from pinned memory the context owns
    if ((st = ctx->stage_acquire(host.size() * 4, &pinned)) != SS_OK) return st;
    memcpy(pinned, host.data(), host.size() * 4);
    HIP_TRY(hipMemcpyAsync(ctx->scratch, pinned, host.size() * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(ctx->stage_event, s));
    QGenArgs a;
    for (int c = 0; c < QG_MAX_COLS; ++c) a.cols[c] = c < (int)ncols ? (const Fp *)d_lde_cols[c] : nullptr;
    a.tables = (const Fp *)prog->d_tables;
    a.tables_scaled = (const Fp *)((char *)ctx->scratch + scaled_at);
    if (sc.n) {
        sc.tables = a.tables; sc.out = (Fp *)((char *)ctx->scratch + scaled_at);
        HIP_TRY(launch_qg_scale_tables(s, sc, scaled_felts));
    }
    a.consts = (const uint32_t *)ctx->scratch;
    a.tdesc = a.consts + (size_t)nc * QG_CONST_STRIDE;
    a.out = (Fp *)d_out;
    a.sink = (Fp *)((char *)ctx->scratch + ((host.size() * 4 + 63) / 64) * 64);      // inside ensure_scratch's 256 spare bytes
    a.npoints = npoints; a.row0 = (uint32_t)row0; a.log_blowup = log_blowup;
    a.trace_mask = block ? 0xffffffffu : (uint32_t)((1ull << log_N) - 1ull);
    a.w = w_dom;
    a.offset = fp_mul(offset ? fp_from_limbs64(offset) : fp_one(), fp_pow_u64(a.w, row0));
    // the program's parts, one launch each on the context's stream: part 0 stores its sum, the others add theirs (a lane owns the
    // same points in every part only if the grids agree - they need not: a part reads out[i] written by the PREVIOUS launch)
    for (uint32_t p = 0; p < gen.n_parts; ++p) {
        const QGenPart &part = gen.parts[p];
        a.wstep_ptr = reinterpret_cast<const Fp *>(a.consts + wstep_at + 8 * (size_t)p);
        ss_ctx::Scope prof(ctx, SS_PROF_QUOTIENT);
        HIP_TRY(part.launch(s, a, (uint32_t)part_blocks[p]));
    }
    return SS_OK;
}
}  // namespace

extern "C" {

// whole domain: row0 = 0, npoints = N, block_rows = 0.  Row block: the columns hold LDE rows row0 .. row0 + block_rows.
static ss_status eval_quotient_interpreted(ss_ctx *ctx, const ss_air_program *prog, const uint64_t *const *d_lde_cols, uint32_t ncols, uint32_t log_N,
                                           uint32_t log_blowup, const uint64_t offset[4], uint64_t row0, uint64_t N, bool block, uint64_t *d_out);

static ss_status eval_quotient_impl(ss_ctx *ctx, const ss_air_program *prog, const uint64_t *const *d_lde_cols,
                                    uint32_t ncols, uint32_t log_n, uint32_t log_blowup, const uint64_t offset[4],
                                    uint64_t row0, uint64_t npoints, uint64_t block_rows, uint64_t *d_out) {
    if (!ctx || !prog || !d_lde_cols || !d_out) return fail(SS_ERR_INVALID, "NULL argument");
    if (!prog->code || prog->n_instr == 0) return fail(SS_ERR_INVALID, "empty program");
    const uint32_t log_N = log_n + log_blowup;
    if (!valid_log(log_n) || !valid_log(log_N)) return fail(SS_ERR_INVALID, "size out of range");
    if (ncols > (uint32_t)MAX_COLS) return fail(SS_ERR_UNSUPPORTED, "ncols %u > %d", ncols, MAX_COLS);
    if (has_null((const void *const *)d_lde_cols, ncols)) return fail(SS_ERR_INVALID, "NULL column");
    const bool block = block_rows != 0;
    const uint64_t N = block ? npoints : 1ull << log_N;
    if (block && (npoints == 0 || row0 + npoints > (1ull << log_N))) return fail(SS_ERR_INVALID, "row block out of the domain");
    // validate the program against its own declarations (it is caller-supplied data)
    for (uint32_t pc = 0; pc < prog->n_instr; ++pc) {
        const uint32_t w0 = prog->code[2 * pc], w1 = prog->code[2 * pc + 1];
        const uint32_t op = w0 & 0xff, d = (w0 >> 8) & 0xf, kind = (w0 >> 12) & 0xf;
        if (op > SS_OP_OUT || d > 3) return fail(SS_ERR_INVALID, "instruction %u: bad opcode/accumulator", pc);
        if (op == SS_OP_ST && w1 >= prog->n_slots) return fail(SS_ERR_INVALID, "instruction %u: slot %u out of range", pc, w1);
        if (op <= SS_OP_MUL) {
            if (kind > SS_SRC_X) return fail(SS_ERR_INVALID, "instruction %u: bad operand kind", pc);
            if (kind == SS_SRC_ACC && w1 > 3) return fail(SS_ERR_INVALID, "instruction %u: bad accumulator", pc);
            if (kind == SS_SRC_SLOT && w1 >= prog->n_slots) return fail(SS_ERR_INVALID, "instruction %u: slot %u out of range", pc, w1);
            if (kind == SS_SRC_CONST && w1 >= prog->n_consts) return fail(SS_ERR_INVALID, "instruction %u: constant %u out of range", pc, w1);
            if (kind == SS_SRC_TRACE && (w1 >> 24) >= ncols) return fail(SS_ERR_INVALID, "instruction %u: column %u out of range", pc, w1 >> 24);
            if (kind == SS_SRC_TRACE && block && npoints + ((uint64_t)(w1 & 0xffffffu) << log_blowup) > block_rows)
                return fail(SS_ERR_INVALID, "instruction %u: row offset %u reaches beyond the block's %llu rows", pc, w1 & 0xffffffu,
                            (unsigned long long)block_rows);
            if (kind == SS_SRC_TABLE && (w1 >= prog->n_tables || !prog->d_tables || !prog->table_desc))
                return fail(SS_ERR_INVALID, "instruction %u: table %u out of range", pc, w1);
        }
    }
    // A layout's composition constraint has a compiled kernel (quotient_gen_<layout>.hip, generated from exactly this
    // program): recognised by the hash of its code words.  Everything per proof (constants, tables, columns, size) is data.
    if (!getenv("SS_QUOTIENT_INTERPRET")) {
        bool variant_missing = false;
        const QGenKernel *gen = quotient_gen_find(prog->code, prog->n_instr, &variant_missing);
        if (!gen && variant_missing) return fail(SS_ERR_UNSUPPORTED, "SS_QG_VARIANT names a kernel variant this build does not hold (make QG_AB=1)");
        if (gen && gen->n_consts == prog->n_consts && gen->n_tables == prog->n_tables && gen->ncols <= ncols)
            return eval_quotient_compiled(ctx, *gen, prog, d_lde_cols, ncols, log_N, log_blowup, offset, d_out, row0, N, block);
    }
    return eval_quotient_interpreted(ctx, prog, d_lde_cols, ncols, log_N, log_blowup, offset, row0, N, block, d_out);
}

static ss_status eval_quotient_interpreted(ss_ctx *ctx, const ss_air_program *prog, const uint64_t *const *d_lde_cols, uint32_t ncols, uint32_t log_N,
                                           uint32_t log_blowup, const uint64_t offset[4], uint64_t row0, uint64_t N, bool block, uint64_t *d_out) {
    uint64_t lanes = 256ull * 256 * 4;                 // 4 workgroups of 256 per CU
    if (lanes > N) lanes = N < 256 ? 256 : (N + 255) / 256 * 256;
    const size_t code_b = ((size_t)prog->n_instr + 1) * 32, const_b = (size_t)(prog->n_consts ? prog->n_consts : 1) * 32;
    const size_t slots_b = (size_t)(prog->n_slots ? prog->n_slots : 1) * lanes * 32;
    ss_status st = ctx->ensure_scratch(slots_b + 2 * const_b + code_b + 256);
    if (st != SS_OK) return st;
    char *p = (char *)ctx->scratch;
    Fp *d_slots = (Fp *)p; p += slots_b;
    Fp *d_consts = (Fp *)p; p += const_b;
    Fp *d_consts_r280 = (Fp *)p; p += const_b;
    uint32_t *d_code = (uint32_t *)p;
    hipStream_t s = ctx->stream;
    // resolve operands and lazy-form bounds on the host (quotient.hip: "device program")
    VmResolve rs;
    for (int c = 0; c < MAX_COLS; ++c) rs.cols[c] = c < (int)ncols ? (const void *)d_lde_cols[c] : nullptr;
    rs.consts = d_consts; rs.consts_r280 = d_consts_r280; rs.tables = prog->d_tables; rs.slots = d_slots; rs.table_desc = prog->table_desc;
    rs.lanes = lanes; rs.log_blowup = log_blowup;
    rs.trace_mask = block ? 0xffffffffu : (uint32_t)((1ull << log_N) - 1ull);
    rs.row0 = (uint32_t)row0;
    std::vector<uint32_t> dev_code(((size_t)prog->n_instr + 1) * 8);
    quotient_build_device_code(prog->code, prog->n_instr, rs, dev_code.data());
    HIP_TRY(hipMemcpyAsync(d_code, dev_code.data(), code_b, hipMemcpyHostToDevice, s));
    std::vector<Fp> consts_r280(prog->n_consts);
    if (prog->n_consts) {
        HIP_TRY(hipMemcpyAsync(d_consts, prog->consts, (size_t)prog->n_consts * 32, hipMemcpyHostToDevice, s));
        Fp two24 = fp_zero(); two24.v[0] = 1u << 24;
        const Fp f = fp_to_mont(two24);
        for (uint32_t k = 0; k < prog->n_consts; ++k) consts_r280[k] = fp_mul(fp_from_limbs64(prog->consts + 4 * (size_t)k), f);
        HIP_TRY(hipMemcpyAsync(d_consts_r280, consts_r280.data(), (size_t)prog->n_consts * 32, hipMemcpyHostToDevice, s));
    }
    const Fp w = root_of_unity(log_N);
    const Fp off = fp_mul(offset ? fp_from_limbs64(offset) : fp_one(), fp_pow_u64(w, row0));
    // optional XCD-contiguous sweep (measured: no gain, the per-XCD window still exceeds L2)
    const uint32_t xcd_split = (lanes % (8 * 256) == 0 && (N >> 3) >= (lanes >> 3) && (N >> 3) % (lanes >> 3) == 0 && (N & (N - 1)) == 0 &&
                                getenv("SS_QUOTIENT_XCD_SPLIT") != nullptr) ? 1u : 0u;
    const Fp wstep = fp_pow_u64(w, xcd_split ? (lanes >> 3) : lanes);
    ss_ctx::Scope prof(ctx, SS_PROF_QUOTIENT);
    HIP_TRY(launch_quotient_vm(s, d_code, prog->n_instr + 1, d_slots, lanes, off, w, wstep, N, xcd_split, (Fp *)d_out));
    HIP_TRY(hipStreamSynchronize(s));      // the caller's host arrays may go away after return
    return SS_OK;
}

ss_status ss_eval_quotient(ss_ctx *ctx, const ss_air_program *prog, const uint64_t *const *d_lde_cols,
                           uint32_t ncols, uint32_t log_n, uint32_t log_blowup, const uint64_t offset[4],
                           uint64_t *d_out) {
    return eval_quotient_impl(ctx, prog, d_lde_cols, ncols, log_n, log_blowup, offset, 0, 0, 0, d_out);
}

ss_status ss_eval_quotient_rows(ss_ctx *ctx, const ss_air_program *prog, const uint64_t *const *d_col_blocks,
                                uint32_t ncols, uint32_t log_n, uint32_t log_blowup, const uint64_t offset[4],
                                uint64_t row0, uint64_t nrows, uint64_t block_rows, uint64_t *d_out) {
    if (block_rows == 0 || nrows == 0 || nrows > block_rows) return fail(SS_ERR_INVALID, "empty row block");
    return eval_quotient_impl(ctx, prog, d_col_blocks, ncols, log_n, log_blowup, offset, row0, nrows, block_rows, d_out);
}


// ------------------------------------------------------------------- the 64-bit field variant (X4)
}  // extern "C"

namespace {
uint64_t gl_mulh(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) % GL_P); }

ss_status gl_get_plan(ss_ctx *ctx, uint32_t log_n, bool inverse, uint64_t offset, const uint64_t **out) {
    auto key = std::make_tuple(log_n, inverse ? 1 : 0, offset);
    auto it = ctx->gl_plans.find(key);
    if (it != ctx->gl_plans.end()) { *out = it->second; return SS_OK; }
    const uint64_t n = 1ull << log_n;
    uint64_t r = gl_root_of_unity_host(log_n), h = offset;
    if (inverse) { r = gl_inv_host(r); h = gl_inv_host(h); }
    const uint64_t half = n / 2 ? n / 2 : 1;
    const uint64_t n_lo = half < 4096 ? half : 4096, n_hi = half < 4096 ? 1 : half / 4096;
    std::vector<uint64_t> host(n_lo + n_hi + log_n + 1);
    uint64_t *lo = host.data(), *hi = lo + n_lo, *hp = hi + n_hi;
    lo[0] = 1;
    for (uint64_t i = 1; i < n_lo; ++i) lo[i] = gl_mulh(lo[i - 1], r);
    const uint64_t r4096 = n_lo == 4096 ? gl_mulh(lo[4095], r) : 1;
    hi[0] = 1;
    for (uint64_t i = 1; i < n_hi; ++i) hi[i] = gl_mulh(hi[i - 1], r4096);
    hp[log_n - 1] = h;                                       // T_s carries h^(n / 2^(s+1))
    for (uint32_t s = log_n - 1; s-- > 0;) hp[s] = gl_mulh(hp[s + 1], hp[s + 1]);
    uint64_t *d_tabs = nullptr, *d_tw = nullptr;
    HIP_TRY(hipMalloc(&d_tabs, host.size() * 8));
    HIP_TRY(hipMalloc(&d_tw, n * 8));
    HIP_TRY(hipMemcpyAsync(d_tabs, host.data(), host.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(launch_gl_twiddles(ctx->stream, d_tw, d_tabs, d_tabs + n_lo, d_tabs + n_lo + n_hi, log_n));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(d_tabs));
    ctx->gl_plans[key] = d_tw;
    *out = d_tw;
    return SS_OK;
}

std::vector<Pass> gl_plan_passes(uint32_t log_n) {          // csrc/gl_ntt.h: the plan the host test of the pass code runs too
    ss::GlPass passes[8];
    const int np = ss::gl_plan_passes_into(log_n, gl_log_tile_max(), passes, gl_log_min_run());
    std::vector<Pass> v;
    for (int i = 0; i < np; ++i) v.push_back({passes[i].s0, passes[i].r});
    return v;
}
// forward: bit-reversed (optionally sub-sampled) src -> natural dst
ss_status gl_run_forward(ss_ctx *ctx, const void *const *src, void *const *dst, uint32_t ncols, uint32_t log_n, const uint64_t *tw, uint32_t log_expand) {
    const uint32_t lt = gl_log_tile_max(), log_tile = log_n < lt ? log_n : lt;
    const std::vector<Pass> passes = gl_plan_passes(log_n);
    if (log_expand > passes[0].r) return fail(SS_ERR_INVALID, "log_blowup %u too large", log_expand);
    for (size_t i = 0; i < passes.size(); ++i) {
        ss_ctx::Scope prof(ctx, SS_PROF_NTT_PASS);
        HIP_TRY(launch_gl_ntt_pass(ctx->stream, false, i == 0 ? src : (const void *const *)dst, dst, ncols, tw, log_n, passes[i].s0, passes[i].r, log_tile,
                                   i == 0 ? log_expand : 0, i == 0 ? log_expand : 0, 1));
    }
    return SS_OK;
}
// inverse: natural src -> bit-reversed dst, scaled by 1/n
ss_status gl_run_inverse(ss_ctx *ctx, const void *const *src, void *const *dst, uint32_t ncols, uint32_t log_n, const uint64_t *tw) {
    const uint32_t lt = gl_log_tile_max(), log_tile = log_n < lt ? log_n : lt;
    const std::vector<Pass> passes = gl_plan_passes(log_n);
    const uint64_t ninv = gl_inv_host((1ull << log_n) % GL_P);
    for (size_t i = passes.size(); i-- > 0;) {
        const bool first = i == passes.size() - 1;
        ss_ctx::Scope prof(ctx, SS_PROF_NTT_PASS);
        HIP_TRY(launch_gl_ntt_pass(ctx->stream, true, first ? src : (const void *const *)dst, dst, ncols, tw, log_n, passes[i].s0, passes[i].r, log_tile, 0, 0,
                                   i == 0 ? ninv : 1));
    }
    return SS_OK;
}
bool gl_valid_log(uint32_t log_n) { return log_n >= 1 && log_n <= 30; }
}  // namespace

extern "C" {

ss_status ss_ntt_gl64(ss_ctx *ctx, uint64_t *const *d_cols, uint32_t ncols, uint32_t log_n, int direction, uint64_t offset,
                      int in_order, int out_order) {
    if (!ctx || !d_cols) return fail(SS_ERR_INVALID, "NULL argument");
    if (!gl_valid_log(log_n)) return fail(SS_ERR_INVALID, "log_n out of range");
    if (ncols == 0 || ncols > (uint32_t)MAX_COLS) return fail(SS_ERR_UNSUPPORTED, "ncols %u out of range", ncols);
    if (offset == 0 || offset >= GL_P) return fail(SS_ERR_INVALID, "the coset offset is not a non-zero field element");
    if (direction != SS_NTT_FORWARD && direction != SS_NTT_INVERSE) return fail(SS_ERR_INVALID, "bad direction");
    if (!valid_order(in_order) || !valid_order(out_order)) return fail(SS_ERR_INVALID, "bad element order %d / %d", in_order, out_order);
    if (has_null((const void *const *)d_cols, ncols)) return fail(SS_ERR_INVALID, "NULL column");
    const bool inverse = direction == SS_NTT_INVERSE;
    const uint64_t *tw = nullptr;
    ss_status st = gl_get_plan(ctx, log_n, inverse, offset, &tw);
    if (st != SS_OK) return st;
    const uint64_t n = 1ull << log_n;
    // the networks want: forward bit-reversed in / natural out, inverse natural in / bit-reversed out; other orders go through a
    // bit-reversal copy in scratch
    const bool pre = inverse ? in_order == SS_ORDER_BITREV : in_order == SS_ORDER_NATURAL;
    const bool post = inverse ? out_order == SS_ORDER_NATURAL : out_order == SS_ORDER_BITREV;
    if (pre || post) { st = ctx->ensure_scratch(n * 8); if (st != SS_OK) return st; }
    for (uint32_t c = 0; c < ncols; ++c) {
        uint64_t *col = d_cols[c];
        const void *src = col; void *dst = col;
        if (pre) { HIP_TRY(launch_gl_bitrev_copy(ctx->stream, col, (uint64_t *)ctx->scratch, log_n)); src = ctx->scratch; }
        st = inverse ? gl_run_inverse(ctx, &src, &dst, 1, log_n, tw) : gl_run_forward(ctx, &src, &dst, 1, log_n, tw, 0);
        if (st != SS_OK) return st;
        if (post) {
            HIP_TRY(launch_gl_bitrev_copy(ctx->stream, col, (uint64_t *)ctx->scratch, log_n));
            HIP_TRY(hipMemcpyAsync(col, ctx->scratch, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
        }
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SS_OK;
}

ss_status ss_lde_gl64(ss_ctx *ctx, const uint64_t *const *d_in, uint32_t ncols, uint32_t log_n, uint32_t log_blowup, uint64_t offset,
                      uint64_t *const *d_evals, uint64_t *const *d_coeffs) {
    if (!ctx || !d_in || !d_evals) return fail(SS_ERR_INVALID, "NULL argument");
    if (!gl_valid_log(log_n) || !gl_valid_log(log_n + log_blowup)) return fail(SS_ERR_INVALID, "size out of range");
    if (offset == 0 || offset >= GL_P) return fail(SS_ERR_INVALID, "the coset offset is not a non-zero field element");
    if (ncols == 0) return SS_OK;
    if (has_null((const void *const *)d_in, ncols) || has_null((const void *const *)d_evals, ncols) || (d_coeffs && has_null((const void *const *)d_coeffs, ncols)))
        return fail(SS_ERR_INVALID, "NULL column");
    const uint64_t n = 1ull << log_n;
    const uint64_t *tw_inv = nullptr, *tw_fwd = nullptr;
    ss_status st = gl_get_plan(ctx, log_n, true, 1, &tw_inv);
    if (st != SS_OK) return st;
    st = gl_get_plan(ctx, log_n + log_blowup, false, offset, &tw_fwd);
    if (st != SS_OK) return st;
    for (uint32_t base = 0; base < ncols; base += MAX_COLS) {
        const uint32_t nc = ncols - base < (uint32_t)MAX_COLS ? ncols - base : (uint32_t)MAX_COLS;
        if (!d_coeffs) { st = ctx->ensure_scratch(n * 8 * nc); if (st != SS_OK) return st; }
        const void *src[MAX_COLS]; void *co[MAX_COLS]; void *ev[MAX_COLS];
        for (uint32_t c = 0; c < nc; ++c) {
            src[c] = d_in[base + c];
            co[c] = d_coeffs ? (void *)d_coeffs[base + c] : (void *)((char *)ctx->scratch + n * 8 * c);
            ev[c] = d_evals[base + c];
        }
        st = gl_run_inverse(ctx, src, co, nc, log_n, tw_inv);                     // natural evaluations -> bit-reversed coefficients
        if (st != SS_OK) return st;
        st = gl_run_forward(ctx, (const void *const *)co, ev, nc, log_n + log_blowup, tw_fwd, log_blowup);
        if (st != SS_OK) return st;
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SS_OK;
}

// Fq3 arithmetic on the host (coefficients, constants): Fp[X] / (X^3 - 2)
struct HGl3 { uint64_t c[3]; };
static uint64_t gl_addh(uint64_t a, uint64_t b) { const uint64_t s = a + b; return (s < a || s >= GL_P) ? s - GL_P : s; }
static uint64_t gl_subh(uint64_t a, uint64_t b) { return a >= b ? a - b : a + (GL_P - b); }
static HGl3 gl3_mulh(const HGl3 &a, const HGl3 &b) {
    const uint64_t d0 = gl_mulh(a.c[0], b.c[0]), d1 = gl_addh(gl_mulh(a.c[0], b.c[1]), gl_mulh(a.c[1], b.c[0]));
    const uint64_t d2 = gl_addh(gl_addh(gl_mulh(a.c[0], b.c[2]), gl_mulh(a.c[1], b.c[1])), gl_mulh(a.c[2], b.c[0]));
    const uint64_t d3 = gl_addh(gl_mulh(a.c[1], b.c[2]), gl_mulh(a.c[2], b.c[1])), d4 = gl_mulh(a.c[2], b.c[2]);
    return HGl3{{gl_addh(d0, gl_addh(d3, d3)), gl_addh(d1, gl_addh(d4, d4)), d2}};
}
static HGl3 gl3_addh(const HGl3 &a, const HGl3 &b) { return HGl3{{gl_addh(a.c[0], b.c[0]), gl_addh(a.c[1], b.c[1]), gl_addh(a.c[2], b.c[2])}}; }
static HGl3 gl3_scaleh(const HGl3 &a, uint64_t s) { return HGl3{{gl_mulh(a.c[0], s), gl_mulh(a.c[1], s), gl_mulh(a.c[2], s)}}; }
static bool gl3_valid(const uint64_t *v) { return v[0] < GL_P && v[1] < GL_P && v[2] < GL_P; }

ss_status ss_running_product_gl64x3(ss_ctx *ctx, const uint64_t *d_num_addr, const uint64_t *d_num_val, const uint64_t *d_den_addr,
                                    const uint64_t *d_den_val, uint64_t stride, uint64_t count, const uint64_t z[3], const uint64_t alpha[3],
                                    uint64_t *const d_out[3], uint64_t out_stride, uint64_t out_offset, uint64_t last_out[3]) {
    if (!ctx || !d_num_addr || !d_den_addr || !z || !d_out || !d_out[0] || !d_out[1] || !d_out[2]) return fail(SS_ERR_INVALID, "NULL argument");
    if ((d_num_val || d_den_val) && (!d_num_val || !d_den_val || !alpha)) return fail(SS_ERR_INVALID, "value columns and alpha come together");
    if (count == 0 || stride == 0 || out_stride == 0) return fail(SS_ERR_INVALID, "empty product");
    if (!gl3_valid(z) || (alpha && !gl3_valid(alpha))) return fail(SS_ERR_INVALID, "challenge is not an element of the extension");
    const uint64_t zero3[3] = {0, 0, 0};
    ss_status st = ctx->ensure_scratch2((6 * (count / 7 + 64) + 8) * 8);
    if (st != SS_OK) return st;
    uint64_t *scratch = (uint64_t *)ctx->scratch2, *d_last = scratch + 6 * (count / 7 + 64);
    ss_ctx::Scope prof(ctx, SS_PROF_EXT);
    HIP_TRY(launch_gl3_running_product(ctx->stream, d_num_addr, d_num_val, d_den_addr, d_den_val, stride, count, z, alpha ? alpha : zero3, scratch,
                                       d_out[0], d_out[1], d_out[2], out_stride, out_offset, last_out ? d_last : nullptr));
    if (last_out) {
        HIP_TRY(hipMemcpyAsync(last_out, d_last, 24, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return SS_OK;
}

ss_status ss_hash_rows_gl64(ss_ctx *ctx, int hash_kind, const uint64_t *const *d_segments, uint32_t nseg, uint32_t seg_len, uint64_t nrows,
                            uint8_t *d_digests) {
    if (!ctx || !d_segments || !d_digests) return fail(SS_ERR_INVALID, "NULL argument");
    if (hash_kind != SS_HASH_KECCAK && hash_kind != SS_HASH_BLAKE2S && hash_kind != SS_HASH_SHA256)
        return fail(SS_ERR_INVALID, "hash kind %d: Keccak-256, Blake2s-256 or SHA-256", hash_kind);
    if (nseg == 0 || nseg > (uint32_t)MAX_COLS || seg_len == 0 || seg_len > 64) return fail(SS_ERR_UNSUPPORTED, "row shape %u x %u out of range", nseg, seg_len);
    ConstColPtrs segs;
    memset(&segs, 0, sizeof segs);
    for (uint32_t c = 0; c < nseg; ++c) { if (!d_segments[c]) return fail(SS_ERR_INVALID, "NULL segment"); segs.p[c] = d_segments[c]; }
    ss_ctx::Scope prof(ctx, SS_PROF_HASH_ROWS);
    HIP_TRY(launch_hash_rows_u64(ctx->stream, hash_kind == SS_HASH_KECCAK ? 0 : hash_kind == SS_HASH_SHA256 ? 2 : 1, segs, nseg, seg_len, nrows, d_digests));
    return SS_OK;
}

ss_status ss_gather_rows_gl64(ss_ctx *ctx, const uint64_t *const *d_segments, uint32_t nseg, uint32_t seg_len, uint64_t nrows, const uint64_t *idx,
                              uint32_t nidx, uint64_t *out) {
    if (!ctx || !d_segments || (nidx && (!idx || !out))) return fail(SS_ERR_INVALID, "NULL argument");
    for (uint32_t j = 0; j < nidx; ++j)
        if (idx[j] >= nrows) return fail(SS_ERR_INVALID, "row index %llu out of range", (unsigned long long)idx[j]);
    if (nseg == 0 || nseg > (uint32_t)MAX_COLS || seg_len == 0 || seg_len > 64) return fail(SS_ERR_UNSUPPORTED, "row shape %u x %u out of range", nseg, seg_len);
    if (!nidx) return SS_OK;
    ConstColPtrs segs;
    memset(&segs, 0, sizeof segs);
    for (uint32_t c = 0; c < nseg; ++c) segs.p[c] = d_segments[c];
    const size_t total = (size_t)nidx * nseg * seg_len;
    ss_status st = ctx->ensure_scratch((size_t)nidx * 8 + total * 8);
    if (st != SS_OK) return st;
    uint64_t *d_idx = (uint64_t *)ctx->scratch, *d_out = d_idx + nidx;
    HIP_TRY(hipMemcpyAsync(d_idx, idx, (size_t)nidx * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(launch_gather_rows_u64(ctx->stream, segs, nseg, seg_len, d_idx, nidx, d_out));
    HIP_TRY(hipMemcpyAsync(out, d_out, total * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SS_OK;
}

ss_status ss_eval_quotient_gl64x3(ss_ctx *ctx, const ss_air_program *prog, const uint64_t *const *d_lde_cols, uint32_t ncols, uint32_t log_n,
                                  uint32_t log_blowup, uint64_t offset, uint64_t *d_out) {
    if (!ctx || !prog || !prog->code || !d_lde_cols || !d_out) return fail(SS_ERR_INVALID, "NULL argument");
    if (!gl_valid_log(log_n + log_blowup)) return fail(SS_ERR_INVALID, "size out of range");
    if (ncols > (uint32_t)MAX_COLS) return fail(SS_ERR_UNSUPPORTED, "too many columns");
    if (offset == 0 || offset >= GL_P) return fail(SS_ERR_INVALID, "the coset offset is not a non-zero field element");
    if (prog->n_consts && !prog->consts) return fail(SS_ERR_INVALID, "NULL constants");
    if (prog->n_tables && (!prog->table_desc || !prog->d_tables)) return fail(SS_ERR_INVALID, "NULL tables");
    const uint64_t N = 1ull << (log_n + log_blowup);
    // validate every operand before anything is launched: the kernel trusts the program
    for (uint32_t pc = 0; pc < prog->n_instr; ++pc) {
        const uint32_t w0 = prog->code[2 * pc], w1 = prog->code[2 * pc + 1], op = w0 & 0xff, d = (w0 >> 8) & 0xf, kind = (w0 >> 12) & 0xf;
        if (op > SS_OP_OUT || d > 3) return fail(SS_ERR_INVALID, "instruction %u: bad opcode or accumulator", pc);
        if (op == SS_OP_ST && w1 >= prog->n_slots) return fail(SS_ERR_INVALID, "instruction %u: slot %u out of range", pc, w1);
        if (op > SS_OP_MUL) continue;
        const bool ok = kind == SS_SRC_ACC ? w1 < 4 : kind == SS_SRC_SLOT ? w1 < prog->n_slots : kind == SS_SRC_CONST ? w1 < prog->n_consts
                      : kind == SS_SRC_TRACE ? (w1 >> 24) < ncols : kind == SS_SRC_TABLE ? w1 < prog->n_tables : kind == SS_SRC_X;
        if (!ok) return fail(SS_ERR_INVALID, "instruction %u: operand out of range", pc);
    }
    for (uint32_t k = 0; k < prog->n_consts; ++k)
        if (!gl3_valid(prog->consts + 3 * (size_t)k)) return fail(SS_ERR_INVALID, "constant %u is not an element of the extension", k);
    std::vector<uint32_t> tdesc(2 * (size_t)(prog->n_tables ? prog->n_tables : 1), 0);
    for (uint32_t t = 0; t < prog->n_tables; ++t) {
        if (prog->table_desc[2 * t + 1] > 30) return fail(SS_ERR_INVALID, "table %u: length out of range", t);
        tdesc[2 * t] = prog->table_desc[2 * t];
        tdesc[2 * t + 1] = (uint32_t)((1ull << prog->table_desc[2 * t + 1]) - 1ull);
    }
    const uint64_t lanes = gl3_vm_lanes(N);
    const size_t code_bytes = (size_t)prog->n_instr * 8, const_bytes = (size_t)(prog->n_consts ? prog->n_consts : 1) * 24;
    ss_status st = ctx->ensure_scratch(code_bytes + const_bytes + tdesc.size() * 4 + 256);
    if (st != SS_OK) return st;
    // the plain layout's composition has a compiled kernel (quotient_gen_plain_gl.inc); SS_QUOTIENT_INTERPRET=1 forces the interpreter
    const bool compiled = !getenv("SS_QUOTIENT_INTERPRET") && ncols <= (uint32_t)MAX_COLS &&
                          gl3_compiled_matches(prog->code, prog->n_instr, prog->consts, prog->n_consts, prog->n_tables);
    if (!compiled) {
        st = ctx->ensure_scratch2((size_t)(prog->n_slots ? prog->n_slots : 1) * 3 * lanes * 8);
        if (st != SS_OK) return st;
    }
    char *p = (char *)ctx->scratch;
    uint64_t *d_consts = (uint64_t *)p; p += const_bytes;
    uint32_t *d_code = (uint32_t *)p; p += code_bytes;
    uint32_t *d_tdesc = (uint32_t *)p;
    hipStream_t s = ctx->stream;
    if (prog->n_instr) HIP_TRY(hipMemcpyAsync(d_code, prog->code, code_bytes, hipMemcpyHostToDevice, s));
    if (prog->n_consts) HIP_TRY(hipMemcpyAsync(d_consts, prog->consts, (size_t)prog->n_consts * 24, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_tdesc, tdesc.data(), tdesc.size() * 4, hipMemcpyHostToDevice, s));
    {
        ss_ctx::Scope prof(ctx, SS_PROF_QUOTIENT);
        if (compiled)
            HIP_TRY(launch_gl3_plain(s, d_consts, prog->d_tables, d_tdesc, d_lde_cols, ncols, d_out, offset, gl_root_of_unity_host(log_n + log_blowup),
                                     log_blowup, N));
        else
            HIP_TRY(launch_gl3_vm(s, d_code, prog->n_instr, d_consts, prog->d_tables, d_tdesc, d_lde_cols, ncols, (uint64_t *)ctx->scratch2, d_out, offset,
                                  gl_root_of_unity_host(log_n + log_blowup), log_blowup, N));
    }
    HIP_TRY(hipStreamSynchronize(s));
    return SS_OK;
}

ss_status ss_ood_eval_gl64x3(ss_ctx *ctx, const uint64_t *const *d_coeffs_bitrev, uint32_t ncols, uint32_t log_n, const uint32_t *cell_col,
                             const uint32_t *cell_off, uint32_t ncells, const uint64_t z[3], uint64_t *out) {
    if (!ctx || !d_coeffs_bitrev || !z || (ncells && (!cell_col || !cell_off || !out))) return fail(SS_ERR_INVALID, "NULL argument");
    if (!gl_valid_log(log_n)) return fail(SS_ERR_INVALID, "log_n out of range");
    if (!gl3_valid(z)) return fail(SS_ERR_INVALID, "z is not an element of the extension");
    for (uint32_t j = 0; j < ncells; ++j)
        if (cell_col[j] >= ncols) return fail(SS_ERR_INVALID, "cell %u names column %u", j, cell_col[j]);
    if (!ncells) return SS_OK;
    const uint64_t n = 1ull << log_n;
    // P(z w^k) = sum_j (c_j z^j) w^(jk): scale the (bit-reversed) coefficients by z^j - three Fp arrays - and transform each; the
    // k-th output of component t is the t-th coordinate of P(z w^k).  One table of powers serves every column.
    const uint64_t *tw = nullptr;
    ss_status st = gl_get_plan(ctx, log_n, false, 1, &tw);
    if (st != SS_OK) return st;
    // columns in batches of five: 15 component arrays per transform launch
    constexpr uint32_t BATCH = 5;
    st = ctx->ensure_scratch2((3 + 3 * BATCH) * n * 8 + (size_t)ncells * (8 + 24) + 64);
    if (st != SS_OK) return st;
    uint64_t *zp = (uint64_t *)ctx->scratch2, *comp = zp + 3 * n, *d_idx = comp + 3 * BATCH * n, *d_vals = d_idx + ncells;
    hipStream_t s = ctx->stream;
    HIP_TRY(launch_gl3_zpow_bitrev(s, zp, zp + n, zp + 2 * n, log_n, z));
    std::vector<uint32_t> used;                                  // columns some cell names at a row offset other than 0
    for (uint32_t col = 0; col < ncols; ++col) {
        bool any = false, shifted = false;
        for (uint32_t j = 0; j < ncells; ++j) if (cell_col[j] == col) { any = true; shifted |= (cell_off[j] & (n - 1)) != 0; }
        if (any && shifted) { used.push_back(col); continue; }
        if (!any) continue;
        // read at z only: P(z) = sum_j c_j z^j is one pass over the coefficients against the power table, not three transforms
        const uint32_t nb = gl3_dot_blocks(n);
        uint64_t *d_part = comp;                                 // the component area is free here
        HIP_TRY(launch_gl3_dot(s, d_coeffs_bitrev[col], zp, zp + n, zp + 2 * n, n, d_part));
        std::vector<uint64_t> part(3 * (size_t)nb);
        HIP_TRY(hipMemcpyAsync(part.data(), d_part, part.size() * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        uint64_t sum[3] = {0, 0, 0};
        for (uint32_t b = 0; b < nb; ++b) for (int t = 0; t < 3; ++t) sum[t] = gl_addh(sum[t], part[3 * (size_t)b + t]);
        for (uint32_t j = 0; j < ncells; ++j) if (cell_col[j] == col) memcpy(out + 3 * (size_t)j, sum, 24);
    }
    std::vector<uint64_t> idx(ncells), vals(3 * (size_t)ncells);
    for (size_t b0 = 0; b0 < used.size(); b0 += BATCH) {
        const uint32_t nb = (uint32_t)std::min<size_t>(BATCH, used.size() - b0);
        const void *src[3 * BATCH]; void *dst[3 * BATCH];
        for (uint32_t k = 0; k < nb; ++k) {
            uint64_t *c0 = comp + (size_t)3 * k * n;
            HIP_TRY(launch_gl3_scale_columns(s, d_coeffs_bitrev[used[b0 + k]], zp, zp + n, zp + 2 * n, n, c0, c0 + n, c0 + 2 * n));
            for (int t = 0; t < 3; ++t) { src[3 * k + t] = c0 + t * n; dst[3 * k + t] = c0 + t * n; }
        }
        st = gl_run_forward(ctx, src, dst, 3 * nb, log_n, tw, 0);
        if (st != SS_OK) return st;
        for (uint32_t k = 0; k < nb; ++k) {
            std::vector<uint32_t> mine;
            for (uint32_t j = 0; j < ncells; ++j) if (cell_col[j] == used[b0 + k]) mine.push_back(j);
            for (size_t t = 0; t < mine.size(); ++t) idx[t] = cell_off[mine[t]] & (n - 1);
            uint64_t *c0 = comp + (size_t)3 * k * n;
            HIP_TRY(hipMemcpyAsync(d_idx, idx.data(), mine.size() * 8, hipMemcpyHostToDevice, s));
            HIP_TRY(launch_gl3_gather(s, c0, c0 + n, c0 + 2 * n, d_idx, (uint32_t)mine.size(), d_vals));
            HIP_TRY(hipMemcpyAsync(vals.data(), d_vals, 3 * mine.size() * 8, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            for (size_t t = 0; t < mine.size(); ++t) memcpy(out + 3 * (size_t)mine[t], vals.data() + 3 * t, 24);
        }
    }
    return SS_OK;
}

ss_status ss_deep_compose_gl64x3(ss_ctx *ctx, const uint64_t *const *d_trace_lde, uint32_t ntrace_cols, const uint64_t *const *d_comp_lde,
                                 uint32_t ncomp, uint32_t log_n, uint32_t log_blowup, uint64_t offset, const uint32_t *mask_col,
                                 const uint32_t *mask_off, uint32_t nmask, const uint64_t *ood_trace, const uint64_t *coeff_trace,
                                 const uint64_t *ood_comp, const uint64_t *coeff_comp, const uint64_t z[3], const uint64_t z_comp[3],
                                 uint64_t *d_out) {
    if (!ctx || !d_trace_lde || !z || !d_out) return fail(SS_ERR_INVALID, "NULL argument");
    if (nmask && (!mask_col || !mask_off || !ood_trace || !coeff_trace)) return fail(SS_ERR_INVALID, "NULL mask argument");
    if (ncomp && (!d_comp_lde || !ood_comp || !coeff_comp || !z_comp)) return fail(SS_ERR_INVALID, "NULL composition argument");
    if (!gl_valid_log(log_n) || !gl_valid_log(log_n + log_blowup)) return fail(SS_ERR_INVALID, "size out of range");
    if (ntrace_cols > (uint32_t)MAX_COLS || ncomp > 12) return fail(SS_ERR_UNSUPPORTED, "too many columns");
    if (offset == 0 || offset >= GL_P) return fail(SS_ERR_INVALID, "the coset offset is not a non-zero field element");
    if (!gl3_valid(z) || (ncomp && !gl3_valid(z_comp))) return fail(SS_ERR_INVALID, "z is not an element of the extension");
    const uint64_t n = 1ull << log_n, N = n << log_blowup;
    const uint64_t wn = gl_root_of_unity_host(log_n), wn_inv = gl_inv_host(wn);
    // taps of the denominator table, column by column (csrc/deep.hip's layout), plus the column of constants -K_off
    std::vector<uint32_t> order(nmask);
    for (uint32_t j = 0; j < nmask; ++j) {
        if (mask_col[j] >= ntrace_cols) return fail(SS_ERR_INVALID, "mask cell %u names column %u", j, mask_col[j]);
        if (!gl3_valid(ood_trace + 3 * j) || !gl3_valid(coeff_trace + 3 * j)) return fail(SS_ERR_INVALID, "mask cell %u: value out of range", j);
        order[j] = j;
    }
    auto off_of = [&](uint32_t j) { return mask_off[j] & (uint32_t)(n - 1); };
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        return mask_col[x] != mask_col[y] ? mask_col[x] < mask_col[y] : off_of(x) < off_of(y);
    });
    std::map<uint32_t, HGl3> k_of;
    std::vector<uint32_t> tap_shift, cdesc;
    std::vector<uint64_t> tap_coef;
    auto load3 = [](const uint64_t *p) { return HGl3{{p[0], p[1], p[2]}}; };
    // a tap's coefficient as the kernel's right-hand factor (gl64.h Gl3Rhs): c0, c1, c2, 2 c1, 2 c2
    auto push_tap = [&](const HGl3 &c) {
        for (int t = 0; t < 3; ++t) tap_coef.push_back(c.c[t]);
        tap_coef.push_back(gl_addh(c.c[1], c.c[1])); tap_coef.push_back(gl_addh(c.c[2], c.c[2]));
    };
    for (uint32_t t = 0; t < nmask;) {
        const uint32_t col = mask_col[order[t]], first = (uint32_t)tap_shift.size();
        for (; t < nmask && mask_col[order[t]] == col; ++t) {
            const uint32_t j = order[t], offv = off_of(j);
            const HGl3 cprime = gl3_scaleh(load3(coeff_trace + 3 * j), gl_pow_host(wn_inv, offv));
            auto it = k_of.find(offv);
            if (it == k_of.end()) it = k_of.emplace(offv, HGl3{{0, 0, 0}}).first;
            it->second = gl3_addh(it->second, gl3_mulh(cprime, load3(ood_trace + 3 * j)));
            tap_shift.push_back(offv);
            push_tap(cprime);
        }
        cdesc.push_back(col); cdesc.push_back(first); cdesc.push_back((uint32_t)tap_shift.size() - first);
    }
    if (nmask) {
        const uint32_t first = (uint32_t)tap_shift.size();
        for (auto &kv : k_of) { tap_shift.push_back(kv.first); push_tap(HGl3{{gl_subh(0, kv.second.c[0]), gl_subh(0, kv.second.c[1]), gl_subh(0, kv.second.c[2])}}); }
        cdesc.push_back(0xffffffffu); cdesc.push_back(first); cdesc.push_back((uint32_t)tap_shift.size() - first);
    }
    const uint32_t ntaps = (uint32_t)tap_shift.size(), ncoldesc = (uint32_t)(cdesc.size() / 3);
    HGl3 comp_k{{0, 0, 0}};
    for (uint32_t k = 0; k < ncomp; ++k) {
        if (!gl3_valid(ood_comp + 3 * k) || !gl3_valid(coeff_comp + 3 * k)) return fail(SS_ERR_INVALID, "composition cell %u: value out of range", k);
        comp_k = gl3_addh(comp_k, gl3_mulh(load3(coeff_comp + 3 * k), load3(ood_comp + 3 * k)));
    }
    // device: tables D, Dc [n][3], the sub-coset values [3][n], then per component iNTT(n) + coset NTT(N), interleaved into d_out
    const size_t small = (size_t)(ntaps + 1) * (4 + 40) + (size_t)(ncoldesc + 1) * 12 + (size_t)(ncomp + 1) * 24 + 256;
    ss_status st = ctx->ensure_scratch(small);
    if (st != SS_OK) return st;
    st = ctx->ensure_scratch2((6 * n + 3 * n + 3 * N) * 8);
    if (st != SS_OK) return st;
    uint64_t *D = (uint64_t *)ctx->scratch2, *Dc = D + 3 * n, *sub = Dc + 3 * n, *lde = sub + 3 * n;
    char *p = (char *)ctx->scratch;
    uint64_t *d_tap_coef = (uint64_t *)p; p += (size_t)(ntaps + 1) * 40;
    uint64_t *d_comp_coef = (uint64_t *)p; p += (size_t)(ncomp + 1) * 24;
    uint32_t *d_tap_shift = (uint32_t *)p; p += (size_t)(ntaps + 1) * 4;
    uint32_t *d_cdesc = (uint32_t *)p;
    hipStream_t s = ctx->stream;
    if (ntaps) {
        HIP_TRY(hipMemcpyAsync(d_tap_coef, tap_coef.data(), (size_t)ntaps * 40, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_tap_shift, tap_shift.data(), (size_t)ntaps * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_cdesc, cdesc.data(), (size_t)ncoldesc * 12, hipMemcpyHostToDevice, s));
    }
    if (ncomp) HIP_TRY(hipMemcpyAsync(d_comp_coef, coeff_comp, (size_t)ncomp * 24, hipMemcpyHostToDevice, s));
    const uint64_t *tw_inv = nullptr, *tw_fwd = nullptr;
    st = gl_get_plan(ctx, log_n, true, offset, &tw_inv);
    if (st != SS_OK) return st;
    st = gl_get_plan(ctx, log_n + log_blowup, false, offset, &tw_fwd);
    if (st != SS_OK) return st;
    {
        ss_ctx::Scope prof(ctx, SS_PROF_DEEP);
        HIP_TRY(launch_gl3_inverse_table(s, D, n, offset, wn, z));
        if (ncomp) HIP_TRY(launch_gl3_inverse_table(s, Dc, n, offset, wn, z_comp));
        HIP_TRY(launch_gl3_deep(s, d_trace_lde, ntrace_cols, d_comp_lde, ncomp, D, Dc, d_tap_shift, d_tap_coef, d_cdesc, ncoldesc, d_comp_coef,
                                comp_k.c, n, log_blowup, sub, sub + n, sub + 2 * n));
    }
    const void *src[3] = {sub, sub + n, sub + 2 * n};
    void *co[3] = {sub, sub + n, sub + 2 * n};
    void *ev[3] = {lde, lde + N, lde + 2 * N};
    st = gl_run_inverse(ctx, src, co, 3, log_n, tw_inv);           // values on offset * <w_n> -> bit-reversed coefficients
    if (st != SS_OK) return st;
    st = gl_run_forward(ctx, (const void *const *)co, ev, 3, log_n + log_blowup, tw_fwd, log_blowup);
    if (st != SS_OK) return st;
    HIP_TRY(launch_gl3_interleave(s, lde, lde + N, lde + 2 * N, N, d_out));
    HIP_TRY(hipStreamSynchronize(s));
    return SS_OK;
}

ss_status ss_fri_fold_gl64x3(ss_ctx *ctx, const uint64_t *d_evals, uint32_t log_len, uint32_t fold, const uint64_t alpha[3],
                             uint64_t domain_offset, uint32_t flags, uint64_t *d_out) {
    if (!ctx || !d_evals || !alpha || !d_out) return fail(SS_ERR_INVALID, "NULL argument");
    if (fold != 2 && fold != 4 && fold != 8 && fold != 16) return fail(SS_ERR_INVALID, "fold must be 2, 4, 8 or 16");
    uint32_t log_fold = 0;
    while ((1u << log_fold) < fold) ++log_fold;
    if (!gl_valid_log(log_len) || log_len < log_fold) return fail(SS_ERR_INVALID, "layer length out of range");
    if (domain_offset == 0 || domain_offset >= GL_P) return fail(SS_ERR_INVALID, "the domain offset is not a non-zero field element");
    if (flags & ~(uint32_t)SS_FRI_UNNORMALISED) return fail(SS_ERR_UNSUPPORTED, "only SS_FRI_UNNORMALISED is defined for this field");
    ss_ctx::Scope prof(ctx, SS_PROF_FRI);
    HIP_TRY(launch_gl3_fri_fold(ctx->stream, d_evals, log_len, fold, alpha, domain_offset, (flags & SS_FRI_UNNORMALISED) != 0, d_out));
    return SS_OK;
}

}  // extern "C"
