// TEST INFRASTRUCTURE ONLY - the scheduler of tests/hipemu/hip/hip_runtime.h: one workgroup at a time, its lanes as fibers
// resumed round-robin, one barrier-to-barrier stretch per pass.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <cstdio>
#include <thread>
#include <vector>

thread_local hipemu_uint3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

// the dynamically sized LDS arrays the kernels declare `extern __shared__` (here: extern thread_local), at the device's limit
namespace ss {
thread_local uint64_t gl_tile[160 * 1024 / 8];                                          // goldilocks.hip
thread_local __attribute__((aligned(16))) unsigned char smem[160 * 1024];               // ntt.hip
}  // namespace ss
// quotient_gen.h declares its array inside an unnamed namespace (one per generated file): the declarations all mangle to this name
thread_local __attribute__((aligned(16))) unsigned char hipemu_qg_smem[160 * 1024] asm("_ZN2ss12_GLOBAL__N_17qg_smemE");

// a fiber switch that is nothing but the callee-saved registers and the stack pointer (x86-64 System V): swapcontext() would also make a
// signal-mask system call per switch, and a launch switches twice per lane
extern "C" void hipemu_switch(void **save_sp, void *load_sp);
asm(R"(
    .text
    .globl hipemu_switch
    .type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipemu_switch, .-hipemu_switch
)");

namespace {
constexpr size_t STACK_BYTES = 1 << 20;                      // the generated constraint kernels keep a few hundred 36-byte values live
struct Fiber {
    void *sp = nullptr;
    char *stack = nullptr;
    bool done = true;
    hipemu_uint3 tid{0, 0, 0};
};
thread_local std::vector<Fiber> fibers;
thread_local void *sched_sp = nullptr;
thread_local Fiber *current = nullptr;
thread_local const std::function<void()> *body = nullptr;
thread_local std::vector<int> exchange;                      // __shfl_xor: one word per lane of the workgroup
const int block_order = [] { const char *e = getenv("HIPEMU_BLOCKS"); return e && e[0] == 'r' ? 1 : 0; }();
const int lane_order = [] { const char *e = getenv("HIPEMU_ORDER"); return !e ? 0 : e[0] == 'r' ? 1 : e[0] == 's' ? 2 : 0; }();   // forward / reverse / shuffle

void fiber_main() {
    (*body)();
    current->done = true;
    hipemu_switch(&current->sp, sched_sp);
    abort();                                                 // a finished lane is never resumed
}
void fiber_prepare(Fiber &f) {
    // [r15 r14 r13 r12 rbx rbp][entry][fake return address]: after hipemu_switch pops and returns, fiber_main starts with the
    // stack a call would have left (rsp = 8 mod 16)
    void **top = (void **)(((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15);
    *--top = nullptr;
    *--top = (void *)&fiber_main;
    for (int i = 0; i < 6; ++i) *--top = nullptr;
    f.sp = top;
}
}  // namespace

void hipemu_barrier() {
    Fiber *me = current;
    hipemu_switch(&me->sp, sched_sp);                        // resumed in the next pass, after every live lane got here
    threadIdx = me->tid;
}

int hipemu_shfl_xor(int v, int mask, int width) {
    const unsigned lane = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    exchange[lane] = v;
    hipemu_barrier();
    const unsigned partner = (lane & ~(unsigned)(width - 1)) | ((lane ^ (unsigned)mask) & (unsigned)(width - 1));
    const int r = exchange[partner];
    hipemu_barrier();
    return r;
}

namespace {
// the workgroups [first, last) of a launch (x fastest), on the calling OS thread: all the state above is thread_local
void run_blocks(dim3 grid, dim3 block, const std::function<void()> &lane_body, uint64_t first, uint64_t last) {
    const size_t lanes = (size_t)block.x * block.y * block.z;
    if (fibers.size() < lanes) fibers.resize(lanes);
    for (size_t i = 0; i < lanes; ++i)
        if (!fibers[i].stack) fibers[i].stack = (char *)malloc(STACK_BYTES);
    exchange.assign(lanes, 0);
    body = &lane_body;
    gridDim = grid; blockDim = block;
    for (uint64_t bb = first; bb < last; ++bb) {
        // HIPEMU_BLOCKS=reverse: workgroups in descending order (with HIPEMU_THREADS=1: strictly) - nothing may depend on the order
        const uint64_t b = block_order ? first + (last - 1 - bb) : bb;
        blockIdx = {(unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((uint64_t)grid.x * grid.y))};
        size_t i = 0;
        for (unsigned tz = 0; tz < block.z; ++tz)
            for (unsigned ty = 0; ty < block.y; ++ty)
                for (unsigned tx = 0; tx < block.x; ++tx, ++i) {
                    Fiber &f = fibers[i];
                    fiber_prepare(f);
                    f.done = false;
                    f.tid = {tx, ty, tz};
                }
        for (size_t live = lanes, pass = 0; live; ++pass) {  // one pass = every live lane from its barrier to the next (or to its end)
            live = 0;
            for (size_t j = 0; j < lanes; ++j) {
                // HIPEMU_ORDER: between two barriers the lanes of a workgroup may run in any order - forward (default), reverse, or a
                // different rotation + direction every pass; code that is missing a barrier passes in one order and fails in another
                size_t k = j;
                if (lane_order == 1) k = lanes - 1 - j;
                else if (lane_order == 2) { const size_t rot = (pass * 7919 + b * 104729) % lanes; k = ((pass + b) & 1) ? (lanes - 1 - j + rot) % lanes : (j + rot) % lanes; }
                Fiber &f = fibers[k];
                if (f.done) continue;
                current = &f;
                threadIdx = f.tid;
                hipemu_switch(&sched_sp, f.sp);
                if (!f.done) ++live;
            }
        }
    }
    body = nullptr;
}
}  // namespace

// Workgroups are independent (the library's only cross-workgroup traffic is atomicMin), so a large grid is spread over a few OS
// threads - a pool that lives as long as the library, so that each worker keeps its fiber stacks - each taking chunks of workgroups
// off a shared counter.  HIPEMU_THREADS=1 keeps everything on the calling thread.
namespace {
struct Pool {
    std::mutex m;
    std::condition_variable wake, idle;
    std::vector<std::thread> threads;
    uint64_t generation = 0;
    unsigned running = 0;
    bool stop = false;
    // the current launch
    dim3 grid, block;
    const std::function<void()> *lane_body = nullptr;
    std::atomic<uint64_t> next{0};
    uint64_t nblocks = 0, chunk = 1;

    void work() {
        for (;;) {
            const uint64_t b0 = next.fetch_add(chunk);
            if (b0 >= nblocks) break;
            run_blocks(grid, block, *lane_body, b0, std::min(nblocks, b0 + chunk));
        }
    }
    void worker_main() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            wake.wait(lk, [&] { return stop || generation != seen; });
            if (stop) return;
            seen = generation;
            lk.unlock();
            work();
            lk.lock();
            if (--running == 0) idle.notify_all();
        }
    }
    void launch(unsigned nthreads, dim3 g, dim3 b, const std::function<void()> &body_fn, uint64_t nb) {
        while (threads.size() + 1 < nthreads) threads.emplace_back([this] { worker_main(); });
        {
            std::lock_guard<std::mutex> lk(m);
            grid = g; block = b; lane_body = &body_fn; nblocks = nb;
            chunk = std::max<uint64_t>(1, nb / (nthreads * 8ull));
            next.store(0);
            running = (unsigned)threads.size();
            ++generation;
        }
        wake.notify_all();
        work();
        std::unique_lock<std::mutex> lk(m);
        idle.wait(lk, [&] { return running == 0; });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        wake.notify_all();
        for (auto &t : threads) t.join();
    }
};
Pool &pool() { static Pool *p = new Pool; return *p; }     // never destroyed: workers may outlive static destruction order otherwise
}  // namespace

void hipemu_launch(dim3 grid, dim3 block, const std::function<void()> &lane_body) {
    const uint64_t nblocks = (uint64_t)grid.x * grid.y * grid.z;
    static const unsigned max_threads = [] {
        const char *e = getenv("HIPEMU_THREADS");
        unsigned t = e ? (unsigned)atoi(e) : std::thread::hardware_concurrency();
        return t < 1 ? 1u : t > 16 ? 16u : t;
    }();
    if (nblocks < 32 || max_threads <= 1) { run_blocks(grid, block, lane_body, 0, nblocks); return; }
    pool().launch(max_threads, grid, block, lane_body, nblocks);
}
