// trace_common.hpp — what the layouts' base-trace generators (trace_recursive.cpp, trace_starknet.cpp) share: the
// instruction word (binary/src/lib.rs:565-721), memory access, the curve, the Pedersen builtin's element steps
// (builtins/src/pedersen/mod.rs:121-163) and the diluted form of the bitwise builtin (builtins/src/bitwise/mod.rs).
#pragma once
#include <omp.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <cstdint>
#include <cstring>
#include <exception>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "trace_recursive.hpp"

namespace ssh {
namespace tracedetail {

constexpr uint32_t DILUTED_N_BITS = 16, DILUTED_SPACING = 4;
constexpr uint64_t HALF_OFFSET = 1ull << 15;
// enum Flag (binary/src/lib.rs:740-772)
enum { F_DST_REG, F_OP0_REG, F_OP1_IMM, F_OP1_FP, F_OP1_AP, F_RES_ADD, F_RES_MUL, F_PC_JUMP_ABS, F_PC_JUMP_REL, F_PC_JNZ, F_AP_ADD, F_AP_ADD1,
       F_OPCODE_CALL, F_OPCODE_RET, F_OPCODE_ASSERT_EQ, F_ZERO };


[[noreturn]] inline void fail(const std::string &m) { throw std::runtime_error("trace: " + m); }

// the address half of the memory pool as integers (sorting, gap search): a vector whose elements are NOT value-initialised when it is
// sized - the generators assign every element in their parallel first pass, and a serial fill of its 64 MB (2^20 steps) took as long as
// a builtin's section
template <class T> struct default_init_allocator : std::allocator<T> {
    template <class U> struct rebind { using other = default_init_allocator<U>; };
    template <class U> void construct(U *p) { ::new (static_cast<void *>(p)) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
using AddrArray = std::vector<uint64_t, default_init_allocator<uint64_t>>;

using CountArray = std::vector<uint32_t, default_init_allocator<uint32_t>>;
// The range-check pool's ordered values (utils.rs:357-380): every v of [lo, hi] max(count[v], 1) times, ascending - RUNS, laid out by a
// prefix sum and filled by all threads (3.3 million entries at 2^20 steps, a third of them one offset of an idling run: pushed one by one
// on one thread they took half of the section that makes them, before any column can be written)
inline void ordered_runs(const std::vector<uint32_t> &count, uint32_t lo, uint32_t hi, CountArray &out) {
    out.clear();
    if (lo > hi) return;
    std::vector<uint64_t> first((size_t)(hi - lo) + 2, 0);
    for (uint32_t v = lo; v <= hi; ++v) first[v - lo + 1] = first[v - lo] + std::max(count[v], 1u);
    out.resize(first.back());
    uint32_t *o = out.data();
    constexpr uint64_t LONG_RUN = 1 << 14;
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t v = lo; v <= (int64_t)hi; ++v) {
        const uint64_t a = first[(size_t)v - lo], b = first[(size_t)v - lo + 1];
        if (b - a < LONG_RUN) std::fill(o + a, o + b, (uint32_t)v);
    }
    for (uint32_t v = lo; v <= hi; ++v) {
        const uint64_t a = first[v - lo], b = first[v - lo + 1];
        if (b - a < LONG_RUN) continue;
#pragma omp parallel for schedule(static)
        for (int64_t k = (int64_t)a; k < (int64_t)b; ++k) o[k] = v;
    }
}

inline bool felt_is_zero(const Felt &f) { return (f[0] | f[1] | f[2] | f[3]) == 0; }
inline bool felt_eq(const Felt &a, const Felt &b) { return a == b; }
inline U256 shr(const U256 &v, unsigned k) {
    U256 r{};
    const unsigned w = k / 64, b = k % 64;
    for (unsigned i = 0; i + w < 4; ++i) {
        r[i] = v[i + w] >> b;
        if (b && i + w + 1 < 4) r[i] |= v[i + w + 1] << (64 - b);
    }
    return r;
}
inline bool bit(const U256 &v, unsigned k) { return (v[k / 64] >> (k % 64)) & 1; }

struct Word {                       // binary/src/lib.rs:565-721 (instruction fields live in the low 63 bits)
    uint64_t w;
    bool flag(int f) const { return (w >> (48 + f)) & 1; }
    uint64_t flag_prefix(int f) const { return f == F_ZERO ? 0 : (w >> (48 + f)) & ((1ull << (15 - f)) - 1); }
    uint64_t off_dst() const { return w & 0xffff; }
    uint64_t off_op0() const { return (w >> 16) & 0xffff; }
    uint64_t off_op1() const { return (w >> 32) & 0xffff; }
    int op1_src() const { return flag(F_OP1_IMM) + 2 * flag(F_OP1_FP) + 4 * flag(F_OP1_AP); }
    int res_logic() const { return flag(F_RES_ADD) + 2 * flag(F_RES_MUL); }
    int pc_update() const { return flag(F_PC_JUMP_ABS) + 2 * flag(F_PC_JUMP_REL) + 4 * flag(F_PC_JNZ); }
};

struct Mem {
    const std::vector<U256> &m;
    const std::vector<uint8_t> &present;
    const U256 &at(uint64_t a) const {
        if (a >= m.size() || !present[a]) fail("the run reads memory cell " + std::to_string(a) + " that memory.bin does not hold");
        return m[a];
    }
    uint64_t small(uint64_t a) const {
        const U256 &v = at(a);
        if (v[1] | v[2] | v[3]) fail("memory cell " + std::to_string(a) + " is used as an address but is not one");
        return v[0];
    }
};

// ---- curve y^2 = x^3 + x + beta, affine (builtins/src/utils.rs)
struct Pt { Felt x, y; };
inline Pt ec_double(const Pt &p) {
    const Felt xx = felt_mul(p.x, p.x);
    const Felt lam = felt_mul(felt_add(felt_add(felt_add(xx, xx), xx), felt_from_u64(1)), felt_inv(felt_add(p.y, p.y)));
    const Felt x3 = felt_sub(felt_mul(lam, lam), felt_add(p.x, p.x));
    return Pt{x3, felt_sub(felt_mul(lam, felt_sub(p.x, x3)), p.y)};
}
inline Pt ec_add(const Pt &a, const Pt &b) {
    if (felt_eq(a.x, b.x)) {
        if (!felt_eq(a.y, b.y)) fail("point at infinity in a Pedersen partial sum");
        return ec_double(a);
    }
    const Felt lam = felt_mul(felt_sub(b.y, a.y), felt_inv(felt_sub(b.x, a.x)));
    const Felt x3 = felt_sub(felt_sub(felt_mul(lam, lam), a.x), b.x);
    return Pt{x3, felt_sub(felt_mul(lam, felt_sub(a.x, x3)), a.y)};
}
// ---- the same curve in Jacobian coordinates, for CHAINS of steps: a chain of doublings or of additions costs one inversion (the batch
// below) instead of one per step - a signature's trace is ~3600 affine steps, a scalar multiplication's ~770
// v[k] <- 1 / v[k] with one inversion (Montgomery's trick); an element without an inverse is the reference's inverse().unwrap() panic
inline void batch_invert(Felt *v, size_t n) {
    if (!n) return;
    std::vector<Felt> prefix(n);
    for (size_t k = 0; k < n; ++k) {
        if (felt_is_zero(v[k])) fail("a curve step divides by zero");
        prefix[k] = k ? felt_mul(prefix[k - 1], v[k]) : v[k];
    }
    Felt run = felt_inv(prefix[n - 1]);
    for (size_t k = n; k-- > 0;) {
        const Felt inv = k ? felt_mul(run, prefix[k - 1]) : run;
        if (k) run = felt_mul(run, v[k]);
        v[k] = inv;
    }
}
struct Jac { Felt X, Y, Z; };                              // (X / Z^2, Y / Z^3)
inline Jac jac_of(const Pt &p) { return Jac{p.x, p.y, felt_from_u64(1)}; }
inline Jac jac_double(const Jac &p) {                      // a = 1: M = 3 X^2 + Z^4, S = 4 X Y^2
    const Felt xx = felt_mul(p.X, p.X), yy = felt_mul(p.Y, p.Y), yyyy = felt_mul(yy, yy), zz = felt_mul(p.Z, p.Z);
    const Felt t = felt_mul(p.X, yy), t2 = felt_add(t, t), s4 = felt_add(t2, t2);
    const Felt m = felt_add(felt_add(felt_add(xx, xx), xx), felt_mul(zz, zz));
    const Felt x3 = felt_sub(felt_mul(m, m), felt_add(s4, s4));
    const Felt y4 = felt_add(yyyy, yyyy), y8 = felt_add(y4, y4);
    const Felt yz = felt_mul(p.Y, p.Z);
    return Jac{x3, felt_sub(felt_mul(m, felt_sub(s4, x3)), felt_add(y8, y8)), felt_add(yz, yz)};
}
// p + q for an affine q that does not share p's x (h = q.x Z^2 - X, not zero, is the caller's: it has looked at it already)
inline Jac jac_add_affine(const Jac &p, const Pt &q, const Felt &zz, const Felt &h) {
    const Felt r = felt_sub(felt_mul(q.y, felt_mul(p.Z, zz)), p.Y);
    const Felt hh = felt_mul(h, h), hhh = felt_mul(h, hh), v = felt_mul(p.X, hh);
    const Felt x3 = felt_sub(felt_sub(felt_mul(r, r), hhh), felt_add(v, v));
    return Jac{x3, felt_sub(felt_mul(r, felt_sub(v, x3)), felt_mul(p.Y, hhh)), felt_mul(p.Z, h)};
}
inline std::vector<Pt> batch_normalize(const std::vector<Jac> &j) {
    std::vector<Felt> zi(j.size());
    for (size_t k = 0; k < j.size(); ++k) zi[k] = j[k].Z;
    batch_invert(zi.data(), zi.size());
    std::vector<Pt> out(j.size());
    for (size_t k = 0; k < j.size(); ++k) {
        const Felt zi2 = felt_mul(zi[k], zi[k]);
        out[k] = Pt{felt_mul(j[k].X, zi2), felt_mul(j[k].Y, felt_mul(zi2, zi[k]))};
    }
    return out;
}

// get_res of a conditional jump is dst^-1 (trace.rs:172-232): the reference inverts per such cycle - one cycle in ten of a real program,
// ten microseconds each.  The jnz cycles of a BLOCK of cycles are looked up first and inverted together; the block's main pass then
// takes them in the same order.  A cycle the main pass is going to refuse (not an instruction, a cell memory.bin does not hold) is
// skipped here - the main pass says why.  take() checks the product, so an inverse can never land on another cycle's cell.
struct JnzInverses {
    std::vector<Felt> inv;
    size_t next = 0;
    JnzInverses(const RegisterStates &states, const Mem &mem, uint64_t first_cycle, uint64_t end_cycle) {
        auto readable = [&](uint64_t a) { return a < mem.m.size() && mem.present[a]; };
        for (uint64_t cycle = first_cycle; cycle < end_cycle; ++cycle) {
            const RegisterState st = states[cycle];
            if (!readable(st.pc)) continue;
            const U256 &iw = mem.m[st.pc];
            const Word w{iw[0]};
            if ((iw[1] | iw[2] | iw[3]) || w.flag(F_ZERO) || w.pc_update() != 4) continue;
            const uint64_t dst_addr = w.off_dst() + (w.flag(F_DST_REG) ? st.fp : st.ap) - HALF_OFFSET;
            if (!readable(dst_addr)) continue;
            const Felt dst = felt_from_canonical(mem.m[dst_addr]);
            if (!felt_is_zero(dst)) inv.push_back(dst);
        }
        batch_invert(inv.data(), inv.size());
    }
    size_t fallbacks = 0;                                   // cycles the pre-pass and the main pass disagreed about (none, unless the two decodes drift apart)
    Felt take(const Felt &dst) {                           // dst is not zero
        static const Felt one = felt_from_u64(1);
        if (next < inv.size() && felt_eq(felt_mul(inv[next], dst), one)) return inv[next++];
        ++fallbacks;
        return felt_inv(dst);
    }
    ~JnzInverses() {
        static const bool timing = getenv("SSH_TRACE_TIMING") != nullptr;
        if (timing && (fallbacks || next != inv.size())) fprintf(stderr, "[trace timing] jnz inverses: %zu cycles took the per-cycle inversion, %zu batched inverses unused\n", fallbacks, inv.size() - next);
    }
};

// builtins/src/pedersen/constants.rs:5-30, canonical little-endian limbs
inline constexpr uint64_t PEDERSEN_POINTS[5][2][4] = {
    {{0x551fde4050ca6804ull, 0x716b0b1022947733ull, 0x00ee1b87eb599f16ull, 0x049ee3eba8c16007ull}, {0xd0405d266e10268aull, 0x4e621062c0e056c1ull, 0xf346d49d06ea0ed3ull, 0x03ca0cfe4b3bc6ddull}},
    {{0x1080d17957ebe47bull, 0x8fa8120b6d56eb0cull, 0x969c748655fca9e5ull, 0x0234287dcbaffe7full}, {0x6ed0268ee89e5615ull, 0x940135dd7a6c94ccull, 0x1e889527d41f4e39ull, 0x03b056f100f96fb2ull}},
    {{0xb7a6932dba8aa378ull, 0x99099ec1de5e3018ull, 0x3f9dab2656558f33ull, 0x04fa56f376c83db3ull}, {0x5168f4e80ff5b54dull, 0x562761f92a7a23b4ull, 0x8113e0c0e47e4401ull, 0x03fa0984c931c9e3ull}},
    {{0x3aa372f0bd2d6997ull, 0x40c690c74709e90full, 0x764910f75b45f74bull, 0x04ba4cc166be8decull}, {0x48151f27b24b219cull, 0xcac5c59a5ce5ae7cull, 0x4b971e46c4ede85full, 0x0040301cf5c1751full}},
    {{0xd36ff12c49a58202ull, 0x2ca65048d53fb325ull, 0x6e44cca8f61a63bbull, 0x054302dcb0e6cc1cull}, {0x879dcc77e99c2426ull, 0xce98ad783c25561aull, 0xb348046268d8ae25ull, 0x01b77b3e37d13504ull}},
};
inline Pt pedersen_point(int k) {
    U256 x, y;
    memcpy(x.data(), PEDERSEN_POINTS[k][0], 32); memcpy(y.data(), PEDERSEN_POINTS[k][1], 32);
    return Pt{felt_from_canonical(x), felt_from_canonical(y)};
}
inline const std::vector<Pt> &constant_points() {          // gen_element_steps' constant_points for both inputs (512 entries)
    static const std::vector<Pt> pts = [] {                 // (made once, by whichever thread comes first: instances are traced in parallel)
        std::vector<Pt> all;
        for (int e = 0; e < 2; ++e) {
            std::vector<Pt> half;
            Pt acc = pedersen_point(1 + 2 * e);
            for (int i = 0; i < 248; ++i) { half.push_back(acc); acc = ec_double(acc); }
            acc = pedersen_point(2 + 2 * e);
            for (int i = 0; i < 4; ++i) { half.push_back(acc); acc = ec_double(acc); }
            for (int i = 0; i < 4; ++i) half.push_back(half[251]);
            all.insert(all.end(), half.begin(), half.end());
        }
        return all;
    }();
    return pts;
}
struct Step { Pt point; Felt suffix, slope; };
// gen_element_steps (builtins/src/pedersen/mod.rs:121-163) as the reference writes it: affine, one inversion per set bit
inline Pt element_steps_affine(const U256 &x, Pt point, int which, std::vector<Step> &out) {
    const std::vector<Pt> &cp = constant_points();
    for (unsigned i = 0; i < 256; ++i) {
        const U256 suffix = shr(x, i);
        Felt slope = felt_from_u64(0);
        Pt next = point;
        if (suffix[0] & 1) {
            const Pt &c = cp[256 * which + i];
            if (felt_eq(c.x, point.x)) {
                if (!felt_eq(c.y, point.y)) fail("point at infinity in a Pedersen partial sum");
                const Felt xx = felt_mul(c.x, c.x);
                slope = felt_mul(felt_add(felt_add(felt_add(xx, xx), xx), felt_from_u64(1)), felt_inv(felt_add(c.y, c.y)));
            } else {
                slope = felt_mul(felt_sub(point.y, c.y), felt_inv(felt_sub(point.x, c.x)));
            }
            const Felt x3 = felt_sub(felt_sub(felt_mul(slope, slope), point.x), c.x);        // the sum from the slope just made (chord or tangent)
            next = Pt{x3, felt_sub(felt_mul(slope, felt_sub(point.x, x3)), point.y)};
        }
        out.push_back(Step{point, felt_from_canonical(suffix), slope});
        point = next;
    }
    return point;
}
// The same cells with ONE inversion per element instead of one per set bit (a real instance cost 16 ms of host time that way - 500
// inversions; a 2^20-step run has room for 8192 / 32768 instances).  The partial sums are accumulated in Jacobian coordinates
// (mixed additions of the affine constant points); a sum's slope is R / Z3 of the addition that made the NEXT sum - the chord's slope
// (y2 - y1) / (x2 - x1) with x1 = X1 / Z1^2, y1 = Y1 / Z1^3 is R / (H Z1), and Z3 = Z1 H - so one batch inversion of the Z3's gives
// every affine sum and every slope.  The affine cells are field elements: whichever way they are computed they are the same
// (tests/test_layout_recursive.py holds the two paths together).  A sum that meets its constant point (a tangent; never for inputs
// that are not made for it) sends the element through the affine path.
inline Pt element_steps(const U256 &x, Pt point, int which, std::vector<Step> &out) {
    static const bool affine_only = getenv("SSH_TRACE_AFFINE_STEPS") != nullptr;
    if (affine_only) return element_steps_affine(x, point, which, out);
    const std::vector<Pt> &cp = constant_points();
    struct Sum { Felt X, Y, Z, R; };
    Sum acc[256];                                       // the sum after the k-th set bit, with the R of the addition that made it
    unsigned m = 0;
    Felt X1 = point.x, Y1 = point.y, Z1 = felt_from_u64(1);
    for (unsigned i = 0; i < 256; ++i) {
        if (!bit(x, i)) continue;
        const Pt &c = cp[256 * which + i];
        const Felt zz = felt_mul(Z1, Z1), u2 = felt_mul(c.x, zz), s2 = felt_mul(c.y, felt_mul(Z1, zz));
        const Felt h = felt_sub(u2, X1), r = felt_sub(s2, Y1);
        if (felt_is_zero(h)) {
            if (!felt_is_zero(r)) fail("point at infinity in a Pedersen partial sum");
            return element_steps_affine(x, point, which, out);
        }
        const Felt hh = felt_mul(h, h), hhh = felt_mul(h, hh), v = felt_mul(X1, hh);
        const Felt x3 = felt_sub(felt_sub(felt_mul(r, r), hhh), felt_add(v, v));
        const Felt y3 = felt_sub(felt_mul(r, felt_sub(v, x3)), felt_mul(Y1, hhh));
        const Felt z3 = felt_mul(Z1, h);
        acc[m++] = Sum{x3, y3, z3, r};
        X1 = x3; Y1 = y3; Z1 = z3;
    }
    // Montgomery's trick over the Z3's (none is zero: every H was not)
    Felt prefix[256], zinv[256];
    for (unsigned k = 0; k < m; ++k) prefix[k] = k ? felt_mul(prefix[k - 1], acc[k].Z) : acc[k].Z;
    if (m) {
        Felt run = felt_inv(prefix[m - 1]);
        for (unsigned k = m; k-- > 0;) { zinv[k] = k ? felt_mul(run, prefix[k - 1]) : run; if (k) run = felt_mul(run, acc[k].Z); }
    }
    const Felt zero = felt_from_u64(0);
    unsigned k = 0;                                     // set bits passed
    for (unsigned i = 0; i < 256; ++i) {
        Felt slope = zero;
        Pt next = point;
        if (bit(x, i)) {
            const Felt zi2 = felt_mul(zinv[k], zinv[k]);
            slope = felt_mul(acc[k].R, zinv[k]);
            next = Pt{felt_mul(acc[k].X, zi2), felt_mul(acc[k].Y, felt_mul(zi2, zinv[k]))};
            ++k;
        }
        out.push_back(Step{point, felt_from_canonical(shr(x, i)), slope});
        point = next;
    }
    return point;
}

inline uint64_t dilute(uint64_t v) {                       // bit i -> bit 4 i
    uint64_t out = 0;
    for (unsigned i = 0; i < 16; ++i) out |= ((v >> i) & 1ull) << (DILUTED_SPACING * i);
    return out;
}
inline uint32_t undilute(uint64_t v) {                     // DilutedCheckPool::push_diluted
    uint32_t out = 0;
    for (unsigned i = 0; i < DILUTED_N_BITS; ++i) out |= (uint32_t)((v >> (DILUTED_SPACING * i)) & 1ull) << i;
    if (dilute(out) != v) fail("a value is not in diluted form");
    return out;
}
inline void partition64(uint64_t v, uint64_t segs[4]) {    // Partition64::new
    for (int s = 0; s < 4; ++s) segs[s] = 0;
    for (unsigned b = 0; b < 16; ++b)
        for (unsigned s = 0; s < 4; ++s) segs[s] |= ((v >> (b * 4 + s)) & 1ull) << (b * 4);
}


// How many OpenMP threads the generators use: SSH_HOST_THREADS if set; else what OpenMP would take, but no more than twice the
// CPU time the cgroup grants (cpu.max / cpu.cfs_quota_us).  A container that SEES 256 hardware threads under a quota of 16 CPUs runs
// 256 runnable threads on 16: measured (profiles/r04_end_to_end.txt) 0.53-0.70 s per 2^20-step starknet trace with 128-256
// threads, 0.12-0.15 s with 16-32.
inline int host_threads() {
    if (const char *e = getenv("SSH_HOST_THREADS")) { const int t = atoi(e); if (t > 0) return t; }
    int want = omp_get_max_threads();
    double cpus = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                      // cgroup v2: "<quota|max> <period>"
        char q[32]; long period = 0;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && period > 0 && strcmp(q, "max") != 0) cpus = atof(q) / (double)period;
        fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {     // cgroup v1
        long quota = -1, period = 0;
        if (fscanf(g, "%ld", &quota) != 1) quota = -1;
        fclose(g);
        if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%ld", &period) != 1) period = 0; fclose(h); }
        if (quota > 0 && period > 0) cpus = (double)quota / (double)period;
    }
    if (cpus >= 1.0) want = std::min(want, std::max(1, (int)(2.0 * cpus + 0.5)));
    return std::max(1, want);
}
// for the span of a generator call: that many threads in its parallel regions, the caller's setting back afterwards
struct HostThreadsScope {
    int before;
    HostThreadsScope() : before(omp_get_max_threads()) { omp_set_num_threads(host_threads()); }
    ~HostThreadsScope() { omp_set_num_threads(before); }
};

// body(k) for k in [0, count) on the host's threads (dynamic: the items are builtin instances of uneven cost); the first exception a
// body throws is rethrown on the caller's thread (none may leave an OpenMP region)
template <class Body> void parallel_items(uint64_t count, const Body &body) {
    std::exception_ptr err;
#pragma omp parallel for schedule(dynamic, 1)
    for (uint64_t k = 0; k < count; ++k) {
        try { body(k); } catch (...) {
#pragma omp critical(ssh_trace_item_error)
            if (!err) err = std::current_exception();
        }
    }
    if (err) std::rethrow_exception(err);
}

// the addresses between the lowest and the highest accessed one that nothing accesses, ascending (the gap fillers of
// trace.rs:594-625 / 890-925): a byte map of the accessed addresses instead of a sort of all n / 2 accesses
inline std::vector<uint64_t> memory_gaps(const AddrArray &npc_addr, const std::vector<MemoryEntry> &public_memory, uint64_t max_gaps) {
    uint64_t top = 0, low = UINT64_MAX;
#pragma omp parallel for schedule(static) reduction(max : top) reduction(min : low)
    for (uint64_t k = 0; k < npc_addr.size(); ++k) { top = std::max(top, npc_addr[k]); low = std::min(low, npc_addr[k]); }
    for (auto &e : public_memory) { top = std::max<uint64_t>(top, e.address); low = std::min<uint64_t>(low, e.address); }
    std::vector<uint64_t> gaps;
    if (low == UINT64_MAX) return gaps;
    // The span holds at most one accessed address per access; whatever else is in it is a gap, and the caller has room for max_gaps
    // of them: a span beyond that fails HERE, with the caller's message, before anything is sized by an address out of a corrupt
    // memory.bin (the map below is then bounded by the trace, not by the file's largest address)
    if (top - low > (uint64_t)npc_addr.size() + public_memory.size() + max_gaps) fail("more memory gaps than cycles to hold them");
    std::vector<uint8_t> seen(top - low + 1, 0);
#pragma omp parallel for schedule(static)
    for (uint64_t k = 0; k < npc_addr.size(); ++k) {                              // (every writer writes the same byte; an idling run
        uint8_t *cell = &seen[npc_addr[k] - low];                                // hits the same few: look before writing, or 256 threads
        if (!__atomic_load_n(cell, __ATOMIC_RELAXED)) __atomic_store_n(cell, (uint8_t)1, __ATOMIC_RELAXED);   // pass one cache line around)
    }
    for (auto &e : public_memory) seen[e.address - low] = 1;
    for (uint64_t a = low + 1; a < top; ++a)
        if (!seen[a - low]) {
            gaps.push_back(a);
            if (gaps.size() > max_gaps) fail("more memory gaps than cycles to hold them");
        }
    return gaps;
}

// get_ordered_memory_accesses (layouts/src/utils.rs:112-152) without a sort.  The accesses - the pool's n / 2 (address, value)
// pairs and the public memory's `cells` (its entries, padded with (1, pad_value)) - ordered by address are RUNS: memory is
// single-valued and continuous, so the ordered column is (a, value(a)) repeated count(a) times for a = 1, 2, ...  Counting per
// address (atomic increments; the pool's zero and padding addresses, which most accesses carry, are counted per thread), a
// prefix sum and a parallel fill replace the sort of ~n / 2 forty-byte records (0.8 s of the recursive layout's 2.6 s at 2^20
// steps on 256 host threads), with the reference's checks kept: the pool's address-0 cells are exactly the public-memory
// cells, memory starts at address 1, has no gaps and one value per address.
inline void ordered_memory_into(Felt *mem_col, uint64_t n, const AddrArray &npc_addr, const Felt *npc, uint64_t cells,
                                const std::vector<MemoryEntry> &public_memory, const Felt &pad_value) {
    if (public_memory.size() > cells) fail("public memory does not fit");
    const uint64_t half = n / 2;
    uint64_t top = 1;
#pragma omp parallel for schedule(static) reduction(max : top)
    for (uint64_t k = 0; k < half; ++k) top = std::max(top, npc_addr[k]);
    for (auto &e : public_memory) top = std::max<uint64_t>(top, e.address);
    // continuous memory has an access per address 1 ... top, and there are n / 2 accesses in all (the checks below): an address
    // beyond that is the discontinuity itself - said before three arrays are sized by it (a corrupt memory.bin must not cost 20 B
    // per address up to its largest one)
    if (top > half) fail("memory is not continuous and single-valued: address " + std::to_string(top) + " with " + std::to_string(half) + " accesses");
    std::vector<uint32_t> count(top + 2, 0);
    std::vector<uint64_t> rep(top + 2, UINT64_MAX);           // the first access of an address: pool index, or half + public index
    // A run that idles in `jmp rel 0` (every padded one) reads the same handful of cells a million times: each thread counts through
    // a small direct-mapped table of its own (address -> count, first index) and only touches the shared arrays when an entry
    // is evicted or at the end - without it the threads of a 256-thread host serialise on a few cache lines (0.6 s at 2^20 steps).
    uint64_t zeros = 0, ones = 0;
#pragma omp parallel reduction(+ : zeros, ones)
    {
        constexpr int SLOTS = 256;
        uint64_t l_addr[SLOTS], l_rep[SLOTS];
        uint32_t l_cnt[SLOTS];
        for (int i = 0; i < SLOTS; ++i) { l_addr[i] = UINT64_MAX; l_cnt[i] = 0; l_rep[i] = UINT64_MAX; }
        auto flush = [&](int i) {
            if (l_addr[i] == UINT64_MAX) return;
            const uint64_t a = l_addr[i];
#pragma omp atomic
            count[a] += l_cnt[i];
            uint64_t seen = __atomic_load_n(&rep[a], __ATOMIC_RELAXED);
            while (l_rep[i] < seen && !__atomic_compare_exchange_n(&rep[a], &seen, l_rep[i], false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
        };
#pragma omp for schedule(static)
        for (uint64_t k = 0; k < half; ++k) {
            const uint64_t a = npc_addr[k];
            if (a == 0) { ++zeros; continue; }
            if (a == 1) { ++ones; continue; }
            const int i = (int)(a & (SLOTS - 1));
            if (l_addr[i] != a) { flush(i); l_addr[i] = a; l_cnt[i] = 0; l_rep[i] = k; }
            ++l_cnt[i];
        }
        for (int i = 0; i < SLOTS; ++i) flush(i);
    }
    if (zeros != cells) fail("the public-memory cells of the pool must be the only accesses of address 0");
    std::vector<Felt> pub_value(public_memory.size());
    for (size_t k = 0; k < public_memory.size(); ++k) {
        const uint64_t a = public_memory[k].address;
        pub_value[k] = felt_from_canonical(public_memory[k].value);
        if (a == 0) fail("the public-memory cells of the pool must be the only accesses of address 0");
        if (a == 1) { ++ones; continue; }
        ++count[a];
        rep[a] = std::min<uint64_t>(rep[a], half + k);
    }
    ones += cells - public_memory.size();                      // the padding entries (1, pad_value)
    if (!ones) fail("memory must start at address 1");
    if (ones > UINT32_MAX) fail("too many accesses of address 1");
    count[1] = (uint32_t)ones;
    auto value_of = [&](uint64_t idx) -> const Felt & { return idx < half ? npc[2 * idx + 1] : pub_value[idx - half]; };
    // one value per address (address 1: the padding value), no address skipped
    uint64_t first_bad = UINT64_MAX;
#pragma omp parallel for schedule(static) reduction(min : first_bad)
    for (uint64_t k = 0; k < half; ++k) {
        const uint64_t a = npc_addr[k];
        if (a == 0) continue;
        if (!felt_eq(npc[2 * k + 1], a == 1 ? pad_value : value_of(rep[a]))) first_bad = std::min(first_bad, a);
    }
    for (size_t k = 0; k < public_memory.size(); ++k) {
        const uint64_t a = public_memory[k].address;
        if (!felt_eq(pub_value[k], a == 1 ? pad_value : value_of(rep[a]))) first_bad = std::min<uint64_t>(first_bad, a);
    }
#pragma omp parallel for schedule(static) reduction(min : first_bad)
    for (uint64_t a = 1; a <= top; ++a) if (!count[a]) first_bad = std::min(first_bad, a - 1);
    if (first_bad != UINT64_MAX) fail("memory is not continuous and single-valued at address " + std::to_string(first_bad));
    // rows: address a fills [start(a), start(a) + count(a)); address 1's long run in parallel, the rest address by address
    std::vector<uint64_t> start(top + 2, 0);
    for (uint64_t a = 1; a <= top; ++a) start[a + 1] = start[a] + count[a];
    if (start[top + 1] != half) fail("the ordered memory does not fill its column");
    const Felt one_f = felt_from_u64(1);
#pragma omp parallel for schedule(static)
    for (uint64_t j = 0; j < ones; ++j) { mem_col[2 * j] = one_f; mem_col[2 * j + 1] = pad_value; }
#pragma omp parallel for schedule(dynamic, 4096)
    for (uint64_t a = 2; a <= top; ++a) {
        const Felt af = felt_from_u64(a), &v = value_of(rep[a]);
        for (uint64_t j = start[a]; j < start[a + 1]; ++j) { mem_col[2 * j] = af; mem_col[2 * j + 1] = v; }
    }
}

}  // namespace tracedetail
}  // namespace ssh
