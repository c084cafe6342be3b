// The generated constraint kernels on the CPU (test infrastructure).  The body of a compiled kernel
// (sandstorm_amd/csrc/quotient_gen_<layout>.inc, written by tools/gen_quotient.py) is plain C++ over the operand macros
// of quotient_gen.h; here those macros read host arrays, a "workgroup" is a loop, and the body runs over a whole
// evaluation domain.  tests/test_quotient_gen_host.py compares the result with the oracle's constraint VM: the generator's
// decisions (lazy-form bounds and reductions, the rotating prefetch registers across the loop edge, slot traffic, the
// row-block form) are checked without a GPU and without a six-minute device build.
//
// usage: quotient_gen_host_test <input file> <output file>     (format: tests/test_quotient_gen_host.py)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../sandstorm_amd/csrc/fp252.h"
#include "../../sandstorm_amd/csrc/fl252.h"

using namespace ss;

struct HostArgs {
    std::vector<std::vector<Fp>> cols;
    std::vector<Fp> tables, tables_scaled;   // the multiplier-only tables times 2^24 (QG_SCALED_H: csrc/quotient_gen_<layout>_scaled.inc)
    std::vector<uint32_t> tdesc;          // per table: first element, index mask
    std::vector<Fl> consts, consts_r280, consts_up, consts_upn;
    std::vector<Fp> out;
    Fp offset, w;
    uint64_t npoints;
    uint32_t row0, trace_mask, log_blowup;
};

#define QG_TRACE_RAW(col, off, idx) a.cols[col][((idx) + ((off) << lb)) & maskN]
#define QG_TABLE_RAW(t, idx) a.tables[a.tdesc[2 * (t)] + (((idx) + row0) & a.tdesc[2 * (t) + 1])]
#define QG_TABLE_SCALED_RAW(t, idx) a.tables_scaled[a.tdesc[2 * (t)] + (((idx) + row0) & a.tdesc[2 * (t) + 1])]
#define QG_CONST(k) a.consts[k]
#define QG_CONST_R280(k) a.consts_r280[k]
#define QG_CONST_R280_UP(k) a.consts_up[k]
#define QG_CONST_R280_UPN(k) a.consts_upn[k]
#define QG_SLOT_STORE(k, v) slots[k] = fl_pack(v)
#define QG_SLOT(k) fl_from_fp(slots[k])
#define QG_OUT(v) *(qg_live ? &a.out[i] : &sink) = fl_to_fp(v)
#define QG_OUT_ACC(v) *(qg_live ? &a.out[i] : &sink) = fl_to_fp(fl_add(fl_weak_reduce(v), fl_from_fp(a.out[i])))
#define QG_SYNC
#define QG_PIN_LOADS
#define QG_FENCE
typedef FlWide QgWide;
#define qg_dot_zero fl_wide_zero
#define qg_dot_mad fl_wide_mad
#define qg_dot_reduce fl_wide_reduce
static inline void qg_wide_tail(QgWide &w, const Fl &l) {
    for (int j = 0; j < 9; ++j) w.c[9 + j] += (u64)l.l[j] << 4;
}
#define QG_POINT_LOOP_BEGIN                                        \
    const uint64_t qg_iters = (N + lanes - 1) / lanes;             \
    for (uint64_t qg_it = 0; qg_it < qg_iters; ++qg_it) {          \
        const uint64_t qg_raw = lane + qg_it * lanes;              \
        const bool qg_live = qg_raw < N;                           \
        const uint64_t i = qg_live ? qg_raw : N - 1;               \
        i32 = (uint32_t)i;                                         \
        const uint32_t inext = (uint32_t)(qg_raw + lanes < N ? qg_raw + lanes : N - 1);
#define QG_POINT_LOOP_END                                          \
        x = fl_mul(x, wstep);                                      \
    }

// one "lane" of a grid of `lanes` lanes: exactly the device kernel's per-lane code.  A compiled program is one or more parts
// (csrc/quotient_gen_<layout>_p<j>.inc) run one after the other over the whole domain, like the device's launches: part 0 stores,
// the others add.  QG_PARTS_H (written by tests/test_quotient_gen_host.py) defines one function per part with QG_LANE_PRELUDE in
// front of the included body, and the table PARTS.
#define QG_LANE_PRELUDE                                                              \
    const uint64_t N = a.npoints;                                                    \
    const uint32_t lb = a.log_blowup, maskN = a.trace_mask, row0 = a.row0;           \
    Fp slots[64], sink;                                                              \
    Fl x = fl_from_fp(fp_mul(a.offset, fp_pow_u64(a.w, lane)));                      \
    const Fl wstep = fl_from_fp(fp_pow_u64(a.w, lanes));
typedef void (*part_fn)(HostArgs &, uint64_t, uint64_t);
#include QG_SCALED_H
#include QG_PARTS_H

template <class T>
static void rd(FILE *f, T *p, size_t n) { if (fread(p, sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } }

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    uint64_t hdr[10];          // ncols, rows per column, ntables_felts, ntables, nconsts, npoints, row0, trace_mask, log_blowup, lanes
    rd(f, hdr, 10);
    HostArgs a;
    a.cols.resize(hdr[0]);
    for (auto &c : a.cols) { c.resize(hdr[1]); rd(f, c.data(), c.size()); }
    a.tables.resize(hdr[2]); rd(f, a.tables.data(), a.tables.size());
    a.tdesc.resize(2 * hdr[3]); rd(f, a.tdesc.data(), a.tdesc.size());
    std::vector<Fp> consts(hdr[4]); rd(f, consts.data(), consts.size());
    rd(f, &a.offset, 1); rd(f, &a.w, 1);
    fclose(f);
    Fp two24 = fp_zero(); two24.v[0] = 1u << 24;
    for (auto &c : consts) {
        const Fp up = fp_mul(c, fp_to_mont(two24));
        a.consts.push_back(fl_from_fp(c)); a.consts_r280.push_back(fl_to_r280(c));
        a.consts_up.push_back(fl_to_r280(up)); a.consts_upn.push_back(fl_to_r280(fp_neg(up)));
    }
    {                                            // what csrc/capi.hip eval_quotient_compiled does per launch
        Fp f24 = fp_zero(); f24.v[0] = 1u << 24;
        const Fp f = fp_to_mont(f24);
        for (uint32_t j = 0; j < QG_N_SCALED; ++j) {
            const uint32_t t = QG_SCALED_TABLES[j], first = a.tdesc[2 * t], len = a.tdesc[2 * t + 1] + 1u;
            a.tdesc.push_back((uint32_t)a.tables_scaled.size());
            a.tdesc.push_back(len - 1u);
            for (uint32_t i = 0; i < len; ++i) a.tables_scaled.push_back(fp_mul(a.tables[first + i], f));
        }
        if (QG_N_TABLES * 2 + QG_N_SCALED * 2 != a.tdesc.size()) { fprintf(stderr, "table count\n"); return 2; }
    }
    a.npoints = hdr[5]; a.row0 = (uint32_t)hdr[6]; a.trace_mask = (uint32_t)hdr[7]; a.log_blowup = (uint32_t)hdr[8];
    a.out.assign(a.npoints, fp_zero());
    a.offset = fp_mul(a.offset, fp_pow_u64(a.w, a.row0));
    const uint64_t lanes = hdr[9];
    for (part_fn part : PARTS) {
#pragma omp parallel for schedule(dynamic, 16)
        for (uint64_t lane = 0; lane < lanes; ++lane) part(a, lane, lanes);
    }
    f = fopen(argv[2], "wb");
    fwrite(a.out.data(), sizeof(Fp), a.out.size(), f);
    fclose(f);
    return 0;
}
