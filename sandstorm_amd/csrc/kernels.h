// kernels.h — internal launch interface between the C ABI (capi.hip) and the
// gfx950 kernels.  Not part of the public boundary (include/sandstorm_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fp252.h"

namespace ss {

static constexpr int MAX_COLS = 16;  // columns per launch (kernarg-resident pointer table)

struct ColPtrs {
    const void *src[MAX_COLS];
    void *dst[MAX_COLS];
};
struct ConstColPtrs {
    const void *p[MAX_COLS];
};

// ---- ntt.hip
int ntt_log_tile_max();
// bytes per twiddle-plan entry (R252 limb planes, ntt.hip); a plan of a size-2^log_n transform has 2^log_n - 1 entries
static constexpr size_t NTT_PLAN_ENTRY_BYTES = 36;
// the three stage networks of ntt_pass_kernel: forward (bit-reversed in, natural out), inverse with Gentleman-Sande butterflies
// (natural in, bit-reversed out; any coset), inverse over the subgroup with Cooley-Tukey butterflies (needs a PLAN_BITREV plan)
enum { NTT_MODE_DIT = 0, NTT_MODE_DIF = 1, NTT_MODE_CTI = 2 };
// a windowed top pass (ntt.hip PassParams): this rank's 2^log_len elements q >= first of every row
struct NttWindow { uint32_t first, log_len; };
hipError_t launch_ntt_pass(hipStream_t st, int mode, const ColPtrs &cols, uint32_t ncols, const Fp *tw,
                           uint32_t log_n, uint32_t s0, uint32_t r, uint32_t log_tile, uint32_t u_first,
                           uint32_t log_expand, uint32_t scale_pow2, bool final_pass, bool cti_trivial = true,
                           const NttWindow *win = nullptr, uint64_t tw_entries = 0);      // tw_entries: of a window's plan (0: 2^log_n - 1)
hipError_t launch_twiddles(hipStream_t st, Fp *tw, const Fp *pow_lo, const Fp *pow_hi, const Fp *hpow,
                           uint32_t log_n, bool h_is_one, bool bitrev_levels, const NttWindow *win = nullptr, uint32_t win_stages = 0);
// entries of the plan of a windowed top pass of `stages` stages (DIT / DIF; ntt.hip PassParams.tw_entries)
static inline uint64_t ntt_window_plan_entries(const NttWindow &win, uint32_t stages) { return (((uint64_t)1 << stages) - 1ull) << win.log_len; }
hipError_t launch_bitrev(hipStream_t st, Fp *a, uint32_t log_n);
hipError_t launch_mul_bench(hipStream_t st, const Fp *a, const Fp *b, Fp *out, uint64_t n, uint32_t reps);
hipError_t ntt_set_func_attributes();

// ---- hash.hip
hipError_t launch_hash_rows(hipStream_t st, int kind, const ConstColPtrs &cols, uint32_t ncols,
                            uint64_t nrows, uint32_t brev_bits, uint8_t *digests);
hipError_t launch_bitrev_copy(hipStream_t st, const Fp *src, uint32_t log_n, Fp *dst);
// one Merkle level: out[k] = H(in[2k] || in[2k+1]) for k < count (64-byte messages)
hipError_t launch_hash_pairs(hipStream_t st, int kind, const uint8_t *in, uint64_t count, uint8_t *out);
// leaf level of single-column trees: out[k] = H::hash_elements([felt[2k], felt[2k+1]])
hipError_t launch_hash_felt_pairs(hipStream_t st, int kind, const Fp *felts, uint64_t count, uint8_t *out);
// felt leaves -> Montgomery big-endian bytes (leaf slots of the node array)
hipError_t launch_felts_to_be(hipStream_t st, const Fp *felts, uint64_t count, uint8_t *out);
hipError_t launch_pow_prefix(hipStream_t st, int coin_kind, const uint8_t digest[32], uint32_t bits,
                             uint64_t *d_prefix);
hipError_t launch_pow_grind(hipStream_t st, int coin_kind, const uint64_t *d_prefix, uint32_t bits,
                            uint64_t start, uint64_t count, unsigned long long *d_best);
hipError_t launch_gather32(hipStream_t st, const uint8_t *in, const uint64_t *d_idx, uint64_t n, uint8_t *out);
hipError_t launch_gather32_cols(hipStream_t st, const void *const *d_cols, uint32_t ncols, uint32_t c0, uint32_t row_cols, const uint64_t *d_idx,
                                uint64_t nidx, uint8_t *out);            // ncols <= MAX_COLS; out[(q * row_cols + c0 + c) * 32 ..]
hipError_t launch_gather8(hipStream_t st, const uint8_t *in, const uint64_t *d_idx, uint64_t n, uint8_t *out);

// ---- pedersen.hip
struct PedersenTables;  // device-resident windowed tables, built once per context
hipError_t pedersen_tables_create(hipStream_t st, PedersenTables **out);
void pedersen_tables_destroy(PedersenTables *t);
void pedersen_tables_trim();         // free the tables no context uses (ss_ctx_trim)
// Every launcher takes `tmp`: PEDERSEN_TMP_FELTS_PER_HASH felts of device scratch per hash (Jacobian X and Z,
// prefix products of the batched inversion, the chained digest of hash_elements).
static constexpr uint64_t PEDERSEN_TMP_FELTS_PER_HASH = 4;
// out[i] = pedersen(a[i], b[i]), Montgomery felts
hipError_t launch_pedersen_felts(hipStream_t st, const PedersenTables *t, const Fp *a, const Fp *b,
                                 uint64_t n, Fp *out, Fp *tmp);
// Merkle level on 32-byte big-endian digests: out[k] = BE(pedersen(int(in[2k]) mod p, int(in[2k+1]) mod p))
hipError_t launch_pedersen_pairs(hipStream_t st, const PedersenTables *t, const uint8_t *in, uint64_t count,
                                 uint8_t *out, Fp *tmp);
// single-column leaf level: out[k] = BE(PedersenHashFn::hash_elements([f[2k], f[2k+1]]))
hipError_t launch_pedersen_felt_pairs(hipStream_t st, const PedersenTables *t, const Fp *felts,
                                      uint64_t count, uint8_t *out, Fp *tmp);

// host-side hash for the Fiat-Shamir coin (Montgomery felts)
Fp pedersen_hash_host(const Fp &a, const Fp &b);

// ---- fri.hip
// rows row0 .. row0 + count of the layer, entry k of row row0 + i at evals[i + k * count] (the whole layer: 0, len / fold)
hipError_t launch_fri_fold(hipStream_t st, const Fp *evals, uint32_t log_len, uint32_t log_fold,
                           const Fp &alpha, const Fp &offset_inv, const Fp &w_inv, const Fp *fold_tw_inv,
                           uint32_t flags, Fp *out, uint64_t row0, uint64_t count, const Fp *pow_tab, uint32_t lo_bits);
// pow_tab of launch_fri_fold: w_inv^k, k < 2^lo_bits, then w_inv^(k << lo_bits), k < 2^hi_bits (null: the power per lane)
hipError_t launch_fri_pow_table(hipStream_t st, Fp *tab, const Fp &w_inv, uint32_t lo_bits, uint32_t hi_bits);

// ---- deep.hip
hipError_t launch_poly_reduce(hipStream_t st, const void *const *in, void *const *out, uint32_t ncols,
                              uint64_t in_len, uint32_t levels, const Fp *mult);
hipError_t launch_batch_inverse(hipStream_t st, Fp *D, uint32_t log_N, const Fp &offset, const Fp &w,
                                const Fp &w_inv, const Fp &z, bool r280);
hipError_t launch_deep(hipStream_t st, const void *const *trace, uint32_t ntrace, const void *const *comp,
                       uint32_t ncomp, const Fp *D, const Fp *Dc, const uint32_t *tap_shift, const Fp *tap_coef,
                       const uint32_t *col_desc, uint32_t ncoldesc, const Fp *comp_coef,
                       const Fp &comp_k, uint64_t count, uint32_t d_bias, uint32_t d_mask, uint32_t log_stride, Fp *out);
// the mask's large columns as rational functions (deep.hip): V[m] <- 1 / V[m] (tmp: len felts), and
// out[m] += (sum_k trace_k[m << log_stride] * A_k[m] - AK[m]) * Binv[m]   (A_k, Binv in R280 form; ncols <= DEEP_RATIONAL_MAX_COLS)
static constexpr uint32_t DEEP_RATIONAL_MAX_COLS = 8;
hipError_t launch_batch_inverse_values(hipStream_t st, Fp *V, Fp *tmp, uint64_t len);
hipError_t launch_deep_rational(hipStream_t st, const void *const *trace, const Fp *const *A, uint32_t ncols, const Fp *AK, const Fp *Binv,
                                uint64_t count, uint32_t log_stride, Fp *out);
static constexpr uint32_t BATCH_INVERSE_RANGE_LOG_CHUNK = 5;
hipError_t launch_batch_inverse_range(hipStream_t st, Fp *D, uint64_t len, const Fp &x0, const Fp &w, const Fp &w_inv,
                                      const Fp &z, bool r280);
// out-of-domain evaluation at few points (deep.hip): blocks of 2^S coefficients transformed in registers, then folded per point
struct OodBlockArgs {
    const Fp *coeffs;        // the column: n bit-reversed coefficients
    Fp *out;                 // [popcount(res_mask)][blocks]
    uint64_t blocks;         // R = n >> S
    uint32_t res_mask;       // residues (k mod 2^S) to keep, in ascending order
    uint32_t tw[15][9];      // twiddles of stages 0 .. S-1 in R280 limb form: stage u at (2^u - 1) + j, j < 2^u
};
struct OodFoldArray { const Fp *in; uint32_t first_point, npoints; };           // an input array and the points that read it
struct OodFoldPoint { Fp *out; uint32_t coef[16][9]; uint32_t pad[2]; };           // coef[t]: R280 limbs of the weight of element t
hipError_t launch_ood_blocks(hipStream_t st, int S, const OodBlockArgs &a);
hipError_t launch_ood_fold(hipStream_t st, const OodFoldArray *d_arrays, uint32_t narrays, uint32_t max_points, const OodFoldPoint *d_points,
                           uint64_t in_len, uint32_t levels, bool canonical);
hipError_t launch_gather_cells(hipStream_t st, const void *const *cols, uint32_t ncols, const uint32_t *col,
                               const uint64_t *idx, uint32_t n, Fp *out);

// ---- ext.hip (extension-trace scans; PermOperand and the scratch sizes are in ext_scan.h)
struct PermOperand;
hipError_t launch_permutation_product(hipStream_t st, const PermOperand &num, const PermOperand &den, uint64_t count,
                                      const Fp &z, const Fp &alpha, Fp *out, uint64_t out_stride, uint64_t out_off, Fp *scratch);
hipError_t launch_diluted_aggregate(hipStream_t st, const Fp *x, uint64_t stride, uint64_t off, uint64_t count, const Fp &z,
                                    const Fp &alpha, Fp *out, uint64_t out_stride, uint64_t out_off, Fp *scratch);
// the scans of ONE column over the row blocks of several devices (ABI 12): a block's own scan, then the blocks before it folded in
hipError_t launch_scale_strided(hipStream_t st, Fp *data, uint64_t stride, uint64_t off, uint64_t count, const Fp &factor);
hipError_t launch_diluted_aggregate_maps(hipStream_t st, const Fp *x, uint64_t stride, uint64_t off, uint64_t count, bool starts_column,
                                         const Fp &z, const Fp &alpha, Fp *mc, Fp *scratch);
hipError_t launch_affine_apply(hipStream_t st, const Fp *mc, uint64_t count, const Fp &start, Fp *out, uint64_t out_stride, uint64_t out_off);

// ---- trace.hip (the base trace on the device; ss_trace_* of the C ABI)
// where the CPU's cells sit in a cycle's 16 rows (= ss_trace_layout): the memory pool's 8 (address, value) pairs, the range-check
// column's and the auxiliary column's 16 cells
struct TraceLayout { uint8_t npc_pair[8], rc_cell[16], aux_cell[16]; };
enum { TRACE_NPC_PAD = 0, TRACE_NPC_PUBLIC = 1, TRACE_NPC_PC = 2, TRACE_NPC_OP0 = 3, TRACE_NPC_DST = 4, TRACE_NPC_OP1 = 5 };
enum { TRACE_RC_FILL = 0, TRACE_RC_ZERO = 1, TRACE_RC_OFF_DST = 2, TRACE_RC_OFF_OP0 = 3, TRACE_RC_OFF_OP1 = 4 };
enum { TRACE_AUX_ZERO = 0, TRACE_AUX_AP = 1, TRACE_AUX_FP = 2, TRACE_AUX_TMP0 = 3, TRACE_AUX_TMP1 = 4, TRACE_AUX_MUL = 5, TRACE_AUX_RES = 6 };
struct TraceTileEntry { uint32_t col, off, kind, arg; };          // = ss_trace_cell
enum { TRACE_TILE_VALUE = 0, TRACE_TILE_ADDRESS = 1 };
struct TraceRcPlan {                                               // = ss_trace_rc_plan
    uint64_t n_slots, n_given, slot_rows, addr_begin, n_padding, pad0;
    uint32_t part_stride, part_off, pair_off, rc_lo, rc_hi, ordered_step, ordered_off, unused_off;
};
struct TraceMemoryArgs {
    uint64_t n;                       // trace rows
    Fp *npc, *memory;                 // the memory pool's column, the ordered column
    uint32_t *d_pool_addr;            // n / 2: the pool's addresses as integers
    const uint32_t *d_public_addr;    // the public memory's entries: addresses, Montgomery values
    const Fp *d_public_value;
    uint32_t n_public;
    uint64_t public_cells;            // the pool's address-0 pairs (n / PUBLIC_MEMORY_STEP)
    Fp pad_value;                     // the value at address 1
    uint32_t unused_off;              // row offset of a cycle's unused pool pair (Npc::UnusedAddr)
    uint32_t *d_status;
};
// the status words of a generation (u32, zeroed by the caller before the first kernel)
enum { TRACE_ST_ERRORS = 0, TRACE_ST_WHERE = 1, TRACE_ST_ZEROS = 2, TRACE_ST_ONES = 3, TRACE_ST_TOP = 4, TRACE_ST_NLOW = 5, TRACE_ST_GAPS = 6, TRACE_ST_WORDS = 16 };
enum { TRACE_ERR_MISSING_CELL = 1, TRACE_ERR_NOT_INSTRUCTION = 2, TRACE_ERR_BAD_OP1_SOURCE = 4, TRACE_ERR_BAD_RES_LOGIC = 8, TRACE_ERR_NOT_AN_ADDRESS = 16,
       TRACE_ERR_ADDRESS_RANGE = 32, TRACE_ERR_PUBLIC_ZERO = 64, TRACE_ERR_PUBLIC_CELLS = 128, TRACE_ERR_NO_ONES = 256, TRACE_ERR_NOT_SINGLE_VALUED = 512,
       TRACE_ERR_NOT_CONTINUOUS = 1024, TRACE_ERR_TOO_MANY_GAPS = 2048, TRACE_ERR_FILL = 4096 };
hipError_t launch_trace_memory_image(hipStream_t st, const uint64_t *d_records, uint64_t n_records, uint64_t *d_image, uint64_t cells);
hipError_t launch_trace_cpu(hipStream_t st, const TraceLayout &L, const uint64_t *d_states, uint64_t num_cycles, const uint64_t *d_image, uint64_t cells,
                            const Fp &pad_value, uint64_t rc_fill, Fp *flags, Fp *npc, Fp *rc, Fp *aux, uint32_t *d_pool_addr, uint32_t *d_status);
hipError_t launch_trace_tile(hipStream_t st, const ColPtrs &cols, uint32_t ncols, const TraceTileEntry *d_entries, uint32_t n_entries, const Fp *d_values,
                             uint32_t n_templates, const uint32_t *d_tmpl_of_block, uint64_t nblocks, uint64_t step, uint64_t addr_begin, uint64_t addr_mult,
                             uint32_t *d_pool_addr);
hipError_t launch_trace_rc_builtin(hipStream_t st, const TraceRcPlan &p, const uint64_t *d_given, const uint16_t *d_padding, Fp *rc, Fp *npc, uint32_t *d_pool_addr);
hipError_t launch_trace_rc_pool(hipStream_t st, const TraceRcPlan &p, const uint32_t *d_first, const uint16_t *d_padding, uint64_t num_cycles, Fp *rc);
hipError_t launch_trace_runs(hipStream_t st, Fp *col, uint64_t stride, uint64_t off, uint64_t slots, const uint32_t *d_first, uint32_t n_values, uint32_t lo,
                             bool diluted);
hipError_t launch_trace_patch(hipStream_t st, Fp *col, uint64_t col_rows, const uint64_t *d_rows, const uint64_t *d_values, uint64_t count);
uint64_t trace_memory_scratch_words(uint64_t half);
hipError_t launch_trace_ordered_memory(hipStream_t st, const TraceMemoryArgs &m, uint32_t *scratch);

// ---- goldilocks.hip (the 64-bit field variant)
uint64_t gl_pow_host(uint64_t a, uint64_t e);
uint64_t gl_root_of_unity_host(uint32_t log_n);
uint64_t gl_inv_host(uint64_t a);
uint32_t gl_log_tile_max();
uint32_t gl_log_min_run();
hipError_t gl_set_func_attributes();
hipError_t launch_gl_ntt_pass(hipStream_t st, bool dif, const void *const *src, void *const *dst, uint32_t ncols, const uint64_t *tw,
                              uint32_t log_n, uint32_t s0, uint32_t r, uint32_t log_tile, uint32_t u_first, uint32_t log_expand,
                              uint64_t scale);
hipError_t launch_gl_twiddles(hipStream_t st, uint64_t *tw, const uint64_t *pow_lo, const uint64_t *pow_hi, const uint64_t *hpow, uint32_t log_n);
hipError_t launch_gl_bitrev_copy(hipStream_t st, const uint64_t *src, uint64_t *dst, uint32_t log_n);
hipError_t launch_gl3_fri_fold(hipStream_t st, const uint64_t *evals, uint32_t log_len, uint32_t fold, const uint64_t alpha[3], uint64_t offset,
                               bool unnormalised, uint64_t *out);
hipError_t launch_gl3_inverse_table(hipStream_t st, uint64_t *D, uint64_t len, uint64_t x0, uint64_t w, const uint64_t z[3]);
hipError_t launch_gl3_deep(hipStream_t st, const uint64_t *const *trace, uint32_t ntrace, const uint64_t *const *comp, uint32_t ncomp,
                           const uint64_t *D, const uint64_t *Dc, const uint32_t *tap_shift, const uint64_t *tap_coef, const uint32_t *col_desc,
                           uint32_t ncoldesc, const uint64_t *comp_coef, const uint64_t comp_k[3], uint64_t count, uint32_t log_stride,
                           uint64_t *out0, uint64_t *out1, uint64_t *out2);
hipError_t launch_gl3_interleave(hipStream_t st, const uint64_t *c0, const uint64_t *c1, const uint64_t *c2, uint64_t len, uint64_t *out);
hipError_t launch_gl3_zpow_bitrev(hipStream_t st, uint64_t *zp0, uint64_t *zp1, uint64_t *zp2, uint32_t log_n, const uint64_t z[3]);
hipError_t launch_gl3_scale_columns(hipStream_t st, const uint64_t *coef, const uint64_t *zp0, const uint64_t *zp1, const uint64_t *zp2, uint64_t n,
                                    uint64_t *o0, uint64_t *o1, uint64_t *o2);
hipError_t launch_hash_rows_u64(hipStream_t st, int kind, const ConstColPtrs &segs, uint32_t nseg, uint32_t seg_len, uint64_t nrows, uint8_t *digests);
hipError_t launch_gather_rows_u64(hipStream_t st, const ConstColPtrs &segs, uint32_t nseg, uint32_t seg_len, const uint64_t *d_idx, uint32_t nidx,
                                  uint64_t *d_out);
hipError_t launch_gl3_running_product(hipStream_t st, const uint64_t *na, const uint64_t *nv, const uint64_t *da, const uint64_t *dv, uint64_t stride,
                                      uint64_t count, const uint64_t z[3], const uint64_t alpha[3], uint64_t *scratch, uint64_t *o0, uint64_t *o1,
                                      uint64_t *o2, uint64_t out_stride, uint64_t out_offset, uint64_t *d_last);
uint32_t gl3_dot_blocks(uint64_t n);
hipError_t launch_gl3_dot(hipStream_t st, const uint64_t *coef, const uint64_t *zp0, const uint64_t *zp1, const uint64_t *zp2, uint64_t n, uint64_t *partial);
bool gl3_compiled_matches(const uint32_t *code, uint32_t n_instr, const uint64_t *consts3, uint32_t n_consts, uint32_t n_tables);
hipError_t launch_gl3_plain(hipStream_t st, const uint64_t *d_consts, const uint64_t *d_tables, const uint32_t *d_tdesc, const uint64_t *const *cols,
                            uint32_t ncols, uint64_t *d_out, uint64_t offset, uint64_t w, uint32_t log_blowup, uint64_t N);
uint32_t gl3_vm_lanes(uint64_t N);
hipError_t launch_gl3_vm(hipStream_t st, const uint32_t *d_code, uint32_t n_instr, const uint64_t *d_consts, const uint64_t *d_tables,
                         const uint32_t *d_tdesc, const uint64_t *const *cols, uint32_t ncols, uint64_t *d_slots, uint64_t *d_out, uint64_t offset,
                         uint64_t w, uint32_t log_blowup, uint64_t N);
hipError_t launch_gl3_gather(hipStream_t st, const uint64_t *c0, const uint64_t *c1, const uint64_t *c2, const uint64_t *idx, uint32_t count, uint64_t *out);

// ---- quotient.hip
// what ss_eval_quotient knows and the device program needs resolved (device addresses, sizes)
struct VmResolve {
    const void *cols[MAX_COLS];
    const void *consts, *consts_r280, *tables, *slots;    // consts_r280: the constants times 2^24 (R280 form)
    const uint32_t *table_desc;      // host copy: [n_tables][2] = (offset in felts, log2 length)
    uint64_t lanes;
    uint32_t log_blowup;
    uint32_t trace_mask;             // trace cells at (point + shift) & trace_mask: N - 1 (whole columns) or ~0 (row block)
    uint32_t row0;                   // global row of the first point (tables are indexed by the global row)
};
// caller's 2-word program -> resolved device program: (n_instr + 1) entries of 8 words (quotient.hip)
void quotient_build_device_code(const uint32_t *code, uint32_t n_instr, const VmResolve &r, uint32_t *dev);
hipError_t launch_quotient_vm(hipStream_t st, const uint32_t *d_code, uint32_t n_entries, Fp *d_slots, uint64_t lanes,
                              const Fp &offset, const Fp &w, const Fp &wstep, uint64_t npoints, uint32_t xcd_split, Fp *out);

}  // namespace ss
