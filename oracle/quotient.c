/* ORACLE — TEST INFRASTRUCTURE ONLY (see fp252.h).
 *
 * Row Q1 of SURVEY.md §8(a): pointwise evaluation of a lowered constraint
 * program over the LDE domain.  The reference evaluates its `Expr` DAG
 * (layouts/src/recursive/air.rs:61-1200) with ministark's default
 * AirConfig::eval_constraint (un-vendored).  This is a direct, one-point-at-a-
 * time interpreter of the program format documented in
 * include/sandstorm_hip.h (ss_air_program).
 */
#include "oracle.h"
#include "sandstorm_hip.h"
#include <stdlib.h>

/* block == 0: the whole domain (row0 = 0, nrows = N; trace cells wrap modulo N).  block != 0: the row-block form of
 * the sharded prover (ss_eval_quotient_rows): lde_cols[c][k] is LDE row row0 + k and carries the rows behind the block. */
static void eval_program(const ss_air_program *prog, const fp_t *tables, const fp_t *const *lde_cols,
                         unsigned log_n, unsigned log_blowup, fp_t offset, uint64_t row0, uint64_t nrows, int block, fp_t *out);

void or_eval_program_ex(const ss_air_program *prog, const fp_t *tables, const fp_t *const *lde_cols,
                        unsigned log_n, unsigned log_blowup, fp_t offset, fp_t *out) {
    eval_program(prog, tables, lde_cols, log_n, log_blowup, offset, 0, (uint64_t)1 << (log_n + log_blowup), 0, out);
}

void or_eval_program_rows(const ss_air_program *prog, const fp_t *tables, const fp_t *const *col_blocks,
                          unsigned log_n, unsigned log_blowup, fp_t offset, uint64_t row0, uint64_t nrows, fp_t *out) {
    eval_program(prog, tables, col_blocks, log_n, log_blowup, offset, row0, nrows, 1, out);
}

static void eval_program(const ss_air_program *prog, const fp_t *tables, const fp_t *const *lde_cols,
                         unsigned log_n, unsigned log_blowup, fp_t offset, uint64_t row0, uint64_t nrows, int block, fp_t *out) {
    unsigned log_N = log_n + log_blowup;
    size_t N = (size_t)nrows;
    const size_t wrap = block ? ~(size_t)0 : (((size_t)1 << log_N) - 1);
    fp_t wN = fp_root_of_unity(log_N);
    const fp_t *consts = (const fp_t *)prog->consts;
#pragma omp parallel if (N >= 256)
    {
        fp_t *slots = (fp_t *)calloc(prog->n_slots ? prog->n_slots : 1, sizeof(fp_t));
#pragma omp for schedule(static)
        for (size_t i = 0; i < N; ++i) {
            fp_t acc[4];
            memset(acc, 0, sizeof acc);
            fp_t x = fp_mul(offset, fp_pow_u64(wN, row0 + (uint64_t)i));
            for (uint32_t pc = 0; pc < prog->n_instr; ++pc) {
                uint32_t w0 = prog->code[2 * pc], w1 = prog->code[2 * pc + 1];
                unsigned op = w0 & 0xff, d = (w0 >> 8) & 0xf, kind = (w0 >> 12) & 0xf;
                fp_t src;
                memset(&src, 0, sizeof src);
                if (op != SS_OP_INV && op != SS_OP_ST && op != SS_OP_OUT) {
                    switch (kind) {
                    case SS_SRC_ACC: src = acc[w1 & 3]; break;
                    case SS_SRC_SLOT: src = slots[w1]; break;
                    case SS_SRC_CONST: src = consts[w1]; break;
                    case SS_SRC_TRACE: {
                        size_t col = w1 >> 24, ro = w1 & 0xffffff;
                        src = lde_cols[col][(i + (ro << log_blowup)) & wrap];
                    } break;
                    case SS_SRC_TABLE: {
                        uint32_t off = prog->table_desc[2 * w1], ll = prog->table_desc[2 * w1 + 1];
                        src = tables[off + ((row0 + i) & (((size_t)1 << ll) - 1))];
                    } break;
                    case SS_SRC_X: src = x; break;
                    default: break;
                    }
                }
                switch (op) {
                case SS_OP_MOV: acc[d] = src; break;
                case SS_OP_ADD: acc[d] = fp_add(acc[d], src); break;
                case SS_OP_SUB: acc[d] = fp_sub(acc[d], src); break;
                case SS_OP_RSUB: acc[d] = fp_sub(src, acc[d]); break;
                case SS_OP_MUL: acc[d] = fp_mul(acc[d], src); break;
                case SS_OP_INV: acc[d] = fp_inv(acc[d]); break;
                case SS_OP_ST: slots[w1] = acc[d]; break;
                case SS_OP_OUT: out[i] = acc[d]; break;
                default: break;
                }
            }
        }
        free(slots);
    }
}
