/* ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, gcc) of Sandstorm's proving hot path
 * (SURVEY.md §8a).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path never does.
 *
 * PARITY STATUS: the field, NTT convention, Keccak/Blake2s, Pedersen and the
 * two Fiat-Shamir coins are pinned by the reference's own known-answer tests
 * (tests/golden/, see oracle/README.md).  Everything ministark decides
 * internally and no reference test pins — LDE coset offset, row order at
 * commit, Merkle node indexing / depth numbering, FRI reshape + fold, DEEP
 * term order — is "parity unpinned": those
 * functions follow the mathematical definition with the assumed convention
 * exposed as a parameter.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stddef.h>
#include "fp252.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { OR_HASH_KECCAK = 0, OR_HASH_KECCAK_M20 = 1, OR_HASH_BLAKE2S = 2, OR_HASH_BLAKE2S_M20 = 3 };
/* Merkle tree configurations of src/claims.rs:12-33 */
enum {
    OR_TREE_KECCAK = 0,      /* LeafVariantMerkleTree<Keccak256HashFn>            */
    OR_TREE_KECCAK_M20 = 1,  /* LeafVariantMerkleTree<MaskedKeccak256HashFn<20>>  */
    OR_TREE_FRIENDLY = 2     /* FriendlyMerkleTree<N, PedersenHashFn>             */
};
enum { OR_LEAF_DIGEST = 0, OR_LEAF_FELT = 1 };

void or_init(void);

/* ---- ntt.c */
void or_bitrev_permute(fp_t *a, unsigned log_n);
void or_ntt_forward(fp_t *a, unsigned log_n, const fp_t *offset);
void or_ntt_inverse(fp_t *a, unsigned log_n, const fp_t *offset);
void or_lde(const fp_t *in, unsigned log_n, unsigned log_blowup, const fp_t *offset,
            fp_t *evals_out, fp_t *coeffs_out);
fp_t or_poly_eval(const fp_t *coeffs, size_t n, fp_t x);

/* ---- hash.c */
void or_keccak256(const uint8_t *msg, size_t len, uint8_t out[32]);
void or_blake2s256(const uint8_t *msg, size_t len, uint8_t out[32]);
void or_sha256(const uint8_t *msg, size_t len, uint8_t out[32]);       /* FIPS 180-4 (the 64-bit field's Sha256HashFn: cli/src/main.rs:105,119) */
void or_apply_mask(int kind, uint8_t d[32]);
void or_hash_bytes(int kind, const uint8_t *msg, size_t len, uint8_t out[32]);
void or_hash_elements(int kind, const fp_t *e, size_t n, uint8_t out[32]);
void or_hash_merge(int kind, const uint8_t a[32], const uint8_t b[32], uint8_t out[32]);
void or_hash_rows(int kind, const fp_t *const *cols, size_t ncols, size_t nrows, uint8_t *out);

fp_t or_mulmod_chain(fp_t x, fp_t y, uint64_t iters);       /* fp252.c: a dependent chain of products (timing) */

/* ---- pedersen.c */
void or_pedersen_init(void);
fp_t or_pedersen_hash(fp_t a, fp_t b);
fp_t or_pedersen_hash_bitwise(fp_t a, fp_t b);      /* the definition the table form is held to */
/* PedersenHashFn::hash_elements chain (crypto/src/hash/pedersen.rs:65-76) */
fp_t or_pedersen_hash_elements(const fp_t *e, size_t n);
/* affine doubling chain 2^i * P_k, i < count (periodic-column KAT helper) */
void or_pedersen_doublings(int k, size_t count, fp_t *xs, fp_t *ys);

/* ---- merkle.c
 * nodes: heap layout, 2*n entries of 32 bytes; nodes[1] = root, children of
 * k are 2k and 2k+1, leaves (digests, or raw Montgomery felts for
 * OR_LEAF_FELT) conceptually at n..2n-1 (for OR_LEAF_FELT the leaf slots
 * hold the 32 Montgomery-BE bytes).  tags: one byte per node, 0 = HighLevel
 * (Pedersen felt, 32-byte BE canonical), 1 = LowLevel (Blake2s), as
 * MixedMerkleDigest (crypto/src/merkle/mixed.rs:34-71); unused (0) for the
 * Keccak trees. */
void or_merkle_build(int tree, unsigned n_friendly_layers, int leaf_kind, const uint8_t *leaves,
                     size_t n, uint8_t *nodes, uint8_t *tags);

/* ---- fri.c */
/* One FRI layer fold: evals of length 2^log_len on
 * offset*<w>, natural order; row j = {evals[j + k*(len/fold)]}; out[j] = value
 * at alpha of the degree<fold interpolant over {offset*w^j * w_fold^k}. */
void or_fri_fold_ex(const fp_t *evals, unsigned log_len, unsigned fold, fp_t alpha, fp_t offset,
                    unsigned flags, fp_t *out);
void or_fri_fold(const fp_t *evals, unsigned log_len, unsigned fold, fp_t alpha, fp_t offset,
                 fp_t *out);

/* ---- deep.c */
/* DEEP composition, evaluated pointwise on the LDE domain:
 * out[i] = sum_j coeff_j * (T_{col_j}(x_i) - ood_j) / (x_i - z*w_n^{off_j})
 *        + sum_k coeffc_k * (H_k(x_i) - oodc_k) / (x_i - z^ncomp)
 * x_i = offset * w_N^i.  lde columns are natural order, length 2^log_N. */
void or_deep_compose(const fp_t *const *trace_lde, const fp_t *const *comp_lde, unsigned log_n,
                     unsigned log_blowup, fp_t offset, const uint32_t *mask_col,
                     const uint32_t *mask_off, size_t nmask, const fp_t *ood_trace,
                     const fp_t *coeff_trace, size_t ncomp, const fp_t *ood_comp,
                     const fp_t *coeff_comp, fp_t z, fp_t *out);

/* ---- ext.c: Trace::build_extension_columns (layouts/src/recursive/trace.rs:699-814, starknet/trace.rs:997-1100) */
/* out[i*out_stride + out_off] = prod_{k<=i} term(num, k) * batch_inversion(prod_{k<=i} term(den, k)), i < count;
 * term = z - (alpha * item[v] + item[a]) or, with v < 0, z - item[a]; item k = column + k*stride. */
void or_permutation_product(const fp_t *num, uint64_t num_stride, uint64_t num_a, int64_t num_v,
                            const fp_t *den, uint64_t den_stride, uint64_t den_a, int64_t den_v,
                            uint64_t count, fp_t z, fp_t alpha, fp_t *out, uint64_t out_stride, uint64_t out_off);
/* out[out_off] = 1; out[i*out_stride + out_off] = acc_i = acc_{i-1} (1 + z u_i) + alpha u_i^2, u_i = x_i - x_{i-1} */
void or_diluted_aggregate(const fp_t *ordered, uint64_t stride, uint64_t off, uint64_t count, fp_t z, fp_t alpha,
                          fp_t *out, uint64_t out_stride, uint64_t out_off);

/* ---- coin.c: the two Fiat-Shamir coins (crypto/src/public_coin/{solidity,cairo}.rs) */
typedef struct { int kind; /* 0 = Solidity/Keccak, 1 = Cairo/Blake2s */ uint8_t digest[32]; uint64_t counter; } or_coin;
void or_coin_new(or_coin *c, int kind, const uint8_t digest[32]);
void or_coin_reseed_bytes(or_coin *c, const uint8_t *bytes, size_t len);
void or_coin_reseed_felts(or_coin *c, const fp_t *v, size_t n);       /* reseed_with_field_elements */
void or_coin_reseed_felt_vector(or_coin *c, const fp_t *v, size_t n); /* reseed_with_field_element_vector */
void or_coin_reseed_int(or_coin *c, uint64_t v);
fp_t or_coin_draw(or_coin *c);
void or_coin_draw_queries(or_coin *c, size_t max_n, uint64_t domain_size, uint64_t *out /* max_n, unsorted */);
uint64_t or_coin_grind(const or_coin *c, unsigned bits); /* smallest valid nonce >= 1 */
int or_coin_verify_pow(const or_coin *c, unsigned bits, uint64_t nonce);

/* ---- quotient.c: constraint-program interpreter; program format in
 * include/sandstorm_hip.h (ss_air_program; `tables` is the host copy of
 * prog->d_tables) */
struct ss_air_program;
void or_eval_program_ex(const struct ss_air_program *prog, const fp_t *tables,
                        const fp_t *const *lde_cols, unsigned log_n, unsigned log_blowup,
                        fp_t offset, fp_t *out);

#ifdef __cplusplus
}
#endif

/* whole-pipeline helpers (oracle/cpu_context.py): the DEEP composition with one inversion per point, over a row block */
void or_deep_compose_rows(const fp_t *const *trace_lde, const fp_t *const *comp_lde, unsigned log_n,
                          unsigned log_blowup, fp_t offset, const uint32_t *mask_col,
                          const uint32_t *mask_off, size_t nmask, const fp_t *ood_trace,
                          const fp_t *coeff_trace, size_t ncomp, const fp_t *ood_comp,
                          const fp_t *coeff_comp, fp_t z, uint64_t row0, uint64_t nrows, uint64_t stride, fp_t *out);
void or_eval_program_rows(const struct ss_air_program *prog, const fp_t *tables, const fp_t *const *col_blocks,
                          unsigned log_n, unsigned log_blowup, fp_t offset, uint64_t row0, uint64_t nrows, fp_t *out);
void or_ood_eval(const fp_t *const *coeffs, unsigned log_n, const uint32_t *mask_col, const uint32_t *mask_off,
                 size_t nmask, fp_t z, fp_t *out);
void or_inverse_table(unsigned log_N, fp_t offset, fp_t c, fp_t *out);

#endif
