/* sandstorm_hip.h — C ABI of libsandstorm_hip.so
 *
 * The MI355X (gfx950) proving backend that slots behind Sandstorm's
 * `sandstorm-cli prove` / ministark prover API.  The reference has no FFI for
 * this path: its seams are Rust traits.  Each entry point below
 * is what a `hip` cargo feature of ministark would bind in place of its Metal
 * `gpu` feature, and cites the reference interface it replaces.  See
 * INTEGRATION.md for the Rust-side `extern "C"` block.
 *
 * Conventions
 *  - Field element: 4 x u64 little-endian limbs, Montgomery form, R = 2^256,
 *    p = 2^251 + 17*2^192 + 1 — the in-memory image of the reference's `Fp`
 *    (crypto/src/utils.rs:8-22).  "felt" below always means this 32-byte image.
 *  - A matrix is column-major: one contiguous device array per column
 *    (ministark `Matrix<F>` = Vec<GpuVec<F>>; layouts/src/recursive/trace.rs:652-660).
 *    `cols` arguments are HOST arrays of DEVICE pointers.
 *  - Digest: 32 raw bytes.  MixedMerkleDigest (crypto/src/merkle/mixed.rs:34-71)
 *    additionally carries a tag byte: 0 = HighLevel (Pedersen felt as 32-byte
 *    big-endian canonical, hash/pedersen.rs:23-28), 1 = LowLevel (Blake2s).
 *  - Every pointer named d_* is device memory (from ss_dev_alloc, hipMalloc or a
 *    torch tensor's data_ptr); every other pointer is host memory owned by the
 *    caller for the duration of the call.
 *  - Work is enqueued on the context's HIP stream; calls that return host data
 *    synchronise that stream, the others do not (use ss_ctx_sync).
 *  - Errors: integer status, never an exception/abort across the ABI; the text
 *    is in ss_last_error() (thread-local).  The reference panics on any error
 *    (cli/src/main.rs:201 `unwrap`), so a Rust shim `expect()`s on non-zero.
 *  - One ctx per host thread and per GPU; no global mutable state.
 */
#ifndef SANDSTORM_HIP_H
#define SANDSTORM_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Every entry point validates what it is handed before anything is launched - sizes, directions / orders / hash, tree, leaf and coin
 * kinds, NULL tables and NULL entries of column tables, mask cells and program operands against the column / constant / slot /
 * table counts of the same call, query indices against the tree - and answers SS_ERR_INVALID / SS_ERR_UNSUPPORTED with a message in
 * ss_last_error(); a call it cannot serve never reaches a kernel (tests/hipemu/extra_bad_arguments.py sweeps all of them).  What
 * it cannot check is the SIZE of device memory behind a pointer: buffers must hold what the call's sizes say. */
typedef int ss_status;
enum {
    SS_OK = 0,
    SS_ERR_INVALID = 1,     /* bad argument */
    SS_ERR_HIP = 2,         /* HIP runtime error, see ss_last_error */
    SS_ERR_NO_DEVICE = 3,   /* no gfx950 device / HIP extension unusable */
    SS_ERR_UNSUPPORTED = 4
};

typedef struct ss_ctx ss_ctx;

/* NTT direction / element order */
enum { SS_NTT_FORWARD = 0, SS_NTT_INVERSE = 1 };
enum { SS_ORDER_NATURAL = 0, SS_ORDER_BITREV = 1 };

/* HashFn implementations of crypto/src/hash/{keccak,blake2s}.rs */
enum {
    SS_HASH_KECCAK = 0,       /* Keccak256HashFn           keccak.rs:13-59  */
    SS_HASH_KECCAK_M20 = 1,   /* MaskedKeccak256HashFn<20> keccak.rs:61-98  */
    SS_HASH_BLAKE2S = 2,      /* Blake2sHashFn             blake2s.rs:10-62 */
    SS_HASH_BLAKE2S_M20 = 3,  /* MaskedBlake2sHashFn<20>   blake2s.rs:64-100 */
    SS_HASH_SHA256 = 4        /* ministark's Sha256HashFn (cli/src/main.rs:105,119: the 64-bit field's claim); FIPS 180-4; ss_hash_rows_gl64 only */
};
/* Merkle tree types chosen by src/claims.rs:12-33 */
enum {
    SS_TREE_KECCAK = 0,       /* LeafVariantMerkleTree<Keccak256HashFn>           */
    SS_TREE_KECCAK_M20 = 1,   /* LeafVariantMerkleTree<MaskedKeccak256HashFn<20>> */
    SS_TREE_FRIENDLY = 2,     /* FriendlyMerkleTree<N, PedersenHashFn>            */
    SS_TREE_BLAKE2S = 3,      /* MatrixMerkleTree over Blake2sHashFn (blake2s.rs:10-62): the 64-bit field's trees here */
    SS_TREE_SHA256 = 4        /* MatrixMerkleTreeImpl<Sha256HashFn> (cli/src/main.rs:119): node = SHA-256(left || right), digests as leaves */
};
enum { SS_LEAF_DIGEST = 0, SS_LEAF_FELT = 1 };
enum { SS_COIN_SOLIDITY = 0, SS_COIN_CAIRO = 1 };

const char *ss_last_error(void);
/* ABI version of this header; bump on any signature change.  ss_abi_version() of the loaded
 * library must equal SS_ABI_VERSION of the header the caller was built against. */
#define SS_ABI_VERSION 12u
uint32_t ss_abi_version(void);

/* ---- context & memory (replaces ministark-gpu's Metal planner/GpuAllocator;
 *      call sites src/lib.rs:27-28, layouts/src/recursive/trace.rs:115-120) */
ss_status ss_ctx_create(int device, ss_ctx **out);
void ss_ctx_destroy(ss_ctx *ctx);
ss_status ss_ctx_set_stream(ss_ctx *ctx, void *hip_stream); /* NULL = ctx-owned stream */
ss_status ss_ctx_sync(ss_ctx *ctx);       /* waits for the ctx stream and for the copy stream of ss_upload_async */
/* ss_dev_alloc/ss_dev_free are pooled (the role of GpuAllocator / PageAlignedAllocator,
 * src/lib.rs:27-28): a freed block is kept for reuse by later allocations of a similar size
 * on the same context, so steady-state proving makes no hipMalloc/hipFree calls.
 * ss_ctx_trim returns the cached blocks to the driver, and with them the Pedersen window tables that no context of the process
 * uses any more.  (Memory a process holds besides its buffers: twiddle plans, 36 B per point and (size, direction, offset);
 * for FriendlyMerkleTree claims ONE fixed-base window table per device and process - 23.6 GB at the default 24-bit windows,
 * narrower windows (6.4 / 1.75 / 0.47 / 0.13 GB) when the device's free memory does not leave 4 GiB beside it or SS_PED_WINDOW
 * says so; built at the first friendly tree in ~0.1 s.) */
ss_status ss_dev_alloc(ss_ctx *ctx, size_t bytes, void **d_out);
ss_status ss_dev_free(ss_ctx *ctx, void *d_ptr);
ss_status ss_ctx_trim(ss_ctx *ctx);
ss_status ss_upload(ss_ctx *ctx, void *d_dst, const void *src, size_t bytes);
ss_status ss_download(ss_ctx *ctx, void *dst, const void *d_src, size_t bytes);
/* Upload without the wait: the copy is enqueued on a copy stream of the context's own and returns at once (src: pinned host memory
 * that stays untouched until the copy is done - the `GpuAllocator` seam, layouts/src/recursive/trace.rs:115-120); *ticket names it.
 * ss_wait_upload(ctx, ticket) orders everything enqueued on the context's stream AFTERWARDS behind that copy (a stream wait: the
 * host does not block); a ticket is waited for once.  ss_upload_async may be called from another host thread than the one that
 * drives the context - the trace generator's, which uploads a column the moment no section will write it again, while the prover
 * already extends the columns that have arrived (`Stark::prove` starts from the witness: src/lib.rs:94-100). */
ss_status ss_upload_async(ss_ctx *ctx, void *d_dst, const void *src, size_t bytes, uint64_t *ticket);
ss_status ss_wait_upload(ss_ctx *ctx, uint64_t ticket);
/* zero-fill on the ctx stream (`Vec::resize(trace_len, Fp::ZERO)`, layouts/src/recursive/trace.rs:741-748) */
ss_status ss_dev_zero(ss_ctx *ctx, void *d_ptr, size_t bytes);

/* ---- data movement of the sharded driver (ABI 7; DESIGN.md section 6: nothing in the reference to replace - its
 * parallelism is rayon loops in one address space, crypto/src/merkle/utils.rs:30-32).  All on the ctx stream, no host sync. */
/* device -> device */
ss_status ss_dev_copy(ss_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);
/* `rows` runs of `width` bytes: row r from d_src + r * src_pitch to d_dst + r * dst_pitch (the stride-R comb of a leaf block) */
ss_status ss_dev_copy_2d(ss_ctx *ctx, void *d_dst, size_t dst_pitch, const void *d_src, size_t src_pitch, size_t width, size_t rows);
/* 32-byte elements: d_dst[bitrev(i)] = d_src[i], i < 2^log_n (a single column's leaves in the commitment order); not in place */
ss_status ss_bitrev_permute32(ss_ctx *ctx, const void *d_src, uint32_t log_n, void *d_dst);

/* One communicator per rank over RCCL (loaded with dlopen at the first call: a process that never shards does not need the
 * library).  Rank 0 makes the 128-byte id (ss_comm_unique_id) and hands it to the other ranks by whatever launched them; every
 * rank then calls ss_comm_create on its own context.  The exchanges run on the context's stream. */
typedef struct ss_comm ss_comm;
ss_status ss_comm_unique_id(uint8_t id_out[128]);
ss_status ss_comm_create(ss_ctx *ctx, const uint8_t id[128], uint32_t rank, uint32_t world, ss_comm **out);
void ss_comm_destroy(ss_comm *comm);
/* One exchange step that EVERY rank enters: message i of this rank goes from d_send[i] (send_bytes[i] bytes) to rank
 * send_peer[i]; message j arrives from rank recv_peer[j] in d_recv[j].  The messages of one ordered pair of ranks are matched
 * in list order (messages to the own rank are device copies); either list may be empty.  ONE grouped ncclSend / ncclRecv
 * batch - on xGMI each pair of GPUs has its own link, so the R (R - 1) transfers of a re-shard run concurrently, no ring. */
ss_status ss_comm_exchange(ss_comm *comm, uint32_t nsend, const uint32_t *send_peer, const void *const *d_send, const uint64_t *send_bytes,
                           uint32_t nrecv, const uint32_t *recv_peer, void *const *d_recv, const uint64_t *recv_bytes);
/* `bytes` bytes of every rank, in rank order, into d_recv (world * bytes) */
ss_status ss_comm_all_gather(ss_comm *comm, const void *d_send, uint64_t bytes, void *d_recv);

/* ---- N1/N2: ministark Matrix::interpolate / Matrix::evaluate (un-vendored;
 *      call sites src/lib.rs:17-26; convention pinned by
 *      builtins/src/pedersen/periodic.rs:1183-1209).
 * In-place NTT of `ncols` columns of 2^log_n felts over offset*<w>, w =
 * 3^((p-1)/2^log_n).  FORWARD: coefficients -> evaluations; INVERSE:
 * evaluations -> coefficients (includes 1/n and offset^-i).  `offset` is a
 * Montgomery felt (NULL = 1).  in_order/out_order select natural or
 * bit-reversed indexing of the array on entry/exit. */
ss_status ss_ntt_fp252(ss_ctx *ctx, uint64_t *const *d_cols, uint32_t ncols, uint32_t log_n,
                       int direction, const uint64_t offset[4], int in_order, int out_order);

/* Low-degree extension of `ncols` columns (pipeline steps 3+4 / 8 / 10 of
 * SURVEY §3.1): interpolate over <w_n>, evaluate over offset*<w_{n*2^log_blowup}>.
 * d_evals[c]: 2^(log_n+log_blowup) felts, natural order.
 * d_coeffs (may be NULL) / d_coeffs[c]: 2^log_n felts, the interpolant's
 * coefficients in BIT-REVERSED index order (kept for OOD/DEEP). d_in may alias
 * d_coeffs column-wise, never d_evals. */
ss_status ss_lde_fp252(ss_ctx *ctx, const uint64_t *const *d_in, uint32_t ncols, uint32_t log_n,
                       uint32_t log_blowup, const uint64_t offset[4], uint64_t *const *d_evals,
                       uint64_t *const *d_coeffs);

/* Evaluate-only half of the LDE (pipeline step 10: the composition columns, whose
 * coefficients come from one inverse NTT): d_coeffs[c] holds 2^log_n coefficients
 * in BIT-REVERSED order; d_evals[c] receives the 2^(log_n+log_blowup) evaluations
 * over offset*<w>, natural order.  d_coeffs is not modified. */
ss_status ss_evaluate_fp252(ss_ctx *ctx, const uint64_t *const *d_coeffs, uint32_t ncols, uint32_t log_n,
                            uint32_t log_blowup, const uint64_t offset[4], uint64_t *const *d_evals);

/* ---- (e) ONE transform spread over R = 2^log_ranks GPUs (nothing in the reference to match: its
 * Matrix::interpolate / evaluate run in one address space, src/lib.rs:17-26).  A rank holds the contiguous block
 * [rank n/R, (rank+1) n/R) of the array.  The network's log2(n/R) stages that pair elements less than n/R apart never leave
 * a block (part LOCAL: on the rank's block, a buffer of n/R felts); its log_ranks stages that pair element x of block b with
 * element x of another block run after ONE equal-split all-to-all (ss_comm_exchange: chunk m of every block to rank m) has
 * given rank m the x-range [m n/R^2, (m+1) n/R^2) of EVERY block, block after block in a buffer of n/R felts (part CROSS).
 *   INVERSE (natural in, bit-reversed out, 1/n and offset^-j included):  CROSS on the exchanged layout, exchange back, LOCAL
 *   FORWARD (bit-reversed in, natural out):  LOCAL on the block, exchange, CROSS; the caller exchanges back if it wants blocks.
 * log_n = log2 of the WHOLE transform, n >= R^2.  log_expand (FORWARD LOCAL only): the block's input is the 2^-log_expand
 * sub-sampled coefficient block (n/R >> log_expand felts in d_cols[c]; d_out[c]: n/R felts) - the zero-padded half of an
 * LDE is not materialised.  d_out = NULL: in place.  Everything is bit-identical to ss_ntt_fp252 / ss_evaluate_fp252 on the
 * gathered array (tests/test_gpu_parity.py::test_ntt_spread_over_ranks). */
enum { SS_NTT_PART_LOCAL = 0, SS_NTT_PART_CROSS = 1 };
ss_status ss_ntt_shard_fp252(ss_ctx *ctx, uint64_t *const *d_cols, uint32_t ncols, uint32_t log_n, uint32_t log_ranks,
                             uint32_t rank, int direction, const uint64_t offset[4], int part, uint32_t log_expand,
                             uint64_t *const *d_out);

/* ---- H1: crypto/src/merkle/utils.rs:19-46 hash_rows::<H>.
 * d_digests[r] = H::hash_elements(row r) for r < nrows; rows are read in
 * natural index order from the column arrays. */
ss_status ss_hash_rows(ss_ctx *ctx, int hash_kind, const uint64_t *const *d_cols, uint32_t ncols,
                       uint64_t nrows, uint8_t *d_digests);
/* The same with the leaf order of the reference's proofs: row_order = SS_ORDER_BITREV makes digest i the hash of
 * matrix row bitrev(i) (index i of a committed vector is the point offset * w^bitrev(i): pinned by
 * tests/golden/make_proof_golden.py).  The matrix stays in natural order - nothing is permuted in memory. */
ss_status ss_hash_rows_ex(ss_ctx *ctx, int hash_kind, const uint64_t *const *d_cols, uint32_t ncols,
                       uint64_t nrows, int row_order, uint8_t *d_digests);

/* ---- H2/H3/H4: MatrixMerkleTree::from_matrix tree build
 *      (crypto/src/merkle/mod.rs:110-123, 289-304; node rules mixed.rs:106-155,
 *      mod.rs:419-437; builder ministark MerkleTreeImpl::new, un-vendored).
 * leaves: n digests (SS_LEAF_DIGEST) or n felts (SS_LEAF_FELT, single-column
 * matrices).  d_nodes: 2n x 32 bytes, heap layout: node k has children 2k,
 * 2k+1; root at index 1; leaf i at n+i (felt leaves stored as Montgomery-BE
 * bytes).  d_tags (SS_TREE_FRIENDLY only, may be NULL): 2n tag bytes.
 * root_out[0..32) digest, root_out[32] tag. */
ss_status ss_merkle_build(ss_ctx *ctx, int tree_kind, uint32_t n_friendly_layers, int leaf_kind,
                          const void *d_leaves, uint64_t n, uint8_t *d_nodes, uint8_t *d_tags,
                          uint8_t root_out[33]);
/* leaf_order = SS_ORDER_BITREV: for SS_LEAF_FELT the leaf slot i holds element bitrev(i) of the column (digest
 * leaves are taken as given - order them with ss_hash_rows_ex). */
ss_status ss_merkle_build_ex(ss_ctx *ctx, int tree_kind, uint32_t n_friendly_layers, int leaf_kind,
                          const void *d_leaves, uint64_t n, int leaf_order, uint8_t *d_nodes, uint8_t *d_tags,
                          uint8_t root_out[33]);
/* MerkleTree::prove: authentication paths for `nidx` leaf indices.
 * out: nidx * log2(n) sibling digests (leaf level first), 32 bytes each;
 * out_tags (may be NULL) the matching tag bytes. */
ss_status ss_merkle_open(ss_ctx *ctx, const uint8_t *d_nodes, const uint8_t *d_tags, uint64_t n,
                         const uint64_t *idx, uint32_t nidx, uint8_t *out, uint8_t *out_tags);
/* gather full rows `idx` of a column-major matrix to the host (query phase) */
ss_status ss_gather_rows(ss_ctx *ctx, const uint64_t *const *d_cols, uint32_t ncols,
                         const uint64_t *idx, uint32_t nidx, uint64_t *out /* nidx*ncols felts */);
/* The whole query phase in one round trip (ABI 10): every job gathers entries `idx` of `ncols` device arrays of `entry_bytes`-byte entries
 * (32: field elements or digests - the opened rows of a matrix, a tree's leaf digests, or authentication paths with idx = the sibling
 * node numbers over the node array as one "column"; 1: the tag bytes of a FriendlyMerkleTree's nodes, ncols = 1) into host memory
 * `out`, nidx * ncols entries, row after row - what ss_gather_rows / ss_merkle_open return, but one index upload, one download and one
 * synchronisation for all jobs (ministark's `Queries::new` - un-vendored, Cargo.lock:894-896 - with `MerkleTree::prove`,
 * crypto/src/merkle/mod.rs:258-304, over the three trace trees and every FRI layer: 27 calls of ~60 us each before).  Jobs with
 * nidx = 0 are skipped. */
typedef struct ss_gather_job {
    const void *const *d_cols;
    uint32_t ncols;
    uint32_t entry_bytes;      /* 32 or 1 */
    const uint64_t *idx;       /* host */
    uint32_t nidx;
    void *out;                 /* host: nidx * ncols * entry_bytes */
} ss_gather_job;
ss_status ss_gather_batch(ss_ctx *ctx, const ss_gather_job *jobs, uint32_t njobs);

/* ---- Q1: AirConfig::eval_constraint over the LDE domain (ministark default
 *      body; DAG built by layouts/src/{recursive,starknet}/air.rs).
 * The Rust side lowers its `Expr` DAG once per (layout, n) into this program
 * for a 4-accumulator register machine (two-address, one instruction = 2 x u32):
 *   word0: [7:0] opcode, [11:8] dst accumulator, [15:12] operand kind
 *   word1: operand payload
 * The program is executed once per LDE point i (x_i = offset * w_N^i). */
enum {
    SS_OP_MOV = 0,   /* acc[d] = src            */
    SS_OP_ADD = 1,   /* acc[d] = acc[d] + src   */
    SS_OP_SUB = 2,   /* acc[d] = acc[d] - src   */
    SS_OP_RSUB = 3,  /* acc[d] = src - acc[d]   */
    SS_OP_MUL = 4,   /* acc[d] = acc[d] * src   */
    SS_OP_INV = 5,   /* acc[d] = 1 / acc[d]  (0 -> 0), operand ignored */
    SS_OP_ST = 6,    /* slot[payload] = acc[d]  */
    SS_OP_OUT = 7    /* out[i] = acc[d]         */
};
enum {
    SS_SRC_ACC = 0,    /* payload = accumulator index                                  */
    SS_SRC_SLOT = 1,   /* payload = slot index (per-point scratch)                     */
    SS_SRC_CONST = 2,  /* payload = index into consts (challenges, hints, alpha^k ...) */
    SS_SRC_TRACE = 3,  /* payload = col << 24 | row_offset: lde[col][(i + (row_offset << log_blowup)) mod N] */
    SS_SRC_TABLE = 4,  /* payload = table index: tables[desc.offset + (i mod 2^desc.log_len)]
                          (periodic columns and X^(n/k) zerofier inverses are periodic in i) */
    SS_SRC_X = 5       /* x_i */
};
#define SS_INSTR(op, dst, kind, payload) ((uint32_t)(op) | ((uint32_t)(dst) << 8) | ((uint32_t)(kind) << 12)), ((uint32_t)(payload))
typedef struct ss_air_program {
    const uint32_t *code;        /* 2 * n_instr words (host) */
    uint32_t n_instr;
    const uint64_t *consts;      /* n_consts felts (host) */
    uint32_t n_consts;
    const uint64_t *d_tables;    /* device: concatenated tables of felts */
    const uint32_t *table_desc;  /* host: 2 * n_tables words: {offset in felts, log_len} */
    uint32_t n_tables;
    uint32_t n_slots;            /* scratch slots used by ST / SS_SRC_SLOT */
} ss_air_program;
/* Table of a single-point zerofier inverse over the whole LDE domain:
 * d_out[i] = 1 / (offset * w_N^i - c), i < 2^log_N (0 where the denominator is 0),
 * by Montgomery batch inversion.  The AIRs' first/last-row boundary terms
 * 1/(X - c) (e.g. layouts/src/recursive/air.rs:295-298) become TABLE operands of
 * length 2^log_N instead of a per-point field inversion. */
ss_status ss_inverse_table(ss_ctx *ctx, uint32_t log_N, const uint64_t offset[4], const uint64_t c[4],
                           uint64_t *d_out);
ss_status ss_eval_quotient(ss_ctx *ctx, const ss_air_program *prog, const uint64_t *const *d_lde_cols,
                           uint32_t ncols, uint32_t log_n, uint32_t log_blowup,
                           const uint64_t offset[4], uint64_t *d_out);
/* Row-block form, for one proof sharded over several GPUs (the reference has only rayon loops over
 * rows to replace, crypto/src/merkle/utils.rs:30-32).  The evaluation domain is cut into contiguous row blocks; a
 * rank evaluates points row0 .. row0 + nrows from column BLOCKS: d_col_blocks[c][k] = LDE row (row0 + k) mod N of
 * column c, k < block_rows, where block_rows >= nrows + (largest row offset of the program << log_blowup) - the rows
 * behind the block that its constraints reach (wrap-around included) travel with it.  Tables stay whole and are
 * indexed by the global row.  d_out[k] = what ss_eval_quotient writes at row0 + k, bit for bit. */
ss_status ss_eval_quotient_rows(ss_ctx *ctx, const ss_air_program *prog, const uint64_t *const *d_col_blocks,
                                uint32_t ncols, uint32_t log_n, uint32_t log_blowup, const uint64_t offset[4],
                                uint64_t row0, uint64_t nrows, uint64_t block_rows, uint64_t *d_out);

/* ---- D1: out-of-domain evaluations + DEEP composition (ministark
 *      DeepPolyComposer, un-vendored; coefficient rule src/lib.rs:102-116).
 * ss_ood_eval: out[j] = T_{mask_col[j]}(z * w_n^{mask_off[j]}) from the
 * bit-reversed coefficient columns kept by ss_lde_fp252. */
ss_status ss_ood_eval(ss_ctx *ctx, const uint64_t *const *d_coeffs, uint32_t ncols, uint32_t log_n,
                      const uint32_t *mask_col, const uint32_t *mask_off, uint32_t nmask,
                      const uint64_t z[4], uint64_t *out /* nmask felts, host */);
/* ss_poly_eval: out[c] = P_c(x) for bit-reversed coefficient columns. */
ss_status ss_poly_eval(ss_ctx *ctx, const uint64_t *const *d_coeffs, uint32_t ncols, uint32_t log_n,
                       const uint64_t x[4], uint64_t *out /* ncols felts, host */);
/* Optional, ahead of ss_deep_compose (ABI 12): the composer's denominators 1 / (x - z w_n^o) and 1 / (x - z^ncomp) over
 * offset * <w_n> depend on the out-of-domain point alone.  Queued here - the call does not wait - they are built while
 * the host hashes the out-of-domain values into the coin (the Cairo coin: one Pedersen hash per value, crypto/src/
 * public_coin/cairo.rs reseed_with_field_elements); the next ss_deep_compose with the same (ncomp, log_n, offset, z)
 * uses them, any other makes its own. */
ss_status ss_deep_prepare(ss_ctx *ctx, uint32_t ncomp, uint32_t log_n, const uint64_t offset[4], const uint64_t z[4]);
/* d_out[i] = sum_j coeff_trace[j] (T_{col_j}(x_i) - ood_trace[j]) / (x_i - z w_n^{off_j})
 *          + sum_k coeff_comp[k]  (H_k(x_i)      - ood_comp[k])  / (x_i - z^ncomp)
 * for every x_i = offset * w_N^i of the LDE domain, natural order. */
ss_status ss_deep_compose(ss_ctx *ctx, const uint64_t *const *d_trace_lde, uint32_t ntrace_cols,
                          const uint64_t *const *d_comp_lde, uint32_t ncomp, uint32_t log_n,
                          uint32_t log_blowup, const uint64_t offset[4], const uint32_t *mask_col,
                          const uint32_t *mask_off, uint32_t nmask, const uint64_t *ood_trace,
                          const uint64_t *coeff_trace, const uint64_t *ood_comp,
                          const uint64_t *coeff_comp, const uint64_t z[4], uint64_t *d_out);
/* The two halves of ss_deep_compose, for one proof sharded over several GPUs.  The DEEP polynomial has degree < n, so
 * it is composed on the sub-coset offset * <w_n> (every blowup-th LDE row) and then interpolated and re-expanded:
 * ss_deep_compose_rows writes its values at the sub-coset points m0 <= m < m0 + count from column blocks with
 * d_*_blocks[c][k] = LDE row (m0 << log_blowup) + k (no rows behind the block are needed: DEEP reads one row per
 * point); ss_deep_extend takes all n sub-coset values (natural order, gathered from the ranks; overwritten) to the N
 * evaluations.  ss_deep_compose(...) == rows(m0 = 0, count = n) followed by extend, bit for bit. */
ss_status ss_deep_compose_rows(ss_ctx *ctx, const uint64_t *const *d_trace_blocks, uint32_t ntrace_cols,
                               const uint64_t *const *d_comp_blocks, uint32_t ncomp, uint32_t log_n,
                               uint32_t log_blowup, const uint64_t offset[4], const uint32_t *mask_col,
                               const uint32_t *mask_off, uint32_t nmask, const uint64_t *ood_trace,
                               const uint64_t *coeff_trace, const uint64_t *ood_comp,
                               const uint64_t *coeff_comp, const uint64_t z[4], uint64_t m0, uint64_t count,
                               uint64_t *d_out_subcoset);
ss_status ss_deep_extend(ss_ctx *ctx, uint64_t *d_subcoset, uint32_t log_n, uint32_t log_blowup,
                         const uint64_t offset[4], uint64_t *d_out);

/* ---- X4: the 64-bit field variant, p = 2^64 - 2^32 + 1 ("Goldilocks") with Fq3 = Fp[X]/(X^3 - 2): the pair the
 *      reference instantiates for its experimental claim (cli/src/main.rs:103-133:
 *      ministark_gpu::fields::p18446744069414584321::ark::{Fp, Fq3}; BASELINE.json configs[4]).  The same conventions as
 *      the 252-bit entry points (w = 7^((p-1)/n), evaluation at offset * w^k in natural order).  Elements are 8-byte
 *      values < p; every operation here is linear in the data, so Montgomery images (arkworks' in-memory Fp64) pass
 *      through unchanged.  `offset` / `alpha` are plain (canonical) values.  PARITY UNPINNED: the field crate is
 *      un-vendored and the reference holds no vector for it; the oracle restates the definitions (oracle/goldilocks.c). */
ss_status ss_ntt_gl64(ss_ctx *ctx, uint64_t *const *d_cols, uint32_t ncols, uint32_t log_n, int direction, uint64_t offset,
                      int in_order, int out_order);
ss_status ss_lde_gl64(ss_ctx *ctx, const uint64_t *const *d_in, uint32_t ncols, uint32_t log_n, uint32_t log_blowup,
                      uint64_t offset, uint64_t *const *d_evals, uint64_t *const *d_coeffs /* nullable; bit-reversed */);
/* one FRI layer over Fq3-valued evaluations (interleaved [2^log_len][3]) on the Fp domain domain_offset * <w>, natural
 * order: row j = {evals[j + k len/fold]}, d_out[j] = (degree < fold interpolant of row j)(alpha), alpha in Fq3;
 * flags: SS_FRI_UNNORMALISED multiplies by fold */
ss_status ss_fri_fold_gl64x3(ss_ctx *ctx, const uint64_t *d_evals, uint32_t log_len, uint32_t fold, const uint64_t alpha[3],
                             uint64_t domain_offset, uint32_t flags, uint64_t *d_out);

/* D1 over the cubic extension (ministark's DeepPolyComposer with Fq = Fq3: the out-of-domain point and every coefficient
 * are elements of Fq3 - three values < p each - the trace columns are Fp arrays; an Fq3-valued column (extension trace,
 * composition) is passed as its three component columns, its cells carrying the coefficients c, c X, c X^2 and the
 * out-of-domain value on the first of them: the sum is linear).
 * ss_ood_eval_gl64x3: out[3 j ..) = P_{cell_col[j]}(z * w_n^{cell_off[j]}) for bit-reversed coefficient columns (as
 *   ss_lde_gl64 writes them); out is HOST memory.
 * ss_deep_compose_gl64x3: d_out[i] (interleaved [n * blowup][3], the layout ss_fri_fold_gl64x3 reads) =
 *   sum_j coeff_trace[j] (T_{col_j}(x_i) - ood_trace[j]) / (x_i - z w_n^{off_j})
 *   + sum_k coeff_comp[k] (H_k(x_i) - ood_comp[k]) / (x_i - z_comp),      x_i = offset * w_{n blowup}^i,
 *   composed on the n-point sub-coset and extended per component (the polynomial has degree < n). */
/* A2 over the cubic extension (plain layout: Trace::build_extension_columns, layouts/src/plain/trace.rs:274-330): the running
 * quotient of a permutation argument, out[k] = prod_{i<=k} (z - (alpha nv_i + na_i)) / prod_{i<=k} (z - (alpha dv_i + da_i)) in Fq3,
 * item i reading element i * stride of each input (value pointers NULL: single-column argument, terms z - a_i), written as three
 * coordinate columns at row out_offset + k * out_stride.  last_out (nullable, host): the final quotient - 1 for a permutation. */
ss_status ss_running_product_gl64x3(ss_ctx *ctx, const uint64_t *d_num_addr, const uint64_t *d_num_val, const uint64_t *d_den_addr,
                                    const uint64_t *d_den_val, uint64_t stride, uint64_t count, const uint64_t z[3], const uint64_t alpha[3],
                                    uint64_t *const d_out[3], uint64_t out_stride, uint64_t out_offset, uint64_t last_out[3]);
/* H1 / openings for matrices of 8-byte elements: digest i = Keccak-256 or Blake2s-256 (hash_kind) of row i's elements as little-endian bytes, segment by
 * segment (element e of segment s = d_segments[s][i * seg_len + e]): a trace matrix is nseg columns with seg_len 1, the rows of
 * an Fq3 FRI layer ([len][3] interleaved, row j = {evals[j + k rows]}) are nseg = fold segments d_evals + 3 k rows of seg_len 3.
 * The tree over the digests is ss_merkle_build's SS_TREE_KECCAK / SS_TREE_BLAKE2S / SS_TREE_SHA256.  (The reference instantiates this
 * field with ministark's Sha256HashFn trees, cli/src/main.rs:105,119 - SS_HASH_SHA256 / SS_TREE_SHA256 here, the hash pinned by the
 * FIPS 180-4 vectors; how ministark lays a row out as bytes is not in the reference: little-endian elements, as for the other two.)  ss_gather_rows_gl64: the opened rows, to HOST memory
 * [nidx][nseg][seg_len]. */
ss_status ss_hash_rows_gl64(ss_ctx *ctx, int hash_kind, const uint64_t *const *d_segments, uint32_t nseg, uint32_t seg_len,
                            uint64_t nrows, uint8_t *d_digests);
ss_status ss_gather_rows_gl64(ss_ctx *ctx, const uint64_t *const *d_segments, uint32_t nseg, uint32_t seg_len, uint64_t nrows,
                              const uint64_t *idx, uint32_t nidx, uint64_t *out);
/* Q1 over the cubic extension: the program format of ss_eval_quotient (ss_air_program above) with accumulators, scratch
 * slots and constants in Fq3 - prog->consts holds n_consts x 3 values < p - and trace cells, tables (prog->d_tables: 8-byte
 * elements) and x in Fp, read into the first coordinate.  d_out: [n * blowup][3] interleaved.  Interpreted (no compiled
 * kernel for this field's experimental layout). */
ss_status ss_eval_quotient_gl64x3(ss_ctx *ctx, const ss_air_program *prog, const uint64_t *const *d_lde_cols, uint32_t ncols,
                                  uint32_t log_n, uint32_t log_blowup, uint64_t offset, uint64_t *d_out);
ss_status ss_ood_eval_gl64x3(ss_ctx *ctx, const uint64_t *const *d_coeffs_bitrev, uint32_t ncols, uint32_t log_n,
                             const uint32_t *cell_col, const uint32_t *cell_off, uint32_t ncells, const uint64_t z[3],
                             uint64_t *out);
ss_status ss_deep_compose_gl64x3(ss_ctx *ctx, const uint64_t *const *d_trace_lde, uint32_t ntrace_cols,
                                 const uint64_t *const *d_comp_lde, uint32_t ncomp, uint32_t log_n, uint32_t log_blowup,
                                 uint64_t offset, const uint32_t *mask_col, const uint32_t *mask_off, uint32_t nmask,
                                 const uint64_t *ood_trace, const uint64_t *coeff_trace, const uint64_t *ood_comp,
                                 const uint64_t *coeff_comp, const uint64_t z[3], const uint64_t z_comp[3], uint64_t *d_out);

/* ---- F1: one FRI layer fold (ministark FriProver::build_layers, un-vendored;
 *      defaults cli/src/main.rs:57-60).  d_evals: 2^log_len felts on
 *      domain_offset*<w>, natural order; row j = {evals[j + k*len/fold]};
 *      d_out[j] = degree<fold interpolant of row j evaluated at alpha.
 *      fold in {2,4,8,16}.  The next layer's offset is domain_offset^fold.
 *      The matrix whose rows get committed needs no entry point: its column k
 *      is the slice d_evals[k*len/fold ..) of the layer itself (no copy). */
ss_status ss_fri_fold(ss_ctx *ctx, const uint64_t *d_evals, uint32_t log_len, uint32_t fold,
                      const uint64_t alpha[4], const uint64_t domain_offset[4], uint64_t *d_out);
/* The same fold under the conventions found in the proof files the reference ships (data-only pin:
 * tests/golden/make_fri_golden.py, tests/golden/fri_saved_proofs.json).  flags = 0 is ss_fri_fold.
 *   SS_FRI_BITREV_ROWS    d_evals is in bit-reversed order: row r = evals[fold*r .. fold*r + fold), entry j of
 *                         the row at x_r * w_fold^bitrev(j) with x_r = domain_offset * w^bitrev(r); d_out is the
 *                         next layer, again in bit-reversed order (row r of this layer -> row r >> log2(fold),
 *                         slot r & (fold-1) of the next)
 *   SS_FRI_UNNORMALISED   d_out = fold * interpolant(alpha)  (StarkWare's fold: no 1/2 per halving)
 * example/array-sum.proof.saved and bootloader-proof.bin (masked-20 digests, the current code path) carry
 * BITREV_ROWS | UNNORMALISED; example/bootloader/bootloader-proof.bin (older path) carries flags = 0. */
enum { SS_FRI_BITREV_ROWS = 1, SS_FRI_UNNORMALISED = 2 };
ss_status ss_fri_fold_ex(ss_ctx *ctx, const uint64_t *d_evals, uint32_t log_len, uint32_t fold,
                         const uint64_t alpha[4], const uint64_t domain_offset[4], uint32_t flags,
                         uint64_t *d_out);

/* `count` rows of a layer from row `row0` on (one FRI layer folded by several GPUs): d_evals holds the
 * rows' `fold` entries column after column, entry k of row row0 + i at d_evals[k * count + i] (= the layer's
 * evals[row0 + i + k * len/fold]); d_out[i] = the fold of row row0 + i.  Natural order only (no SS_FRI_BITREV_ROWS). */
ss_status ss_fri_fold_rows(ss_ctx *ctx, const uint64_t *d_evals, uint32_t log_len, uint32_t fold,
                           const uint64_t alpha[4], const uint64_t domain_offset[4], uint32_t flags,
                           uint64_t row0, uint64_t count, uint64_t *d_out);

/* ---- C2: PublicCoin::grind_proof_of_work (crypto/src/public_coin/
 *      solidity.rs:120-141, cairo.rs:133-154).  Returns the SMALLEST nonce >= 1
 *      (the reference's non-parallel `find` semantics, solidity.rs:138). */
ss_status ss_pow_grind(ss_ctx *ctx, int coin_kind, const uint8_t digest[32], uint32_t bits,
                       uint64_t *nonce_out);

/* ---- builtins/src/pedersen/mod.rs:31-36 pedersen_hash, batched on device:
 *      d_out[i] = pedersen_hash(d_a[i], d_b[i]) (Montgomery felts in and out). */
ss_status ss_pedersen_hash(ss_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, uint64_t n,
                           uint64_t *d_out);

/* ---- A2 / X1: Trace::build_extension_columns (layouts/src/recursive/trace.rs:699-814,
 *      layouts/src/starknet/trace.rs:997-1100), the sequential host loops between the base-trace and the
 *      extension-trace commitments ("TODO: multithread", trace.rs:700), as device scans.
 *
 * An operand describes `array_chunks::<STEP>()` of one column: item k is the STEP felts at d_data + k*stride.
 * Its term is   z - (alpha * item[value_offset] + item[addr_offset])    (memory: trace.rs:713-714)
 * or            z - item[addr_offset]            when value_offset < 0  (range check / diluted check: trace.rs:727-728). */
typedef struct {
    const uint64_t *d_data;    /* device column, Montgomery felts */
    uint64_t stride;           /* felts per item */
    uint64_t addr_offset;      /* felt index of a_k (or of the single value x_k) inside item k */
    int64_t value_offset;      /* felt index of v_k inside item k; negative: single-value term */
} ss_perm_operand;
/* d_out[out_offset + i*out_stride] = numerator_acc_i * batch_inversion(denominator_acc)_i for i < count, where
 * *_acc_i is the product of the first i+1 terms of the operand (trace.rs:712-719, 767-769).  Like ark-ff's
 * batch_inversion a zero denominator product inverts to zero.  last_out (nullable, host): the final value — the
 * reference asserts it is one for the range-check and diluted-check products (trace.rs:734, 757-760); that check
 * stays with the caller.  The other cells of d_out are not touched (the memory and range-check products share a
 * column). */
ss_status ss_permutation_product(ss_ctx *ctx, const ss_perm_operand *num, const ss_perm_operand *den, uint64_t count,
                                 const uint64_t z[4], const uint64_t alpha[4], uint64_t *d_out, uint64_t out_stride,
                                 uint64_t out_offset, uint64_t last_out[4]);
/* Diluted-check aggregate (trace.rs:787-803): with x_i = d_ordered[i*stride + offset],
 * d_out[out_offset] = 1 and d_out[out_offset + i*out_stride] = acc_i,
 * acc_i = acc_{i-1} * (1 + z*u_i) + alpha * u_i^2,  u_i = x_i - x_{i-1},  for 0 < i < count. */
ss_status ss_diluted_aggregate(ss_ctx *ctx, const uint64_t *d_ordered, uint64_t stride, uint64_t offset, uint64_t count,
                               const uint64_t z[4], const uint64_t alpha[4], uint64_t *d_out, uint64_t out_stride,
                               uint64_t out_offset);

/* ---- the same loops (trace.rs:699-814 "TODO: multithread") over the ROW BLOCKS of several devices (ABI 12): device r holds
 *      items [r count, (r + 1) count) of the operands.  A running product is the block's own ss_permutation_product (its
 *      last_out = the block's total) times the totals of the blocks before it: d_data[offset + i*stride] *= factor, i < count. */
ss_status ss_scale_strided(ss_ctx *ctx, uint64_t *d_data, uint64_t stride, uint64_t offset, uint64_t count,
                           const uint64_t factor[4]);
/* The aggregate's recurrence acc_i = acc_{i-1} * m_i + c_i (m_i = 1 + z*u_i, c_i = alpha*u_i^2) composes as affine maps:
 * d_maps[2i], d_maps[2i+1] = (M_i, C_i) with acc_i = M_i * start + C_i, where `start` is the value before the block.
 * starts_column != 0: item 0 is the column's first cell (acc_0 = 1, a constant map).  Otherwise item 0 is left the identity:
 * its term needs the ordered value before the block, which another device holds; the caller composes that one map from the two
 * boundary values.  total_out (nullable, host): (M, C) of the block's last item.  d_maps: 2*count felts, the caller's. */
ss_status ss_diluted_aggregate_block(ss_ctx *ctx, const uint64_t *d_ordered, uint64_t stride, uint64_t offset, uint64_t count,
                                     int starts_column, const uint64_t z[4], const uint64_t alpha[4], uint64_t *d_maps,
                                     uint64_t total_out[8]);
/* d_out[out_offset + i*out_stride] = d_maps[2i] * start + d_maps[2i+1] for i < count. */
ss_status ss_affine_apply(ss_ctx *ctx, const uint64_t *d_maps, uint64_t count, const uint64_t start[4], uint64_t *d_out,
                          uint64_t out_stride, uint64_t out_offset);

/* ---- the BASE trace on the device (ABI 11; SURVEY.md section 8a row A1 / "next" row X1).  ExecutionTrace::new
 *      (layouts/src/starknet/trace.rs:99-987, layouts/src/recursive/trace.rs:89-688, layouts/src/utils.rs:112-152,
 *      357-380) makes the 9 / 7 base columns from `cairo-run`'s trace.bin / memory.bin with rayon loops and sequential
 *      sorts on the host; uploaded, they are 3.6 - 4.8 GB of PCIe traffic for cells that are functions of ~25 MB.  These entry
 *      points make the cells in HBM from the raw files' bytes: the caller (host/device_trace.cpp, which knows the layouts)
 *      uploads the files, a placement table and one template per DISTINCT builtin instance, and calls them in the order the
 *      reference's sections run (a later section overwrites an earlier one's cells, as there).  Everything on the ctx stream,
 *      no host sync; errors of the INPUT (a cell memory.bin does not hold, memory that is not continuous, ...) are bits of a
 *      status block the caller reads once at the end (ss_trace_status).  Columns: n = 16 * num_cycles Montgomery felts. */
/* where the CPU's cells sit in a cycle's 16 rows (CYCLE_HEIGHT; the Npc / RangeCheck / Auxiliary enums of
 * layouts/src/{recursive,starknet}/air.rs): the memory pool's 8 (address, value) pairs, the range-check and the auxiliary column's 16 cells */
enum { SS_TRACE_NPC_PAD = 0,      /* (1, the value at address 1): the padding pair (trace.rs:204-207)      */
       SS_TRACE_NPC_PUBLIC = 1,   /* (0, 0): a public-memory slot (Npc::PubMemAddr)                       */
       SS_TRACE_NPC_PC = 2, SS_TRACE_NPC_OP0 = 3, SS_TRACE_NPC_DST = 4, SS_TRACE_NPC_OP1 = 5 };  /* (pc, instruction), (op0 address, op0), ... */
enum { SS_TRACE_RC_FILL = 0,      /* the column's filler (rc_max / the pool's largest value); the pool's own cells come with ss_trace_rc_pool */
       SS_TRACE_RC_ZERO = 1, SS_TRACE_RC_OFF_DST = 2, SS_TRACE_RC_OFF_OP0 = 3, SS_TRACE_RC_OFF_OP1 = 4 };
enum { SS_TRACE_AUX_ZERO = 0, SS_TRACE_AUX_AP = 1, SS_TRACE_AUX_FP = 2, SS_TRACE_AUX_TMP0 = 3, SS_TRACE_AUX_TMP1 = 4, SS_TRACE_AUX_OP0_MUL_OP1 = 5,
       SS_TRACE_AUX_RES = 6 };
typedef struct { uint8_t npc_pair[8], rc_cell[16], aux_cell[16]; } ss_trace_layout;
/* the status block: SS_TRACE_STATUS_WORDS u32 in device memory, zeroed (ss_dev_zero) before the first call of a generation.
 * word 0: SS_TRACE_ERR_* bits; word 1: ~(the smallest cycle or address an error names); the rest: the ordered memory's counters */
#define SS_TRACE_STATUS_WORDS 16u
enum { SS_TRACE_ERR_MISSING_CELL = 1,        /* the run reads a cell memory.bin does not hold                        */
       SS_TRACE_ERR_NOT_INSTRUCTION = 2,     /* pc points at a word that is not an instruction (binary/src/lib.rs:565-721) */
       SS_TRACE_ERR_BAD_OP1_SOURCE = 4, SS_TRACE_ERR_BAD_RES_LOGIC = 8,
       SS_TRACE_ERR_NOT_AN_ADDRESS = 16,     /* op0 is used as an address and does not fit one                       */
       SS_TRACE_ERR_ADDRESS_RANGE = 32,      /* an address beyond n / 2: continuous memory cannot reach it            */
       SS_TRACE_ERR_PUBLIC_ZERO = 64,        /* a public-memory entry at address 0                                   */
       SS_TRACE_ERR_PUBLIC_CELLS = 128,      /* the pool's address-0 pairs are not exactly the public-memory slots    */
       SS_TRACE_ERR_NO_ONES = 256,           /* memory does not start at address 1                                   */
       SS_TRACE_ERR_NOT_SINGLE_VALUED = 512, SS_TRACE_ERR_NOT_CONTINUOUS = 1024,      /* utils.rs:132-150            */
       SS_TRACE_ERR_TOO_MANY_GAPS = 2048,    /* more unaccessed addresses than cycles to hold them (trace.rs:594-625) */
       SS_TRACE_ERR_FILL = 4096 };           /* the ordered accesses do not fill the column                          */
/* memory.bin on the device: d_records = the file's bytes (n_records x (u64 address, 32-byte little-endian word), uploaded by
 * the caller) -> d_image[address] as 4 x u64; cells the file does not name are marked (all-ones: not a field element).
 * cells: entries of d_image; records beyond it are dropped (no address above n / 2 can be accessed by a valid run). */
ss_status ss_trace_memory_image(ss_ctx *ctx, const uint64_t *d_records, uint64_t n_records, uint64_t *d_image, uint64_t cells);
/* The CPU's cells (starknet trace.rs:177-244, 294-302; recursive 172-232): one lane per cycle decodes the instruction at pc,
 * reads dst / op0 / op1 from d_image, computes res (dst^-1 for a conditional jump), tmp0, tmp1, op0 * op1, and the workgroup
 * writes the cycle's 16 rows of the flags, memory-pool, range-check and auxiliary columns whole (every cell: what no later
 * call overwrites keeps the padding written here) and the pool's 8 addresses per cycle as integers (d_pool_addr, n / 2 u32).
 * d_states: trace.bin's bytes ((ap, fp, pc) u64 triples).  pad_value: the value at address 1, Montgomery.  rc_fill: the
 * range-check column's filler. */
ss_status ss_trace_cpu_cells(ss_ctx *ctx, const ss_trace_layout *layout, const uint64_t *d_states, uint64_t num_cycles, const uint64_t *d_image,
                             uint64_t cells, const uint64_t pad_value[4], uint64_t rc_fill, uint64_t *d_flags, uint64_t *d_pool, uint64_t *d_range_check,
                             uint64_t *d_auxiliary, uint32_t *d_pool_addr, uint32_t *d_status);
/* A builtin's instances (Pedersen trace.rs:304-386, ECDSA 428-523, bitwise 525-705, EC op 707-777, Poseidon 779-888): block i
 * of `block_rows` rows holds one instance; cell e of the instance's template goes to column d_cells[e].col, row
 * i * block_rows + d_cells[e].off, and is the template's value e (SS_TRACE_CELL_VALUE: d_values[template * n_cells + e],
 * Montgomery) or the felt of the address addr_begin + addr_per_block * i + d_cells[e].arg (SS_TRACE_CELL_ADDRESS: an even row of
 * the memory pool; d_pool_addr gets the integer).  d_template_of_block: the template of every block (NULL: template 0
 * everywhere - a run whose instances are all the dummy one); n_templates: templates in d_values. */
typedef struct { uint32_t col, off, kind, arg; } ss_trace_cell;
enum { SS_TRACE_CELL_VALUE = 0, SS_TRACE_CELL_ADDRESS = 1 };
ss_status ss_trace_builtin(ss_ctx *ctx, uint64_t *const *d_cols, uint32_t ncols, const ss_trace_cell *d_cells, uint32_t n_cells, const uint64_t *d_values,
                           uint32_t n_templates, const uint32_t *d_template_of_block, uint64_t n_blocks, uint64_t block_rows, uint64_t addr_begin,
                           uint64_t addr_per_block, uint32_t *d_pool_addr);
/* The 16-bit range-check pool (utils.rs:357-380; starknet trace.rs:142-165, 246-292, 388-426).  The caller counts the pool's
 * values (65536 bins: the instructions' offsets, the builtin's parts) and hands over
 *   d_first[j], j <= rc_hi - rc_lo + 1: ordered values before value rc_lo + j (every value of [rc_lo, rc_hi] max(count, 1) times),
 *   d_padding[n_padding]: the values of [rc_lo, rc_hi] nothing uses, ascending (then rc_hi forever).
 * ss_trace_rc_pool writes, per cycle, its 16 / ordered_step ordered values at rows ordered_step * j + ordered_off and, on odd
 * cycles, padding value pad0 + cycle / 2 at row unused_off; ss_trace_rc_builtin slot s < n_slots of slot_rows rows: instance s of
 * d_given (3 u64 each: index, value low, value high) or, from n_given on, a dummy instance whose 8 parts are padding values
 * 8 (s - n_given) ...; the parts at rows part_stride * k + part_off of the range-check column, (addr_begin + index, value) at
 * rows pair_off, pair_off + 1 of the memory pool. */
typedef struct {
    uint64_t n_slots, n_given, slot_rows, addr_begin, n_padding, pad0;
    uint32_t part_stride, part_off, pair_off, rc_lo, rc_hi, ordered_step, ordered_off, unused_off;
} ss_trace_rc_plan;
ss_status ss_trace_rc_pool(ss_ctx *ctx, const ss_trace_rc_plan *plan, const uint32_t *d_first, const uint16_t *d_padding, uint64_t num_cycles,
                           uint64_t *d_range_check);
ss_status ss_trace_rc_builtin(ss_ctx *ctx, const ss_trace_rc_plan *plan, const uint64_t *d_given, const uint16_t *d_padding, uint64_t *d_range_check,
                              uint64_t *d_pool, uint32_t *d_pool_addr);
/* An ordered pool's column (the diluted check: starknet trace.rs:695-705, recursive 560-588): slot k < slots goes to
 * d_col[k * stride + offset]: zero for k < d_first[0], else the value lo + j with d_first[j] <= k < d_first[j + 1]
 * (j < n_values; d_first has n_values + 1 entries), diluted (bit i -> bit 4 i) if asked. */
ss_status ss_trace_ordered_runs(ss_ctx *ctx, uint64_t *d_col, uint64_t stride, uint64_t offset, uint64_t slots, const uint32_t *d_first,
                                uint32_t n_values, uint32_t lo, int diluted);
/* d_col[d_rows[k]] = felt(d_values[k]) for k < count (rows >= col_rows are skipped): the cells that are neither per cycle
 * nor per instance (the diluted pool's padding values: trace.rs:668-693) */
ss_status ss_trace_patch(ss_ctx *ctx, uint64_t *d_col, uint64_t col_rows, const uint64_t *d_rows, const uint64_t *d_values, uint64_t count);
/* get_ordered_memory_accesses (utils.rs:112-152) and the gap fillers before it (trace.rs:594-625 / 890-925), without a sort:
 * the pool's n / 2 accesses (d_pool_addr; values in d_pool) and the public memory's entries (d_public_addr / d_public_value
 * [n_public], the latter Montgomery; its remaining public_cells - n_public entries are (1, pad_value)) are counted per address,
 * the unaccessed addresses between the lowest and the highest become (address, 0) pairs at row unused_off of cycles 0, 1, ...
 * of d_pool, and d_memory receives (a, value(a)) x count(a) for a = 1, 2, ...; the reference's checks set SS_TRACE_ERR_* bits. */
ss_status ss_trace_ordered_memory(ss_ctx *ctx, uint64_t n, uint64_t *d_pool, uint64_t *d_memory, uint32_t *d_pool_addr, const uint32_t *d_public_addr,
                                  const uint64_t *d_public_value, uint32_t n_public, uint64_t public_cells, const uint64_t pad_value[4],
                                  uint32_t unused_off, uint32_t *d_status);
/* waits for the stream and reads the status block (SS_TRACE_STATUS_WORDS u32) */
ss_status ss_trace_status(ss_ctx *ctx, const uint32_t *d_status, uint32_t *status_out);

/* ---- per-kernel timing (bench.py's roofline leg): when enabled, every launch of
 *      the named kernel family on the ctx stream is bracketed by HIP events.
 *      ss_profile_read synchronises the stream and returns the accumulated device
 *      time and launch count since the last reset. */
enum { SS_PROF_NTT_PASS = 0, SS_PROF_HASH_ROWS = 1, SS_PROF_MERKLE = 2, SS_PROF_FRI = 3, SS_PROF_QUOTIENT = 4,
       SS_PROF_DEEP = 5, SS_PROF_EXT = 6, SS_PROF_TRACE = 7, SS_PROF_KINDS = 8 };
ss_status ss_profile_enable(ss_ctx *ctx, int on);
ss_status ss_profile_reset(ss_ctx *ctx);
ss_status ss_profile_read(ss_ctx *ctx, int kind, double *total_ms, uint64_t *launches);
/* ss_profile_enable(ctx, 2): besides the events, every profiled scope stamps the chip's constant-rate reference counter
 * (s_memrealtime, 100 MHz) before and after its launches, and ONE monitor wave on a stream of its own samples its shader-cycle
 * counter (s_memtime: one per compute unit, so only a single wave's readings are comparable) against that reference every ~20 us
 * for as long as level 2 is on (at most 3 s per read interval).  ss_profile_read_clock: the shader cycles - the monitor's count
 * interpolated at the scopes' reference stamps - and the reference ticks accumulated over this kernel family's scopes since the
 * last reset: cycles / ticks x 100 MHz is the clock the chip granted those kernels, in the run it is read in.  The stamps cost a
 * few microseconds per scope: level 2 is for a measuring pass, not a timed one.  (Up to 8192 scopes between two reads.) */
ss_status ss_profile_read_clock(ss_ctx *ctx, int kind, double *shader_cycles, double *ref_ticks);

/* Host-side pedersen_hash for the Fiat-Shamir coin (CairoVerifierPublicCoin::
 * reseed_with_field_elements hashes the OOD evaluations with a sequential
 * Pedersen chain, crypto/src/public_coin/cairo.rs:76-80; the coin stays on the
 * host).  Montgomery felts in and out; no device involved. */
ss_status ss_pedersen_hash_host(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);

/* Host-side Keccak-256 for the SolidityVerifierPublicCoin (crypto/src/public_coin/
 * solidity.rs:36-53 hashes 32..64-byte messages hundreds of times per proof: one
 * reseed per out-of-domain evaluation).  No device involved. */
ss_status ss_keccak256_host(const uint8_t *msg, size_t len, uint8_t out[32]);

/* ---- micro-benchmark hook: d_out[i] = d_a[i] * d_b[i] repeated `reps` times
 *      (dependent chain), used by bench.py to report mulmod/s. */
ss_status ss_fp252_mul_bench(ss_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, uint64_t n,
                             uint32_t reps, uint64_t *d_out);

#ifdef __cplusplus
}
#endif
#endif
