// prover.hpp — the proving pipeline above the C ABI, in C++ (the reference's host side is
// compiled Rust: ministark `Stark::prove` as driven by src/lib.rs:75-125 and
// cli/src/main.rs:180-213).  Pure sequencing: every arithmetic step on
// proof data is a call into libsandstorm_hip.so.
#pragma once
#include <cstdint>
#include <deque>
#include <functional>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/sandstorm_hip.h"
#include "air_program.hpp"
#include "coin.hpp"

namespace ssh {

struct ProofOptions {                   // cli/src/main.rs:51-60 defaults
    uint32_t num_queries = 65;
    uint32_t lde_blowup_factor = 2;
    uint32_t grinding_factor = 16;
    uint32_t fri_folding_factor = 8;
    uint32_t fri_max_remainder_coeffs = 16;
};

struct Conventions {                    // ministark-internal, SURVEY.md Appendix A
    uint64_t lde_offset = 3;            // M2
    uint32_t composition_columns = 2;   // M5
    // pinned by the reference's shipped proofs (tests/golden/make_fri_golden.py, make_proof_golden.py);
    // false reproduces the older code path's proofs
    bool bitrev_commit = true;          // M3: index i of a committed vector is the point offset * w^bitrev(i)
    bool fri_unnormalised = true;       // M8: fold = 8 * interpolant(alpha)
    bool remainder_unshifted = true;    // M9: remainder interpolates the folded last layer over the unshifted domain
    // pinned by the reference's own starknet proof (tests/test_layout_starknet.py): a layer is folded with the coin's draw
    // TIMES its domain offset (the reference folds over the unshifted domain).  false: the bare draw (round-1 proofs)
    bool fri_alpha_times_offset = true;
};

// RAII device allocation (ss_dev_alloc / ss_dev_free)
class DeviceBuffer {
public:
    DeviceBuffer(ss_ctx *ctx, size_t bytes);
    ~DeviceBuffer();
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    uint64_t *u64() const { return (uint64_t *)ptr_; }
    uint8_t *u8() const { return (uint8_t *)ptr_; }
    size_t bytes() const { return bytes_; }
private:
    ss_ctx *ctx_;
    void *ptr_ = nullptr;
    size_t bytes_;
};

// Column-major matrix resident in HBM (ministark::Matrix<Fp>); columns may be borrowed.
struct Matrix {
    std::vector<uint64_t *> cols;
    uint64_t nrows = 0;
    std::vector<std::shared_ptr<DeviceBuffer>> owned;
    static Matrix alloc(ss_ctx *ctx, uint32_t ncols, uint64_t nrows);
    uint32_t num_cols() const { return (uint32_t)cols.size(); }
};

class MerkleTree;
// The query phase's gathers - opened rows, authentication paths, leaf digests of every tree - collected and run as ONE ss_gather_batch:
// one index upload, one download, one synchronisation (27 calls of ~60 us with an idle device before: profiles/r05_host_gaps.txt).
// The output vectors are sized when queued and filled by run(); they must stay where they are until then.
class GatherBatch {
public:
    explicit GatherBatch(ss_ctx *ctx) : ctx_(ctx) {}
    // rows `idx` of a column-major matrix of 32-byte elements -> out[idx.size() * cols.size() * 4], row after row (ss_gather_rows)
    void rows(const std::vector<uint64_t *> &cols, const std::vector<uint64_t> &idx, std::vector<uint64_t> *out);
    // entries `idx` of one device array of `entry_bytes`-byte entries -> out
    void entries(const void *d_array, uint32_t entry_bytes, std::vector<uint64_t> idx, std::vector<uint8_t> *out);
    void run();
private:
    ss_ctx *ctx_;
    std::vector<ss_gather_job> jobs_;
    std::deque<std::vector<uint64_t>> idx_;             // (deque: the jobs point into these)
    std::deque<std::vector<const void *>> cols_;
};

// MatrixMerkleTree::from_matrix / MerkleTree::{root, prove} (crypto/src/merkle/mod.rs:72-123, 258-304)
class MerkleTree {
public:
    // order = SS_ORDER_BITREV: leaf i is row bitrev(i) of the natural-order matrix
    static std::unique_ptr<MerkleTree> from_matrix(ss_ctx *ctx, int tree_kind, uint32_t n_friendly, const Matrix &m,
                                                   int order = SS_ORDER_NATURAL);
    const std::array<uint8_t, 33> &root() const { return root_; }
    // nidx * log2(n) * 32 bytes; tags (optional): nidx * log2(n) MixedMerkleDigest tags of the path entries (friendly trees)
    std::vector<uint8_t> prove(const std::vector<uint64_t> &idx, std::vector<uint8_t> *tags = nullptr) const;
    // the row digests at these leaf indices (32 bytes each); empty for a single-column tree, whose leaves are the
    // elements themselves
    std::vector<uint8_t> leaf_digests(const std::vector<uint64_t> &idx) const;
    // prove + leaf_digests as jobs of a batch (the same bytes, after batch.run())
    void queue_openings(GatherBatch &batch, const std::vector<uint64_t> &idx, std::vector<uint8_t> *paths, std::vector<uint8_t> *tags,
                        std::vector<uint8_t> *leaves) const;
    uint64_t n() const { return n_; }
private:
    ss_ctx *ctx_ = nullptr;
    int tree_kind_ = 0;
    uint64_t n_ = 0;
    std::unique_ptr<DeviceBuffer> nodes_, tags_, leaves_;
    std::array<uint8_t, 33> root_{};
};

struct AirProgramData {
    Program program;
    const uint64_t *d_tables = nullptr;
    std::vector<uint32_t> table_desc;
};

class Air {                              // what the prover needs from an AirConfig
public:
    virtual ~Air() = default;
    std::string name;
    uint32_t num_base_columns = 0, num_extension_columns = 0, num_challenges = 0;
    std::vector<std::pair<uint32_t, uint32_t>> mask;     // trace_arguments(): sorted (column, offset)
    virtual AirProgramData build_program(uint64_t n, const std::vector<Felt> &challenges, const Felt &composition_coeff) = 0;
    // Optional, ahead of build_program: everything of the program that the challenges decide (its code, every constant but the powers
    // of the composition coefficient), so that the host builds it while the device extends and commits the extension trace and the
    // call above only patches the powers in.  Pure host work; a no-op for an AIR that does not take the hint.
    virtual void prepare_program(uint64_t, const std::vector<Felt> &) {}
    // the verifier's side of the out-of-domain identity: the composition constraint at z, its trace cells read from the
    // out-of-domain vector (in `mask` order), its tables evaluated as the functions they tabulate
    virtual Felt composition_at(uint64_t n, const std::vector<Felt> &challenges, const Felt &composition_coeff, const Felt &z,
                                const std::vector<Felt> &ood_trace);
};

struct Claim {                           // src/claims.rs:12-33
    Air *air = nullptr;
    int tree_kind = SS_TREE_KECCAK_M20;
    uint32_t n_friendly_layers = 0;      // 22 for the Cairo-verifier claims
    int coin_kind = SS_COIN_SOLIDITY;
};

struct FriLayerProof {
    std::array<uint8_t, 33> root{};
    uint32_t log_len = 0;
    std::vector<uint64_t> positions;
    std::vector<uint64_t> rows;          // positions x fold felts
    std::vector<uint8_t> paths;
    std::vector<uint8_t> path_tags;      // FriendlyMerkleTree: MixedMerkleDigest tag of every path entry
    std::vector<uint8_t> leaves;         // row digests of the opened rows (the wire format carries them)
};

struct Proof {
    ProofOptions options;
    int tree_kind = SS_TREE_KECCAK_M20;  // the claim's commitment scheme (decides the wire encoding of digests)
    uint64_t trace_len = 0;
    std::array<uint8_t, 33> base_root{}, extension_root{}, composition_root{};
    bool has_extension = false;
    std::vector<Felt> challenges, ood_trace, ood_composition, fri_alphas, fri_remainder;
    Felt composition_coeff{}, z{}, deep_alpha{};
    std::vector<FriLayerProof> fri_layers;
    uint64_t pow_nonce = 0;
    std::vector<uint64_t> query_positions;
    std::vector<uint64_t> base_rows, extension_rows, composition_rows;
    std::vector<uint8_t> base_paths, extension_paths, composition_paths;
    std::vector<uint8_t> base_leaves, extension_leaves, composition_leaves;
    std::vector<uint8_t> base_path_tags, extension_path_tags, composition_path_tags;   // FriendlyMerkleTree only
    std::vector<uint8_t> serialize() const;          // flat dump with the transcript values, for the tests
    // The reference's proof bytes (ministark `Proof`, ark-serialize compressed) as pinned by its shipped proof files
    // (sandstorm_amd/wire.py has the layout; tests/golden/make_proof_golden.py the evidence).  FriendlyMerkleTree proofs
    // use the MixedMerkleDigest / FriendlyMerkleTreeProof encodings of crypto/src/merkle/mixed.rs:46-101 and
    // mod.rs:168-236 (source-pinned: the reference ships no such file).
    std::vector<uint8_t> serialize_wire() const;
};

class PublicCoin;
// FRI commit phase / proof of work / FRI query phase on one device (steps 8-9 of Prover::prove; the sharded prover runs them on rank 0)
struct FriLayerState { std::unique_ptr<MerkleTree> tree; Matrix matrix; std::shared_ptr<DeviceBuffer> evals; };
std::vector<FriLayerState> fri_commit_phase(ss_ctx *ctx, const Claim &claim, const Conventions &conv, const ProofOptions &opt, PublicCoin &coin,
                                            Proof &proof, std::shared_ptr<DeviceBuffer> deep, uint32_t log_N, uint64_t n);
std::vector<FriLayerState> fri_commit_phase_from(ss_ctx *ctx, const Claim &claim, const Conventions &conv, const ProofOptions &opt, PublicCoin &coin,
                                                 Proof &proof, std::shared_ptr<DeviceBuffer> evals, uint32_t log_len, Felt offset, uint64_t degree_bound);
uint64_t proof_of_work(ss_ctx *ctx, const Claim &claim, PublicCoin &coin, const ProofOptions &opt, bool have_nonce, uint64_t nonce);
// (batch: the openings join it and the caller runs it; NULL: a batch of their own)
void fri_open_from(ss_ctx *ctx, const Conventions &conv, const ProofOptions &opt, Proof &proof, std::vector<FriLayerState> &layers_from,
                   const std::vector<uint64_t> &positions, size_t first, GatherBatch *batch = nullptr);
void fri_open(ss_ctx *ctx, const Conventions &conv, const ProofOptions &opt, Proof &proof, std::vector<FriLayerState> &layers,
              const std::vector<uint64_t> &positions, GatherBatch *batch = nullptr);

// build_extension_columns(&challenges) (layouts/src/recursive/trace.rs:699-814): returns the
// extension columns, resident in HBM
using ExtensionBuilder = std::function<Matrix(const std::vector<Felt> &challenges)>;

// Base columns that arrive while the proof is already running: `wait(c)` returns once everything enqueued on the context's stream
// from then on sees column c of the base trace (the trace generator's thread uploads a column the moment no section will write it
// again: trace_*.hpp column_done, ss_upload_async / ss_wait_upload); `order` = the order the columns become final in.  The base
// trace is then extended column by column as it lands instead of as one batch - `Stark::prove` starts from the witness
// (src/lib.rs:94-100, cli/src/main.rs:200-202): generation, upload and the first transforms overlap.
struct ColumnFeed {
    std::function<void(uint32_t)> wait;
    std::vector<uint32_t> order;
};

class Prover {
public:
    Prover(ss_ctx *ctx, const Claim &claim, const ProofOptions &opt = ProofOptions(), const Conventions &conv = Conventions())
        : ctx_(ctx), claim_(claim), opt_(opt), conv_(conv) {}
    Proof prove(const Digest &coin_seed, const Matrix &base_trace, const ExtensionBuilder &build_extension);
    // Use this proof-of-work nonce instead of grinding, if it is valid for the transcript (else prove() throws).  Any
    // nonce with enough leading zeros is a valid proof (the reference's grinder returns whichever its parallel search
    // finds first - `find_any`, crypto/src/public_coin/solidity.rs:120-141); the GPU grinder returns the smallest.
    // Supplying the reference's nonce makes the whole proof comparable byte for byte.
    void set_pow_nonce(uint64_t nonce) { have_nonce_ = true; nonce_ = nonce; }
    void set_base_feed(ColumnFeed feed) { feed_ = std::move(feed); }
private:
    bool have_nonce_ = false;
    uint64_t nonce_ = 0;
    ss_ctx *ctx_;
    Claim claim_;
    ProofOptions opt_;
    Conventions conv_;
    ColumnFeed feed_;
};

// the layouts' AIRs
struct AirPublicInput;
// the real `recursive` layout (air_recursive.cpp; mirror of sandstorm_amd/layouts/recursive.py)
std::unique_ptr<Air> make_recursive_air(ss_ctx *ctx, const AirPublicInput &pi, uint32_t log_n, uint32_t log_blowup, uint64_t lde_offset);
std::unique_ptr<Air> make_starknet_air(ss_ctx *ctx, const AirPublicInput &pi, uint32_t log_n, uint32_t log_blowup, uint64_t lde_offset);
std::vector<uint64_t> layout_air_tables(const Air &air);         // table descriptions of a layout AIR (host-side checks)

}  // namespace ssh
