// sharded.hpp — ONE proof over the GPUs of a node, in the C++ host.
//
// The reference has nothing to match: its parallelism is rayon loops inside one address space
// (crypto/src/merkle/utils.rs:30-32, layouts/src/starknet/trace.rs:182).  One process (or thread) per GPU, each with its own
// ss_ctx; the ranks meet in a Transport: RCCL over xGMI on a multi-GPU node (ss_comm_* of the C ABI), or - for ranks that are
// threads of one process, the way the tests and a single-GPU box run it - a local group that copies between the contexts.
// The proof is the single-device prover's, byte for byte (tests/test_sharded_host.py; the Python mirror of this driver is
// sandstorm_amd/sharded_prover.py, whose module text has the distribution table).
#pragma once
#include <map>
#include <memory>
#include <vector>

#include "prover.hpp"

namespace ssh {

struct Message { uint32_t peer; void *ptr; uint64_t bytes; };       // device memory of this rank's context

class Transport {
public:
    virtual ~Transport() = default;
    uint32_t rank = 0, world = 1;
    // One exchange step that EVERY rank enters, whatever it has to send or receive; the messages of one ordered pair of ranks
    // are matched in list order (ss_comm_exchange).  Runs on the context's stream.
    virtual void exchange(ss_ctx *ctx, const std::vector<Message> &sends, const std::vector<Message> &recvs) = 0;
    // `mine.size()` host bytes of every rank (the same count everywhere), in rank order
    virtual std::vector<uint8_t> all_gather(ss_ctx *ctx, const std::vector<uint8_t> &mine) = 0;
    // any number of bytes per rank -> one vector per rank
    std::vector<std::vector<uint8_t>> all_gather_var(ss_ctx *ctx, const std::vector<uint8_t> &mine);
};

// RCCL: `id` = the 128 bytes of ss_comm_unique_id made by rank 0 and handed out by whatever launched the ranks
std::unique_ptr<Transport> make_rccl_transport(ss_ctx *ctx, const uint8_t id[128], uint32_t rank, uint32_t world);
// The caller's own collectives (MPI, gloo, UCX ...: one process per GPU on a node without RCCL, and how the CPU suite runs this
// driver as 2 / 4 / 8 PROCESSES over the emulated device code).  Both callbacks are entered by every rank, in the same order on
// every rank, and return 0 on success.  All buffers are HOST memory the transport stages through (ss_download / ss_upload).
struct TransportCallbacks {
    void *user = nullptr;
    // MPI_Alltoallv on bytes: `send` = the bytes for rank 0, 1, ... back to back (send_bytes[p] of them for rank p), `recv` the
    // same way with recv_bytes[p] from rank p; this rank's own slot has length 0 (the transport copies on the device)
    int (*all_to_all)(void *user, const uint8_t *send, const uint64_t *send_bytes, uint8_t *recv, const uint64_t *recv_bytes) = nullptr;
    // `bytes` of every rank (the same count everywhere) -> out[p * bytes ...), in rank order
    int (*all_gather)(void *user, const uint8_t *mine, uint64_t bytes, uint8_t *out) = nullptr;
};
std::unique_ptr<Transport> make_callback_transport(const TransportCallbacks &cb, uint32_t rank, uint32_t world);
// ranks = threads of one process (every thread with its own context, on one device or several)
class LocalGroup;
std::shared_ptr<LocalGroup> make_local_group(uint32_t world);
std::unique_ptr<Transport> make_local_transport(std::shared_ptr<LocalGroup> group, uint32_t rank);
void local_group_fail(LocalGroup &group);          // a rank gave up: the others leave their barriers with an error

// Every rank enters: two messages of different sizes per ordered pair of ranks (list-order matching), one message to itself, an
// all_gather and an all_gather_var, all with contents that name (source, destination, message); throws on the rank that sees a
// wrong byte.  bandwidth_bytes > 0: then one timed equal-split all-to-all of that many bytes per pair -> this rank's
// send + receive rate in GB/s through *gbps (what a re-shard gets out of the links).
void transport_self_check(ss_ctx *ctx, Transport &comm, uint64_t bandwidth_bytes = 0, double *gbps = nullptr);

// build_extension_columns on the ranks: -> {global column number: device column of n felts} for the extension columns this
// rank owns (column c lives on rank c % world)
using ShardedExtensionBuilder = std::function<std::map<uint32_t, uint64_t *>(const std::vector<Felt> &challenges)>;

// the same as ROW BLOCKS (extension.hpp build_extension_blocks): on EVERY rank, rows [rank n / world, (rank + 1) n / world) of all
// the extension columns, in column order - the scans divide over the ranks and the owner's scatter falls away
using ShardedExtensionBlocks = std::function<std::vector<uint64_t *>(const std::vector<Felt> &challenges)>;

class ShardedProver {
public:
    ShardedProver(ss_ctx *ctx, const Claim &claim, Transport &comm, const ProofOptions &opt = ProofOptions(), const Conventions &conv = Conventions())
        : ctx_(ctx), claim_(claim), comm_(comm), opt_(opt), conv_(conv) {}
    // my_base: {column c: device column of n felts} for the base columns with c % world == rank.  Every rank calls it; the proof
    // comes out on rank 0 (-> true there, false on the others).
    bool prove(const Digest &coin_seed, const std::map<uint32_t, uint64_t *> &my_base, const ShardedExtensionBuilder &build_extension,
               uint64_t n, Proof *out);
    void set_pow_nonce(uint64_t nonce) { have_nonce_ = true; nonce_ = nonce; }
    // the extension trace comes as row blocks from this builder; prove()'s build_extension is not called
    void set_extension_blocks(ShardedExtensionBlocks b) { ext_blocks_ = std::move(b); }
private:
    struct Commitment;
    using Buf = std::shared_ptr<DeviceBuffer>;
    uint32_t owner(uint32_t col) const { return col % comm_.world; }
    std::vector<Buf> to_row_blocks(const std::map<uint32_t, Buf> &owned, uint32_t ncols, uint32_t first_col, uint64_t N, uint64_t halo);
    std::unique_ptr<Commitment> commit(const std::vector<Buf> &blocks, uint64_t N, int order);
    std::unique_ptr<Commitment> commit(const std::vector<const uint64_t *> &blocks, uint64_t N, int order);
    void open(const Commitment &com, const std::vector<Buf> &blocks, uint64_t N, const std::vector<uint64_t> &positions, int order,
              std::vector<uint64_t> *rows, std::vector<uint8_t> *paths, std::vector<uint8_t> *leaves, std::vector<uint8_t> *tags);
    void open(const Commitment &com, const std::vector<const uint64_t *> &blocks, uint64_t N, const std::vector<uint64_t> &positions, int order,
              std::vector<uint64_t> *rows, std::vector<uint8_t> *paths, std::vector<uint8_t> *leaves, std::vector<uint8_t> *tags);
    // ONE transform over the ranks (ss_ntt_shard_fp252): blocks of n / R values in natural order -> blocks of the bit-reversed
    // coefficient array, and those (2^-log_expand sub-sampled) -> blocks of the evaluations; several vectors per call
    std::vector<Buf> exchange_layout(const std::vector<Buf> &in, uint64_t elems);
    std::vector<Buf> spread_inverse(const std::vector<Buf> &blocks, uint32_t log_n, const Felt *offset);
    std::vector<Buf> spread_forward(const std::vector<Buf> &coeff_blocks, uint32_t log_n, uint32_t log_expand, const Felt *offset);
    std::vector<Buf> with_halo(const std::vector<Buf> &blocks, uint64_t B, uint64_t halo);
    ss_ctx *ctx_;
    Claim claim_;
    Transport &comm_;
    ProofOptions opt_;
    Conventions conv_;
    bool have_nonce_ = false;
    uint64_t nonce_ = 0;
    ShardedExtensionBlocks ext_blocks_;
};

// MerkleTreeConfig::hash_nodes on the host for the log2(world) levels above the ranks' sub-trees (33 bytes: digest + tag)
std::array<uint8_t, 33> merge_nodes(int tree_kind, uint32_t n_friendly_layers, uint32_t depth, const std::array<uint8_t, 33> &left,
                                    const std::array<uint8_t, 33> &right);

}  // namespace ssh
