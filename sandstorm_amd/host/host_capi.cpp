// host_capi.cpp — C entry points of libsandstorm_host.so: the C++ prover for callers
// without C++ (bench.py and the tests reach it through ctypes).
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <cstring>
#include <stdexcept>
#include <string>

#include "extension.hpp"
#include "goldilocks_prover.hpp"
#include "prover.hpp"
#include "sharded.hpp"
#include "public_input.hpp"
#include "trace_recursive.hpp"
#include "trace_starknet.hpp"
#include "verifier.hpp"

using namespace ssh;

namespace { thread_local std::string g_err; }

extern "C" {

typedef struct ssh_air ssh_air;
// build_extension_columns callback: fill d_cols_out[0..next) with device pointers of the
// extension columns for these challenges; return 0 on success
typedef int (*ssh_extension_cb)(void *user, const uint64_t *challenges, uint32_t nchallenges, uint64_t **d_cols_out);

const char *ssh_last_error(void) { return g_err.c_str(); }
// bumped whenever an entry point of this file changes its signature or meaning (hostlib.py checks it at load; ss_abi_version is
// the device library's).  2: ssh_prove_sharded takes transport handles (ssh_rccl_group_create / ssh_callback_group_create)
// instead of the raw RCCL id; ssh_air_create left for the callers' own AIR objects.  3: ssh_prove_files, ssh_base_trace_cb,
// ssh_callback_group_create, ssh_group_self_check.  4: ssh_prove_sharded_blocks, ssh_build_extension_blocks (the extension trace
// as row blocks over the ranks).
#define SSH_HOST_ABI_VERSION 4
uint32_t ssh_abi_version(void) { return SSH_HOST_ABI_VERSION; }

// an `ssh_air` handle is an `Air *` (prover.hpp): the layouts' AIRs come from ssh_air_create_recursive / _starknet below; a caller
// with an AIR of its own (the tests' mini AIR: tests/cpp/mini_air_lib.cpp) hands in its own object
void ssh_air_destroy(ssh_air *a) { delete reinterpret_cast<Air *>(a); }
uint32_t ssh_air_columns(const ssh_air *a, int which) {
    const Air *air = reinterpret_cast<const Air *>(a);
    return which == 0 ? air->num_base_columns : which == 1 ? air->num_extension_columns : (uint32_t)air->mask.size();
}

}  // extern "C"

// format 0: the flat test dump (Proof::serialize); 1: the reference's wire format (Proof::serialize_wire)
static int prove_impl(ss_ctx *ctx, ssh_air *air_h, int tree_kind, uint32_t n_friendly_layers, int coin_kind, const uint8_t seed[32],
                      uint64_t *const *d_base, uint32_t nbase, uint32_t log_n, ssh_extension_cb cb, void *user,
                      const uint32_t options[5], int format, uint8_t **proof_bytes, uint64_t *proof_len,
                      const uint64_t *pow_nonce = nullptr) {
    try {
        Air *air = reinterpret_cast<Air *>(air_h);
        Claim claim;
        claim.air = air; claim.tree_kind = tree_kind; claim.n_friendly_layers = n_friendly_layers; claim.coin_kind = coin_kind;
        ProofOptions opt;
        if (options) {
            opt.num_queries = options[0]; opt.lde_blowup_factor = options[1]; opt.grinding_factor = options[2];
            opt.fri_folding_factor = options[3]; opt.fri_max_remainder_coeffs = options[4];
        }
        Matrix base;
        base.nrows = 1ull << log_n;
        for (uint32_t c = 0; c < nbase; ++c) base.cols.push_back(d_base[c]);
        Digest sd;
        memcpy(sd.data(), seed, 32);
        Prover prover(ctx, claim, opt);
        if (pow_nonce) prover.set_pow_nonce(*pow_nonce);
        Proof proof = prover.prove(sd, base, [&](const std::vector<Felt> &ch) {
            Matrix ext;
            ext.nrows = base.nrows;
            std::vector<uint64_t> flat(4 * ch.size());
            for (size_t i = 0; i < ch.size(); ++i) memcpy(flat.data() + 4 * i, ch[i].data(), 32);
            std::vector<uint64_t *> cols(air->num_extension_columns, nullptr);
            if (!cb || cb(user, flat.data(), (uint32_t)ch.size(), cols.data()) != 0) throw std::runtime_error("extension callback failed");
            ext.cols = cols;
            return ext;
        });
        if (proof_bytes && proof_len) {
            std::vector<uint8_t> b = format == 1 ? proof.serialize_wire() : proof.serialize();
            *proof_bytes = (uint8_t *)malloc(b.size());
            memcpy(*proof_bytes, b.data(), b.size());
            *proof_len = b.size();
        }
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}

extern "C" {

// options: {num_queries, lde_blowup_factor, grinding_factor, fri_folding_factor, fri_max_remainder_coeffs}
int ssh_prove(ss_ctx *ctx, ssh_air *air_h, int tree_kind, uint32_t n_friendly_layers, int coin_kind, const uint8_t seed[32],
              uint64_t *const *d_base, uint32_t nbase, uint32_t log_n, ssh_extension_cb cb, void *user, const uint32_t options[5],
              uint8_t **proof_bytes, uint64_t *proof_len) {
    return prove_impl(ctx, air_h, tree_kind, n_friendly_layers, coin_kind, seed, d_base, nbase, log_n, cb, user, options, 0,
                      proof_bytes, proof_len);
}
// same, with the proof in the reference's wire format (what `sandstorm-cli prove --output` writes: cli/src/main.rs:204-213)
int ssh_prove_wire(ss_ctx *ctx, ssh_air *air_h, int tree_kind, uint32_t n_friendly_layers, int coin_kind, const uint8_t seed[32],
                   uint64_t *const *d_base, uint32_t nbase, uint32_t log_n, ssh_extension_cb cb, void *user,
                   const uint32_t options[5], uint8_t **proof_bytes, uint64_t *proof_len) {
    return prove_impl(ctx, air_h, tree_kind, n_friendly_layers, coin_kind, seed, d_base, nbase, log_n, cb, user, options, 1,
                      proof_bytes, proof_len);
}
// same, with a given proof-of-work nonce instead of grinding (Prover::set_pow_nonce: must be valid for the transcript)
int ssh_prove_wire_with_nonce(ss_ctx *ctx, ssh_air *air_h, int tree_kind, uint32_t n_friendly_layers, int coin_kind, const uint8_t seed[32],
                              uint64_t *const *d_base, uint32_t nbase, uint32_t log_n, ssh_extension_cb cb, void *user,
                              const uint32_t options[5], uint64_t pow_nonce, uint8_t **proof_bytes, uint64_t *proof_len) {
    return prove_impl(ctx, air_h, tree_kind, n_friendly_layers, coin_kind, seed, d_base, nbase, log_n, cb, user, options, 1,
                      proof_bytes, proof_len, &pow_nonce);
}
void ssh_free(void *p) { free(p); }

// ---- one proof over several ranks (sharded.hpp).  Every rank calls ssh_prove_sharded with its own context and AIR handle; the
// proof (reference wire format) comes out on rank 0.  Transport: a local group (ranks = threads of this process:
// ssh_local_group_create) or an RCCL communicator (ssh_rccl_group_create: made ONCE from the 128 bytes of ss_comm_unique_id that
// rank 0 hands out - an id's bootstrap serves one ncclCommInitRank per rank - and kept across proofs).
typedef struct ssh_local_group ssh_local_group;
ssh_local_group *ssh_local_group_create(uint32_t world) { return reinterpret_cast<ssh_local_group *>(new std::shared_ptr<LocalGroup>(make_local_group(world))); }
void ssh_local_group_destroy(ssh_local_group *g) { delete reinterpret_cast<std::shared_ptr<LocalGroup> *>(g); }
typedef struct ssh_rccl_group ssh_rccl_group;
ssh_rccl_group *ssh_rccl_group_create(ss_ctx *ctx, const uint8_t rccl_id[128], uint32_t rank, uint32_t world) {
    try {
        if (!ctx || !rccl_id) throw std::runtime_error("ssh_rccl_group_create: NULL argument");
        return reinterpret_cast<ssh_rccl_group *>(make_rccl_transport(ctx, rccl_id, rank, world).release());
    } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
void ssh_rccl_group_destroy(ssh_rccl_group *g) { delete reinterpret_cast<Transport *>(g); }
// The caller's own collectives instead of RCCL (sharded.hpp TransportCallbacks: an all-to-all and an all-gather of HOST bytes,
// entered by every rank; MPI_Alltoallv / MPI_Allgather fit as they are): ranks that are processes on a node without RCCL, and how
// the CPU suite runs this driver as 2 / 4 / 8 processes over gloo.  The handle is passed to ssh_prove_sharded where an
// ssh_rccl_group goes (both are the driver's `Transport`) and freed by ssh_rccl_group_destroy.
typedef int (*ssh_all_to_all_cb)(void *user, const uint8_t *send, const uint64_t *send_bytes, uint8_t *recv, const uint64_t *recv_bytes);
typedef int (*ssh_all_gather_cb)(void *user, const uint8_t *mine, uint64_t bytes, uint8_t *out);
ssh_rccl_group *ssh_callback_group_create(uint32_t rank, uint32_t world, ssh_all_to_all_cb all_to_all, ssh_all_gather_cb all_gather, void *user) {
    try {
        TransportCallbacks cb;
        cb.user = user; cb.all_to_all = all_to_all; cb.all_gather = all_gather;
        return reinterpret_cast<ssh_rccl_group *>(make_callback_transport(cb, rank, world).release());
    } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
// Before the first proof over a group: every rank enters; messages of different sizes between every ordered pair of ranks with
// contents that name (source, destination, message), an all-gather, a variable-length all-gather (sharded.hpp transport_self_check).
// -> 0, or 1 with the mismatch in ssh_last_error on the rank that saw it.  bandwidth_bytes > 0: then one timed equal-split
// all-to-all of that many bytes per pair -> *gbps_out = this rank's send + receive rate.  group / transport: as ssh_prove_sharded.
int ssh_group_self_check(ss_ctx *ctx, uint32_t rank, uint32_t world, ssh_local_group *group, ssh_rccl_group *transport, uint64_t bandwidth_bytes,
                         double *gbps_out) {
    std::shared_ptr<LocalGroup> *lg = reinterpret_cast<std::shared_ptr<LocalGroup> *>(group);
    try {
        if (!ctx || (!group && !transport)) throw std::runtime_error("ssh_group_self_check: NULL argument");
        std::unique_ptr<Transport> local = lg ? make_local_transport(*lg, rank) : nullptr;
        Transport *comm = lg ? local.get() : reinterpret_cast<Transport *>(transport);
        if (comm->world != world || comm->rank != rank) throw std::runtime_error("ssh_group_self_check: the group has another number of ranks, or this is another rank of it");
        transport_self_check(ctx, *comm, bandwidth_bytes, gbps_out);
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        if (lg) local_group_fail(**lg);
        return 1;
    }
}
typedef int (*ssh_sharded_extension_cb)(void *user, const uint64_t *challenges, uint32_t nchallenges, uint32_t *cols_out, uint64_t **d_cols_out,
                                        uint32_t *ncols_out);
// the extension trace as ROW BLOCKS (ssh_build_extension_blocks): d_blocks_out = one block of 2^log_n / world rows per extension column
typedef int (*ssh_extension_blocks_cb)(void *user, const uint64_t *challenges, uint32_t nchallenges, uint64_t **d_blocks_out);
}  // extern "C"
static int prove_sharded_impl(ss_ctx *ctx, ssh_air *air_h, int tree_kind, uint32_t n_friendly_layers, int coin_kind, const uint8_t seed[32], uint32_t rank,
                              uint32_t world, ssh_local_group *group, ssh_rccl_group *rccl, const uint32_t *base_cols, uint64_t *const *d_base,
                              uint32_t nbase_mine, uint32_t log_n, ssh_sharded_extension_cb cb, ssh_extension_blocks_cb blocks_cb, void *user,
                              const uint32_t options[5], uint8_t **proof_bytes, uint64_t *proof_len) {
    std::shared_ptr<LocalGroup> *lg = reinterpret_cast<std::shared_ptr<LocalGroup> *>(group);
    try {
        if (!ctx || !air_h || !seed || (!group && !rccl)) throw std::runtime_error("ssh_prove_sharded: NULL argument");
        Air *air = reinterpret_cast<Air *>(air_h);
        Claim claim;
        claim.air = air; claim.tree_kind = tree_kind; claim.n_friendly_layers = n_friendly_layers; claim.coin_kind = coin_kind;
        ProofOptions opt;
        if (options) {
            opt.num_queries = options[0]; opt.lde_blowup_factor = options[1]; opt.grinding_factor = options[2];
            opt.fri_folding_factor = options[3]; opt.fri_max_remainder_coeffs = options[4];
        }
        std::unique_ptr<Transport> local = lg ? make_local_transport(*lg, rank) : nullptr;
        Transport *comm = lg ? local.get() : reinterpret_cast<Transport *>(rccl);
        if (comm->world != world || comm->rank != rank) throw std::runtime_error("ssh_prove_sharded: the group has another number of ranks, or this is another rank of it");
        std::map<uint32_t, uint64_t *> mine;
        for (uint32_t k = 0; k < nbase_mine; ++k) mine[base_cols[k]] = d_base[k];
        Digest sd;
        memcpy(sd.data(), seed, 32);
        ShardedProver prover(ctx, claim, *comm, opt);
        if (blocks_cb)
            prover.set_extension_blocks([&](const std::vector<Felt> &ch) {
                std::vector<uint64_t> flat(4 * ch.size());
                for (size_t i = 0; i < ch.size(); ++i) memcpy(flat.data() + 4 * i, ch[i].data(), 32);
                std::vector<uint64_t *> blks(air->num_extension_columns, nullptr);
                if (blocks_cb(user, flat.data(), (uint32_t)ch.size(), blks.data()) != 0) throw std::runtime_error("extension callback failed");
                for (uint64_t *b : blks) if (!b) throw std::runtime_error("extension callback: a row block is missing");
                return blks;
            });
        Proof proof;
        const bool have = prover.prove(sd, mine, [&](const std::vector<Felt> &ch) {
            std::vector<uint64_t> flat(4 * ch.size());
            for (size_t i = 0; i < ch.size(); ++i) memcpy(flat.data() + 4 * i, ch[i].data(), 32);
            uint32_t cols[16], ncols = 0;
            uint64_t *ptrs[16];
            if (!cb || cb(user, flat.data(), (uint32_t)ch.size(), cols, ptrs, &ncols) != 0 || ncols > 16) throw std::runtime_error("extension callback failed");
            std::map<uint32_t, uint64_t *> out;
            for (uint32_t k = 0; k < ncols; ++k) out[cols[k]] = ptrs[k];
            return out;
        }, 1ull << log_n, &proof);
        if (proof_bytes && proof_len) { *proof_bytes = nullptr; *proof_len = 0; }
        if (have && proof_bytes && proof_len) {
            const std::vector<uint8_t> b = proof.serialize_wire();
            *proof_bytes = (uint8_t *)malloc(b.size());
            memcpy(*proof_bytes, b.data(), b.size());
            *proof_len = b.size();
        }
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        if (lg) local_group_fail(**lg);
        return 1;
    }
}
extern "C" {
int ssh_prove_sharded(ss_ctx *ctx, ssh_air *air_h, int tree_kind, uint32_t n_friendly_layers, int coin_kind, const uint8_t seed[32], uint32_t rank,
                      uint32_t world, ssh_local_group *group, ssh_rccl_group *rccl, const uint32_t *base_cols, uint64_t *const *d_base,
                      uint32_t nbase_mine, uint32_t log_n, ssh_sharded_extension_cb cb, void *user, const uint32_t options[5],
                      uint8_t **proof_bytes, uint64_t *proof_len) {
    return prove_sharded_impl(ctx, air_h, tree_kind, n_friendly_layers, coin_kind, seed, rank, world, group, rccl, base_cols, d_base, nbase_mine, log_n, cb, nullptr,
                              user, options, proof_bytes, proof_len);
}
// The same proof with the extension trace built as row blocks on EVERY rank (the scans of Trace::build_extension_columns divided
// over the ranks: ssh_build_extension_blocks inside the callback) - no owner, no scatter of whole extension columns.
int ssh_prove_sharded_blocks(ss_ctx *ctx, ssh_air *air_h, int tree_kind, uint32_t n_friendly_layers, int coin_kind, const uint8_t seed[32], uint32_t rank,
                             uint32_t world, ssh_local_group *group, ssh_rccl_group *rccl, const uint32_t *base_cols, uint64_t *const *d_base,
                             uint32_t nbase_mine, uint32_t log_n, ssh_extension_blocks_cb cb, void *user, const uint32_t options[5],
                             uint8_t **proof_bytes, uint64_t *proof_len) {
    if (!cb) { g_err = "ssh_prove_sharded_blocks: NULL callback"; return 1; }
    return prove_sharded_impl(ctx, air_h, tree_kind, n_friendly_layers, coin_kind, seed, rank, world, group, rccl, base_cols, d_base, nbase_mine, log_n, nullptr, cb,
                              user, options, proof_bytes, proof_len);
}

// Trace::build_extension_columns as row blocks over the ranks of a group (extension.hpp build_extension_blocks).  d_aux: as
// ssh_build_extension_columns, but pointing at THIS rank's rows [rank n / world, (rank + 1) n / world) of the auxiliary columns;
// trace_len = n.  group / transport: as ssh_prove_sharded (every rank of the group enters, with the same challenges; callable
// from inside ssh_prove_sharded_blocks' callback).  The matrix holds the same rows of the extension columns.
typedef struct ssh_matrix ssh_matrix;
int ssh_build_extension_blocks(ss_ctx *ctx, int layout, const uint64_t *const *d_aux, uint64_t trace_len, uint32_t rank, uint32_t world,
                               ssh_local_group *group, ssh_rccl_group *transport, const uint64_t *challenges, int check, ssh_matrix **out) {
    std::shared_ptr<LocalGroup> *lg = reinterpret_cast<std::shared_ptr<LocalGroup> *>(group);
    try {
        if (!ctx || !d_aux || !challenges || !out || (!group && !transport)) throw std::runtime_error("ssh_build_extension_blocks: NULL argument");
        std::unique_ptr<Transport> local = lg ? make_local_transport(*lg, rank) : nullptr;
        Transport *comm = lg ? local.get() : reinterpret_cast<Transport *>(transport);
        if (comm->world != world || comm->rank != rank) throw std::runtime_error("ssh_build_extension_blocks: the group has another number of ranks, or this is another rank of it");
        TraceColumns c;
        c.npc = d_aux[0]; c.memory = d_aux[1]; c.range_check = d_aux[2]; c.trace_len = trace_len;
        if (layout == 1) { c.diluted_unordered = d_aux[3]; c.diluted_ordered = d_aux[4]; }
        std::vector<Felt> ch(6);
        for (int i = 0; i < 6; ++i) memcpy(ch[i].data(), challenges + 4 * i, 32);
        BlockGather g;
        g.rank = rank; g.world = world;
        g.all_gather = [&](const std::vector<uint8_t> &mine) { return comm->all_gather(ctx, mine); };
        *out = reinterpret_cast<ssh_matrix *>(new Matrix(build_extension_blocks(ctx, layout == 1 ? "recursive" : "starknet", c, ch, g, check != 0)));
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        if (lg) local_group_fail(**lg);
        return 1;
    }
}


// Trace::build_extension_columns (extension.hpp).  layout: 1 = recursive, 2 = starknet (the ssh_air_create kinds);
// d_aux = {npc, memory, range_check [, diluted_unordered, diluted_ordered]}; challenges = 6 felts.  The result is a
// matrix handle owning its device columns (3 for recursive, 1 for starknet).
typedef struct ssh_matrix ssh_matrix;
int ssh_build_extension_columns(ss_ctx *ctx, int layout, const uint64_t *const *d_aux, uint64_t trace_len,
                                const uint64_t *challenges, int check, ssh_matrix **out) {
    try {
        TraceColumns c;
        c.npc = d_aux[0]; c.memory = d_aux[1]; c.range_check = d_aux[2]; c.trace_len = trace_len;
        if (layout == 1) { c.diluted_unordered = d_aux[3]; c.diluted_ordered = d_aux[4]; }
        std::vector<Felt> ch(6);
        for (int i = 0; i < 6; ++i) memcpy(ch[i].data(), challenges + 4 * i, 32);
        *out = reinterpret_cast<ssh_matrix *>(new Matrix(build_extension_columns(ctx, layout == 1 ? "recursive" : "starknet", c, ch, check != 0)));
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}
uint32_t ssh_matrix_num_cols(const ssh_matrix *m) { return reinterpret_cast<const Matrix *>(m)->num_cols(); }
uint64_t *ssh_matrix_col(const ssh_matrix *m, uint32_t k) { return reinterpret_cast<const Matrix *>(m)->cols[k]; }
void ssh_matrix_destroy(ssh_matrix *m) { delete reinterpret_cast<Matrix *>(m); }

// CairoPublicCoin::from_public_input (public_input.hpp).  layout: 1 = recursive, 2 = starknet; segments: 9 x
// {present, begin_addr, stop_ptr} in the header's order; memory: n x (address, 4 little-endian limbs of the value)
int ssh_public_coin_seed(int layout, uint32_t rc_min, uint32_t rc_max, uint64_t n_steps, const uint32_t *segments,
                         const uint32_t *mem_addresses, const uint64_t *mem_values, uint64_t n_mem, int coin_kind,
                         uint8_t seed_out[32], uint64_t *elements_out, uint32_t *n_elements) {
    try {
        AirPublicInput pi;
        pi.layout = layout == 1 ? "recursive" : layout == 2 ? "starknet" : "unknown";
        pi.rc_min = (uint16_t)rc_min; pi.rc_max = (uint16_t)rc_max; pi.n_steps = n_steps;
        for (int k = 0; k < 9; ++k) { pi.segments[k].present = segments[3 * k] != 0; pi.segments[k].begin_addr = segments[3 * k + 1]; pi.segments[k].stop_ptr = segments[3 * k + 2]; }
        pi.public_memory.resize(n_mem);
        for (uint64_t i = 0; i < n_mem; ++i) { pi.public_memory[i].address = mem_addresses[i]; memcpy(pi.public_memory[i].value.data(), mem_values + 4 * i, 32); }
        const auto els = public_input_elements(pi, coin_kind);
        if (elements_out && n_elements) { for (size_t i = 0; i < els.size(); ++i) memcpy(elements_out + 4 * i, els[i].data(), 32); *n_elements = (uint32_t)els.size(); }
        const Digest d = public_coin_seed(pi, coin_kind);
        memcpy(seed_out, d.data(), 32);
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}

// Base trace of the recursive layout (trace_recursive.hpp) from the raw `cairo-run` files.  Public input as in
// ssh_public_coin_seed; instances: pedersen n x (index, a[4], b[4]) as 9 u64 each, range_check n x (index, value[4]) as 5,
// bitwise n x (index, x[4], y[4]) as 9.  columns_out: 7 arrays of 4 * (16 * cycles) u64 (Montgomery limbs).
int ssh_recursive_base_trace(const uint8_t *trace_bin, uint64_t trace_len, const uint8_t *memory_bin, uint64_t memory_len,
                             uint32_t rc_min, uint32_t rc_max, uint64_t n_steps, const uint32_t *segments, const uint32_t *mem_addresses,
                             const uint64_t *mem_values, uint64_t n_mem, const uint64_t *pedersen, uint64_t n_pedersen,
                             const uint64_t *range_check, uint64_t n_range_check, const uint64_t *bitwise, uint64_t n_bitwise,
                             uint64_t *const *columns_out) {
    try {
        AirPublicInput pi;
        pi.layout = "recursive";
        pi.rc_min = (uint16_t)rc_min; pi.rc_max = (uint16_t)rc_max; pi.n_steps = n_steps;
        for (int k = 0; k < 9; ++k) { pi.segments[k].present = segments[3 * k] != 0; pi.segments[k].begin_addr = segments[3 * k + 1]; pi.segments[k].stop_ptr = segments[3 * k + 2]; }
        pi.public_memory.resize(n_mem);
        for (uint64_t i = 0; i < n_mem; ++i) { pi.public_memory[i].address = mem_addresses[i]; memcpy(pi.public_memory[i].value.data(), mem_values + 4 * i, 32); }
        PrivateInput priv;
        for (uint64_t i = 0; i < n_pedersen; ++i) { PedersenInstance p; p.index = (uint32_t)pedersen[9 * i]; memcpy(p.a.data(), pedersen + 9 * i + 1, 32); memcpy(p.b.data(), pedersen + 9 * i + 5, 32); priv.pedersen.push_back(p); }
        for (uint64_t i = 0; i < n_range_check; ++i) { RangeCheckInstance p; p.index = (uint32_t)range_check[5 * i]; memcpy(p.value.data(), range_check + 5 * i + 1, 32); priv.range_check.push_back(p); }
        for (uint64_t i = 0; i < n_bitwise; ++i) { BitwiseInstance p; p.index = (uint32_t)bitwise[9 * i]; memcpy(p.x.data(), bitwise + 9 * i + 1, 32); memcpy(p.y.data(), bitwise + 9 * i + 5, 32); priv.bitwise.push_back(p); }
        const auto states = read_register_states(trace_bin, trace_len);
        std::vector<U256> memory;
        std::vector<uint8_t> present;
        read_memory(memory_bin, memory_len, memory, present);
        recursive_base_trace_into(reinterpret_cast<Felt *const *>(columns_out), states, memory, present, pi, priv);
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}

// Base trace of the starknet layout (trace_starknet.hpp).  A flat instance list per builtin, as 1 + 4 k u64 each (index, then
// k 256-bit values): pedersen k = 2 (a, b), range_check 1, ecdsa 4 (pubkey x, message, r, w), bitwise 2 (x, y),
// ec_op 5 (p.x, p.y, q.x, q.y, m), poseidon 3.  counts: the six instance counts in that order.  columns_out: 9 arrays.
int ssh_starknet_base_trace(const uint8_t *trace_bin, uint64_t trace_len, const uint8_t *memory_bin, uint64_t memory_len,
                            uint32_t rc_min, uint32_t rc_max, uint64_t n_steps, const uint32_t *segments, const uint32_t *mem_addresses,
                            const uint64_t *mem_values, uint64_t n_mem, const uint64_t *const *instances, const uint64_t *counts,
                            uint64_t *const *columns_out) {
    try {
        AirPublicInput pi;
        pi.layout = "starknet";
        pi.rc_min = (uint16_t)rc_min; pi.rc_max = (uint16_t)rc_max; pi.n_steps = n_steps;
        for (int k = 0; k < 9; ++k) { pi.segments[k].present = segments[3 * k] != 0; pi.segments[k].begin_addr = segments[3 * k + 1]; pi.segments[k].stop_ptr = segments[3 * k + 2]; }
        pi.public_memory.resize(n_mem);
        for (uint64_t i = 0; i < n_mem; ++i) { pi.public_memory[i].address = mem_addresses[i]; memcpy(pi.public_memory[i].value.data(), mem_values + 4 * i, 32); }
        auto val = [](const uint64_t *rec, int k) { U256 v; memcpy(v.data(), rec + 1 + 4 * k, 32); return v; };
        StarknetPrivateInput priv;
        for (uint64_t i = 0; i < counts[0]; ++i) { const uint64_t *r = instances[0] + 9 * i; priv.pedersen.push_back(PedersenInstance{(uint32_t)r[0], val(r, 0), val(r, 1)}); }
        for (uint64_t i = 0; i < counts[1]; ++i) { const uint64_t *r = instances[1] + 5 * i; priv.range_check.push_back(RangeCheckInstance{(uint32_t)r[0], val(r, 0)}); }
        for (uint64_t i = 0; i < counts[2]; ++i) { const uint64_t *r = instances[2] + 17 * i; priv.ecdsa.push_back(EcdsaInstance{(uint32_t)r[0], val(r, 0), val(r, 1), val(r, 2), val(r, 3)}); }
        for (uint64_t i = 0; i < counts[3]; ++i) { const uint64_t *r = instances[3] + 9 * i; priv.bitwise.push_back(BitwiseInstance{(uint32_t)r[0], val(r, 0), val(r, 1)}); }
        for (uint64_t i = 0; i < counts[4]; ++i) { const uint64_t *r = instances[4] + 21 * i; priv.ec_op.push_back(EcOpInstance{(uint32_t)r[0], val(r, 0), val(r, 1), val(r, 2), val(r, 3), val(r, 4)}); }
        for (uint64_t i = 0; i < counts[5]; ++i) { const uint64_t *r = instances[5] + 13 * i; priv.poseidon.push_back(PoseidonInstance{(uint32_t)r[0], {val(r, 0), val(r, 1), val(r, 2)}}); }
        const auto states = read_register_states(trace_bin, trace_len);
        std::vector<U256> memory;
        std::vector<uint8_t> present;
        read_memory(memory_bin, memory_len, memory, present);
        Felt *out[9];
        for (int c = 0; c < 9; ++c) out[c] = reinterpret_cast<Felt *>(columns_out[c]);
        starknet_base_trace_into(out, states, memory, present, pi, priv);
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}

static AirPublicInput public_input_from_args(int layout, uint32_t rc_min, uint32_t rc_max, uint64_t n_steps, const uint32_t *segments,
                                             const uint32_t *mem_addresses, const uint64_t *mem_values, uint64_t n_mem) {
    AirPublicInput pi;
    pi.layout = layout == 1 ? "recursive" : layout == 2 ? "starknet" : "unknown";
    pi.rc_min = (uint16_t)rc_min; pi.rc_max = (uint16_t)rc_max; pi.n_steps = n_steps;
    for (int k = 0; k < 9; ++k) { pi.segments[k].present = segments[3 * k] != 0; pi.segments[k].begin_addr = segments[3 * k + 1]; pi.segments[k].stop_ptr = segments[3 * k + 2]; }
    pi.public_memory.resize(n_mem);
    for (uint64_t i = 0; i < n_mem; ++i) { pi.public_memory[i].address = mem_addresses[i]; memcpy(pi.public_memory[i].value.data(), mem_values + 4 * i, 32); }
    return pi;
}

// The real `recursive` AIR for a public input (air_recursive.cpp).  ctx may be NULL: no device tables are built and the
// handle only serves ssh_air_dump (host-side checks).
int ssh_air_create_recursive(ss_ctx *ctx, uint32_t rc_min, uint32_t rc_max, uint64_t n_steps, const uint32_t *segments,
                             const uint32_t *mem_addresses, const uint64_t *mem_values, uint64_t n_mem, uint32_t log_n, uint32_t log_blowup,
                             ssh_air **out) {
    try {
        const AirPublicInput pi = public_input_from_args(1, rc_min, rc_max, n_steps, segments, mem_addresses, mem_values, n_mem);
        *out = reinterpret_cast<ssh_air *>(make_recursive_air(ctx, pi, log_n, log_blowup, 3).release());
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}
// The real `starknet` AIR (air_starknet.cpp); same conventions.
int ssh_air_create_starknet(ss_ctx *ctx, uint32_t rc_min, uint32_t rc_max, uint64_t n_steps, const uint32_t *segments,
                            const uint32_t *mem_addresses, const uint64_t *mem_values, uint64_t n_mem, uint32_t log_n, uint32_t log_blowup,
                            ssh_air **out) {
    try {
        const AirPublicInput pi = public_input_from_args(2, rc_min, rc_max, n_steps, segments, mem_addresses, mem_values, n_mem);
        *out = reinterpret_cast<ssh_air *>(make_starknet_air(ctx, pi, log_n, log_blowup, 3).release());
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}
// ---- files -> proof with generation, upload and the first transforms overlapped (cli/src/main.rs:200-202 times
// generate_trace + prove).  A trace job is a layout's generator with its inputs parsed: layout 1 = recursive, 2 = starknet; the
// public input as in ssh_public_coin_seed; instances / counts as ssh_starknet_base_trace takes them (the recursive layout reads
// pedersen, range_check and bitwise: entries 0, 1, 3).
namespace {
struct TraceJob {
    uint32_t ncols = 0;
    uint64_t n = 0;
    std::vector<uint32_t> order;                 // the order the columns become final in (the generators' column_done calls)
    std::function<void(Felt *const *out, const std::function<void(int)> *done)> run;
    std::function<void(ss_ctx *ctx, uint64_t *const *d_cols)> run_device;        // the same columns made in HBM (device_trace.hpp)
};
TraceJob make_trace_job(int layout, const uint8_t *trace_bin, uint64_t trace_len, const uint8_t *memory_bin, uint64_t memory_len, uint32_t rc_min,
                        uint32_t rc_max, uint64_t n_steps, const uint32_t *segments, const uint32_t *mem_addresses, const uint64_t *mem_values,
                        uint64_t n_mem, const uint64_t *const *instances, const uint64_t *counts) {
    if (layout != 1 && layout != 2) throw std::runtime_error("layout: 1 = recursive, 2 = starknet");
    if (!trace_bin || (!memory_bin && memory_len) || !segments || (n_mem && (!mem_addresses || !mem_values))) throw std::runtime_error("trace job: NULL argument");
    // the public input and the files must be of the same run: the columns are sized by trace.bin, the statement by n_steps
    if (trace_len % 24) throw std::runtime_error("trace: trace file is not a sequence of (ap, fp, pc) u64 triples");
    if (!trace_len || ((trace_len / 24) & (trace_len / 24 - 1))) throw std::runtime_error("trace: the number of cycles must be a power of two");
    if (n_steps != trace_len / 24) throw std::runtime_error("trace job: the public input declares " + std::to_string(n_steps) + " steps, trace.bin holds " + std::to_string(trace_len / 24));
    if (rc_min > 0xffff || rc_max > 0xffff) throw std::runtime_error("trace job: rc_min / rc_max are 16-bit values");
    auto pi = std::make_shared<AirPublicInput>();
    pi->layout = layout == 1 ? "recursive" : "starknet";
    pi->rc_min = (uint16_t)rc_min; pi->rc_max = (uint16_t)rc_max; pi->n_steps = n_steps;
    for (int k = 0; k < 9; ++k) { pi->segments[k].present = segments[3 * k] != 0; pi->segments[k].begin_addr = segments[3 * k + 1]; pi->segments[k].stop_ptr = segments[3 * k + 2]; }
    pi->public_memory.resize(n_mem);
    for (uint64_t i = 0; i < n_mem; ++i) { pi->public_memory[i].address = mem_addresses[i]; memcpy(pi->public_memory[i].value.data(), mem_values + 4 * i, 32); }
    auto val = [](const uint64_t *rec, int k) { U256 v; memcpy(v.data(), rec + 1 + 4 * k, 32); return v; };
    auto cnt = [&](int k) -> uint64_t { return counts && instances && instances[k] ? counts[k] : 0; };
    auto priv = std::make_shared<StarknetPrivateInput>();
    for (uint64_t i = 0; i < cnt(0); ++i) { const uint64_t *r = instances[0] + 9 * i; priv->pedersen.push_back(PedersenInstance{(uint32_t)r[0], val(r, 0), val(r, 1)}); }
    for (uint64_t i = 0; i < cnt(1); ++i) { const uint64_t *r = instances[1] + 5 * i; priv->range_check.push_back(RangeCheckInstance{(uint32_t)r[0], val(r, 0)}); }
    for (uint64_t i = 0; i < cnt(2); ++i) { const uint64_t *r = instances[2] + 17 * i; priv->ecdsa.push_back(EcdsaInstance{(uint32_t)r[0], val(r, 0), val(r, 1), val(r, 2), val(r, 3)}); }
    for (uint64_t i = 0; i < cnt(3); ++i) { const uint64_t *r = instances[3] + 9 * i; priv->bitwise.push_back(BitwiseInstance{(uint32_t)r[0], val(r, 0), val(r, 1)}); }
    for (uint64_t i = 0; i < cnt(4); ++i) { const uint64_t *r = instances[4] + 21 * i; priv->ec_op.push_back(EcOpInstance{(uint32_t)r[0], val(r, 0), val(r, 1), val(r, 2), val(r, 3), val(r, 4)}); }
    for (uint64_t i = 0; i < cnt(5); ++i) { const uint64_t *r = instances[5] + 13 * i; priv->poseidon.push_back(PoseidonInstance{(uint32_t)r[0], {val(r, 0), val(r, 1), val(r, 2)}}); }
    const RegisterStates states(trace_bin, trace_len);         // read in place: the caller's bytes outlive the job (both callers run it before they return)
    auto memory = std::make_shared<std::vector<U256>>();
    auto present = std::make_shared<std::vector<uint8_t>>();
    read_memory(memory_bin, memory_len, *memory, *present);
    TraceJob job;
    job.n = 16 * (uint64_t)states.size();
    if (layout == 1) {
        job.ncols = 7;
        job.order = {1, 2, 0, 6, 5, 3, 4};        // the diluted pair, flags, auxiliary, range check, memory pool, sorted memory
        job.run = [=](Felt *const *out, const std::function<void(int)> *done) {
            PrivateInput rp;
            rp.pedersen = priv->pedersen; rp.range_check = priv->range_check; rp.bitwise = priv->bitwise;
            recursive_base_trace_into(out, states, *memory, *present, *pi, rp, done);
        };
        job.run_device = [=](ss_ctx *ctx, uint64_t *const *d_cols) {
            PrivateInput rp;
            rp.pedersen = priv->pedersen; rp.range_check = priv->range_check; rp.bitwise = priv->bitwise;
            recursive_base_trace_device(ctx, d_cols, trace_bin, trace_len, memory_bin, memory_len, *memory, *present, *pi, rp);
        };
    } else {
        job.ncols = 9;
        job.order = {0, 1, 2, 3, 4, 7, 8, 5, 6};  // flags, the four Pedersen columns, range check, auxiliary, memory pool, sorted memory
        job.run = [=](Felt *const *out, const std::function<void(int)> *done) {
            Felt *o[9];
            for (int c = 0; c < 9; ++c) o[c] = out[c];
            starknet_base_trace_into(o, states, *memory, *present, *pi, *priv, done);
        };
        job.run_device = [=](ss_ctx *ctx, uint64_t *const *d_cols) {
            starknet_base_trace_device(ctx, d_cols, trace_bin, trace_len, memory_bin, memory_len, *memory, *present, *pi, *priv);
        };
    }
    return job;
}
}  // namespace

// the generators with a per-column callback: column_done(user, c) is called (on the calling thread, between two sections of the
// generator) as soon as column c of columns_out is final.  order_out (optional, room for 9): the order the callbacks come in.
typedef void (*ssh_column_cb)(void *user, int column);
int ssh_base_trace_cb(int layout, const uint8_t *trace_bin, uint64_t trace_len, const uint8_t *memory_bin, uint64_t memory_len, uint32_t rc_min,
                      uint32_t rc_max, uint64_t n_steps, const uint32_t *segments, const uint32_t *mem_addresses, const uint64_t *mem_values,
                      uint64_t n_mem, const uint64_t *const *instances, const uint64_t *counts, uint64_t *const *columns_out, ssh_column_cb column_done,
                      void *user, uint32_t *order_out) {
    try {
        const TraceJob job = make_trace_job(layout, trace_bin, trace_len, memory_bin, memory_len, rc_min, rc_max, n_steps, segments, mem_addresses, mem_values,
                                            n_mem, instances, counts);
        if (order_out) for (uint32_t k = 0; k < job.ncols; ++k) order_out[k] = job.order[k];
        const std::function<void(int)> done = [&](int c) { if (column_done) column_done(user, c); };
        job.run(reinterpret_cast<Felt *const *>(columns_out), &done);
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}

// the base columns made ON the device from the raw files (csrc/trace.hip; host/device_trace.hpp): d_cols = 7 / 9 device columns of
// 16 * cycles felts (ss_dev_alloc'd or the caller's own device memory).  Returns when the columns are final (the input's errors -
// a cell memory.bin does not hold, memory that is not continuous, ... - are the host generator's refusals).
int ssh_base_trace_device(ss_ctx *ctx, int layout, const uint8_t *trace_bin, uint64_t trace_len, const uint8_t *memory_bin, uint64_t memory_len, uint32_t rc_min,
                          uint32_t rc_max, uint64_t n_steps, const uint32_t *segments, const uint32_t *mem_addresses, const uint64_t *mem_values, uint64_t n_mem,
                          const uint64_t *const *instances, const uint64_t *counts, uint64_t *const *d_cols) {
    try {
        if (!ctx || !d_cols) throw std::runtime_error("ssh_base_trace_device: NULL argument");
        const TraceJob job = make_trace_job(layout, trace_bin, trace_len, memory_bin, memory_len, rc_min, rc_max, n_steps, segments, mem_addresses, mem_values,
                                            n_mem, instances, counts);
        for (uint32_t c = 0; c < job.ncols; ++c) if (!d_cols[c]) throw std::runtime_error("ssh_base_trace_device: NULL column");
        job.run_device(ctx, d_cols);
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}

// `sandstorm-cli prove` from the raw files to the proof bytes in ONE call: the generator runs on a thread of its own and writes the
// caller's PINNED host columns (the GpuAllocator seam, layouts/src/recursive/trace.rs:115-120); every column leaves for the device
// (d_cols) on the context's copy stream the moment it is final (ss_upload_async), and the prover - on the calling thread - extends
// the columns as they land (ColumnFeed) and goes on as ssh_prove_wire does.  times_out (optional, 2 doubles): the generator's wall
// time and the whole call's.  The proof is the one ssh_prove_wire writes from the same columns (tests/test_gpu_full_size.py).
int ssh_prove_files(ss_ctx *ctx, int layout, const uint8_t *trace_bin, uint64_t trace_len, const uint8_t *memory_bin, uint64_t memory_len, uint32_t rc_min,
                    uint32_t rc_max, uint64_t n_steps, const uint32_t *segments, const uint32_t *mem_addresses, const uint64_t *mem_values, uint64_t n_mem,
                    const uint64_t *const *instances, const uint64_t *counts, uint64_t *const *pinned_cols, uint64_t *const *d_cols, ssh_air *air_h,
                    int tree_kind, uint32_t n_friendly_layers, int coin_kind, const uint8_t seed[32], ssh_extension_cb cb, void *user,
                    const uint32_t options[5], double *times_out, uint8_t **proof_bytes, uint64_t *proof_len) {
    try {
        if (!ctx || !air_h || !seed || !pinned_cols || !d_cols) throw std::runtime_error("ssh_prove_files: NULL argument");
        const auto t_start = std::chrono::steady_clock::now();
        const TraceJob job = make_trace_job(layout, trace_bin, trace_len, memory_bin, memory_len, rc_min, rc_max, n_steps, segments, mem_addresses, mem_values,
                                            n_mem, instances, counts);
        Air *air = reinterpret_cast<Air *>(air_h);
        if (air->num_base_columns != job.ncols) throw std::runtime_error("ssh_prove_files: the AIR is another layout's");
        Claim claim;
        claim.air = air; claim.tree_kind = tree_kind; claim.n_friendly_layers = n_friendly_layers; claim.coin_kind = coin_kind;
        ProofOptions opt;
        if (options) {
            opt.num_queries = options[0]; opt.lde_blowup_factor = options[1]; opt.grinding_factor = options[2];
            opt.fri_folding_factor = options[3]; opt.fri_max_remainder_coeffs = options[4];
        }
        std::mutex m;
        std::condition_variable cv;
        std::vector<uint64_t> ticket(job.ncols, 0);
        bool failed = false;
        std::string err;
        double gen_s = 0.0;
        const std::function<void(int)> done = [&](int c) {           // the generator's thread: column c is final -> its upload leaves
            uint64_t t = 0;
            const ss_status st = ss_upload_async(ctx, d_cols[c], pinned_cols[c], 32 * (size_t)job.n, &t);
            std::lock_guard<std::mutex> lk(m);
            if (st != SS_OK) { failed = true; err = ss_last_error(); } else ticket[c] = t;
            cv.notify_all();
        };
        std::thread producer([&] {
            try { job.run(reinterpret_cast<Felt *const *>(pinned_cols), &done); }
            catch (const std::exception &e) { std::lock_guard<std::mutex> lk(m); failed = true; err = e.what(); cv.notify_all(); }
            gen_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        });
        std::vector<char> waited(job.ncols, 0);
        // when the prover (or the generator) gives up half way: the producer is joined, every upload that left is waited for - a ticket
        // nobody waited for is released, the copy stream is drained - before the caller gets its pinned columns back
        struct Joiner {
            std::thread &t; ss_ctx *ctx; std::vector<uint64_t> &ticket; std::vector<char> &waited;
            ~Joiner() {
                if (t.joinable()) t.join();
                for (size_t c = 0; c < ticket.size(); ++c) if (ticket[c] && !waited[c]) (void)ss_wait_upload(ctx, ticket[c]);
                (void)ss_ctx_sync(ctx);
            }
        } joiner{producer, ctx, ticket, waited};
        Matrix base;
        base.nrows = job.n;
        for (uint32_t c = 0; c < job.ncols; ++c) base.cols.push_back(d_cols[c]);
        ColumnFeed feed;
        feed.order = job.order;
        feed.wait = [&](uint32_t c) {
            uint64_t t;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return ticket[c] != 0 || failed; });
                if (failed) throw std::runtime_error(err);
                t = ticket[c];
                waited[c] = 1;
            }
            if (ss_wait_upload(ctx, t) != SS_OK) throw std::runtime_error(ss_last_error());
        };
        Digest sd;
        memcpy(sd.data(), seed, 32);
        Prover prover(ctx, claim, opt);
        prover.set_base_feed(feed);
        uint32_t log_n = 0;
        while ((1ull << log_n) < job.n) ++log_n;
        Proof proof = prover.prove(sd, base, [&](const std::vector<Felt> &ch) {
            Matrix ext;
            ext.nrows = base.nrows;
            std::vector<uint64_t> flat(4 * ch.size());
            for (size_t i = 0; i < ch.size(); ++i) memcpy(flat.data() + 4 * i, ch[i].data(), 32);
            std::vector<uint64_t *> cols(air->num_extension_columns, nullptr);
            if (!cb || cb(user, flat.data(), (uint32_t)ch.size(), cols.data()) != 0) throw std::runtime_error("extension callback failed");
            ext.cols = cols;
            return ext;
        });
        producer.join();
        if (ss_ctx_sync(ctx) != SS_OK) throw std::runtime_error(ss_last_error());
        if (times_out) { times_out[0] = gen_s; times_out[1] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); }
        if (proof_bytes && proof_len) {
            const std::vector<uint8_t> b = proof.serialize_wire();
            *proof_bytes = (uint8_t *)malloc(b.size());
            memcpy(*proof_bytes, b.data(), b.size());
            *proof_len = b.size();
        }
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}

// files -> proof with the base trace made ON the device: trace.bin / memory.bin go up as they are (25 MB where the host-made columns
// are 3.6 - 4.8 GB of PCIe traffic), csrc/trace.hip makes the columns in d_cols, and the prover goes on as ssh_prove_wire does - what
// the reference's "Proof generated in" timer wraps (cli/src/main.rs:200-202: generate_trace + prove).  times_out (optional, 2
// doubles): until the columns were final (the device's generation, waited for: the input's errors are reported before anything is
// proven), and the whole call.  The proof is the one ssh_prove_files / ssh_prove_wire write for the same statement.
int ssh_prove_files_device(ss_ctx *ctx, int layout, const uint8_t *trace_bin, uint64_t trace_len, const uint8_t *memory_bin, uint64_t memory_len, uint32_t rc_min,
                           uint32_t rc_max, uint64_t n_steps, const uint32_t *segments, const uint32_t *mem_addresses, const uint64_t *mem_values, uint64_t n_mem,
                           const uint64_t *const *instances, const uint64_t *counts, uint64_t *const *d_cols, ssh_air *air_h, int tree_kind,
                           uint32_t n_friendly_layers, int coin_kind, const uint8_t seed[32], ssh_extension_cb cb, void *user, const uint32_t options[5],
                           double *times_out, uint8_t **proof_bytes, uint64_t *proof_len) {
    try {
        if (!ctx || !air_h || !seed || !d_cols) throw std::runtime_error("ssh_prove_files_device: NULL argument");
        const auto t_start = std::chrono::steady_clock::now();
        const TraceJob job = make_trace_job(layout, trace_bin, trace_len, memory_bin, memory_len, rc_min, rc_max, n_steps, segments, mem_addresses, mem_values,
                                            n_mem, instances, counts);
        Air *air = reinterpret_cast<Air *>(air_h);
        if (air->num_base_columns != job.ncols) throw std::runtime_error("ssh_prove_files_device: the AIR is another layout's");
        for (uint32_t c = 0; c < job.ncols; ++c) if (!d_cols[c]) throw std::runtime_error("ssh_prove_files_device: NULL column");
        job.run_device(ctx, d_cols);
        const double gen_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        Claim claim;
        claim.air = air; claim.tree_kind = tree_kind; claim.n_friendly_layers = n_friendly_layers; claim.coin_kind = coin_kind;
        ProofOptions opt;
        if (options) {
            opt.num_queries = options[0]; opt.lde_blowup_factor = options[1]; opt.grinding_factor = options[2];
            opt.fri_folding_factor = options[3]; opt.fri_max_remainder_coeffs = options[4];
        }
        Matrix base;
        base.nrows = job.n;
        for (uint32_t c = 0; c < job.ncols; ++c) base.cols.push_back(d_cols[c]);
        Digest sd;
        memcpy(sd.data(), seed, 32);
        Prover prover(ctx, claim, opt);
        Proof proof = prover.prove(sd, base, [&](const std::vector<Felt> &ch) {
            Matrix ext;
            ext.nrows = base.nrows;
            std::vector<uint64_t> flat(4 * ch.size());
            for (size_t i = 0; i < ch.size(); ++i) memcpy(flat.data() + 4 * i, ch[i].data(), 32);
            std::vector<uint64_t *> cols(air->num_extension_columns, nullptr);
            if (!cb || cb(user, flat.data(), (uint32_t)ch.size(), cols.data()) != 0) throw std::runtime_error("extension callback failed");
            ext.cols = cols;
            return ext;
        });
        if (ss_ctx_sync(ctx) != SS_OK) throw std::runtime_error(ss_last_error());
        if (times_out) { times_out[0] = gen_s; times_out[1] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); }
        if (proof_bytes && proof_len) {
            const std::vector<uint8_t> b = proof.serialize_wire();
            *proof_bytes = (uint8_t *)malloc(b.size());
            memcpy(*proof_bytes, b.data(), b.size());
            *proof_len = b.size();
        }
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}

// ---- the 64-bit field's claim (goldilocks_prover.hpp; cli/src/main.rs:103-133).  options: {num_queries, log_blowup, grinding, fold,
// max_remainder, sha256 (0 / 1)}.  mask: nmask x (column, offset).  ext_cb fills `num_ext` device pointers (the extension trace's
// coordinate columns) for the challenges (3 u64 each); prog_cb returns the lowered composition program for challenges and alpha as a
// blob the callee owns until the next call: [n_instr, n_consts, n_slots, n_tables, d_tables, code (2 n_instr u32 packed in u64
// pairs: one word per u64), consts (3 n_consts), table_desc (2 n_tables)].  The proof comes back as ONE u64 blob (ssh_free):
// [trace_len, pow_nonce, has_ext, n_layers | base_root, ext_root, comp_root (4 u64 each, the 32 bytes) | n_ood, ood_trace... |
// 18 ood_comp | n_rem, remainder... | per layer: root (4), log_len | openings base, ext (if any), comp, then one per layer:
// npos, width, depth, rows (npos x width), paths (npos x depth x 4 u64)].
typedef int (*ssh_gl_extension_cb)(void *user, const uint64_t *challenges, uint32_t nchallenges, uint64_t **d_cols_out);
typedef int (*ssh_gl_program_cb)(void *user, const uint64_t *challenges, uint32_t nchallenges, const uint64_t *alpha, const uint64_t **blob_out,
                                 uint64_t *blob_len);
int ssh_gl_prove(ss_ctx *ctx, const uint32_t options[6], const uint8_t seed[32], const uint8_t statement_digest[32], const uint64_t *const *d_base,
                 uint32_t nbase, uint64_t n, const uint32_t *mask, uint32_t nmask, uint32_t num_challenges, uint32_t num_ext, ssh_gl_extension_cb ext_cb,
                 ssh_gl_program_cb prog_cb, void *user, uint64_t **proof_blob, uint64_t *proof_len) {
    try {
        if (!ctx || !options || !seed || !statement_digest || !d_base || !mask || !prog_cb || !proof_blob || !proof_len) throw std::runtime_error("ssh_gl_prove: NULL argument");
        gl::Options opt;
        opt.num_queries = options[0]; opt.log_blowup = options[1]; opt.grinding = options[2]; opt.fold = options[3]; opt.max_remainder = options[4];
        opt.sha256 = options[5] != 0;
        Digest sd, st;
        memcpy(sd.data(), seed, 32);
        memcpy(st.data(), statement_digest, 32);
        std::vector<const uint64_t *> base(d_base, d_base + nbase);
        std::vector<std::pair<uint32_t, uint32_t>> cells;
        for (uint32_t j = 0; j < nmask; ++j) cells.push_back({mask[2 * j], mask[2 * j + 1]});
        auto flat = [](const std::vector<gl::Fq3> &v) { std::vector<uint64_t> o; for (auto &c : v) o.insert(o.end(), c.begin(), c.end()); return o; };
        const gl::Proof p = gl::prove(ctx, opt, sd, st, base, n, cells, num_challenges, num_ext,
            [&](const std::vector<gl::Fq3> &ch) {
                std::vector<uint64_t *> cols(num_ext, nullptr);
                const std::vector<uint64_t> f = flat(ch);
                if (!ext_cb || ext_cb(user, f.data(), (uint32_t)ch.size(), cols.data()) != 0) throw std::runtime_error("extension callback failed");
                return std::vector<const uint64_t *>(cols.begin(), cols.end());
            },
            [&](const std::vector<gl::Fq3> &ch, const gl::Fq3 &alpha) {
                const uint64_t *blob = nullptr;
                uint64_t len = 0;
                const std::vector<uint64_t> f = flat(ch);
                if (prog_cb(user, f.data(), (uint32_t)ch.size(), alpha.data(), &blob, &len) != 0 || !blob || len < 5) throw std::runtime_error("program callback failed");
                gl::ProgramData pd;
                const uint64_t n_instr = blob[0], n_consts = blob[1], n_tables = blob[3];
                if (len != 5 + 2 * n_instr + 3 * n_consts + 2 * n_tables) throw std::runtime_error("program callback: blob length");
                pd.n_slots = (uint32_t)blob[2];
                pd.d_tables = reinterpret_cast<const uint64_t *>((uintptr_t)blob[4]);
                const uint64_t *q = blob + 5;
                for (uint64_t i = 0; i < 2 * n_instr; ++i) pd.code.push_back((uint32_t)*q++);
                pd.consts.assign(q, q + 3 * n_consts); q += 3 * n_consts;
                for (uint64_t i = 0; i < 2 * n_tables; ++i) pd.table_desc.push_back((uint32_t)*q++);
                return pd;
            });
        std::vector<uint64_t> out{p.trace_len, p.pow_nonce, p.has_ext ? 1ull : 0ull, p.fri_layers.size()};
        auto digest = [&](const Digest &d) { uint64_t w[4]; memcpy(w, d.data(), 32); out.insert(out.end(), w, w + 4); };
        digest(p.base_root); digest(p.ext_root); digest(p.comp_root);
        out.push_back(p.ood_trace.size() / 3); out.insert(out.end(), p.ood_trace.begin(), p.ood_trace.end());
        out.insert(out.end(), p.ood_comp.begin(), p.ood_comp.end());
        out.push_back(p.remainder.size() / 3); out.insert(out.end(), p.remainder.begin(), p.remainder.end());
        for (auto &fl : p.fri_layers) { digest(fl.root); out.push_back(fl.log_len); }
        auto opening = [&](const gl::Opening &o) {
            const uint64_t npos = o.width ? o.rows.size() / o.width : 0;
            out.push_back(npos); out.push_back(o.width); out.push_back(o.depth);
            out.insert(out.end(), o.rows.begin(), o.rows.end());
            const size_t at = out.size();
            out.resize(at + o.paths.size() / 8);
            memcpy(out.data() + at, o.paths.data(), o.paths.size());
        };
        opening(p.base);
        if (p.has_ext) opening(p.ext);
        opening(p.comp);
        for (auto &fl : p.fri_layers) opening(fl.opening);
        *proof_blob = (uint64_t *)malloc(out.size() * 8);
        memcpy(*proof_blob, out.data(), out.size() * 8);
        *proof_len = out.size();
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}

// the lowered composition program for these challenges and its table descriptions, as one u64 blob:
// n_instr, code words..., n_consts, 4 limbs each..., n_slots, then layout_air_tables()
int ssh_air_dump(ssh_air *air_h, uint64_t n, const uint64_t *challenges, uint32_t nchallenges, const uint64_t alpha[4], uint64_t **blob, uint64_t *blob_len) {
    try {
        Air *air = reinterpret_cast<Air *>(air_h);
        std::vector<Felt> ch(nchallenges);
        for (uint32_t i = 0; i < nchallenges; ++i) memcpy(ch[i].data(), challenges + 4 * i, 32);
        Felt a;
        memcpy(a.data(), alpha, 32);
        const AirProgramData pd = air->build_program(n, ch, a);
        std::vector<uint64_t> out{pd.program.n_instr()};
        for (uint32_t w : pd.program.code) out.push_back(w);
        out.push_back(pd.program.consts.size());
        for (auto &c : pd.program.consts) for (int k = 0; k < 4; ++k) out.push_back(c[k]);
        out.push_back(pd.program.n_slots);
        const std::vector<uint64_t> t = layout_air_tables(*air);
        out.insert(out.end(), t.begin(), t.end());
        *blob = (uint64_t *)malloc(out.size() * 8);
        memcpy(*blob, out.data(), out.size() * 8);
        *blob_len = out.size();
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}

// What a prover needs from an AIR handle per proof, for callers that drive the kernels themselves (the sharded prover,
// sandstorm_amd/sharded_prover.py): the lowered composition program for these challenges, the device address of the tables
// and their descriptions.  One u64 blob: n_instr, code words..., n_consts, 4 limbs each..., n_slots, n_tables,
// (offset, log2 length) per table, device address of the tables.
int ssh_air_program(ssh_air *air_h, uint64_t n, const uint64_t *challenges, uint32_t nchallenges, const uint64_t alpha[4], uint64_t **blob,
                    uint64_t *blob_len) {
    try {
        Air *air = reinterpret_cast<Air *>(air_h);
        std::vector<Felt> ch(nchallenges);
        for (uint32_t i = 0; i < nchallenges; ++i) memcpy(ch[i].data(), challenges + 4 * i, 32);
        Felt a;
        memcpy(a.data(), alpha, 32);
        const AirProgramData pd = air->build_program(n, ch, a);
        std::vector<uint64_t> out{pd.program.n_instr()};
        for (uint32_t w : pd.program.code) out.push_back(w);
        out.push_back(pd.program.consts.size());
        for (auto &c : pd.program.consts) for (int k = 0; k < 4; ++k) out.push_back(c[k]);
        out.push_back(pd.program.n_slots);
        out.push_back(pd.table_desc.size() / 2);
        for (uint32_t v : pd.table_desc) out.push_back(v);
        out.push_back((uint64_t)(uintptr_t)pd.d_tables);
        *blob = (uint64_t *)malloc(out.size() * 8);
        memcpy(*blob, out.data(), out.size() * 8);
        *blob_len = out.size();
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}
// Air::prepare_program: the program for these challenges lowered ahead of its composition coefficient (the next ssh_air_program /
// proof with the same challenges only patches the coefficient's powers in); a test hook - the provers call it themselves
int ssh_air_prepare_program(ssh_air *air_h, uint64_t n, const uint64_t *challenges, uint32_t nchallenges) {
    try {
        Air *air = reinterpret_cast<Air *>(air_h);
        std::vector<Felt> ch(nchallenges);
        for (uint32_t i = 0; i < nchallenges; ++i) memcpy(ch[i].data(), challenges + 4 * i, 32);
        air->prepare_program(n, ch);
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}
// the mask (trace_arguments(): sorted (column, row offset) cells): cols_out / offs_out have room for ssh_air_columns(air, 2) entries
int ssh_air_mask(ssh_air *air_h, uint32_t *cols_out, uint32_t *offs_out) {
    Air *air = reinterpret_cast<Air *>(air_h);
    for (size_t j = 0; j < air->mask.size(); ++j) { cols_out[j] = air->mask[j].first; offs_out[j] = air->mask[j].second; }
    return 0;
}
uint32_t ssh_air_num_challenges(ssh_air *air_h) { return reinterpret_cast<Air *>(air_h)->num_challenges; }

// Verify a proof in the reference's wire format (verifier.hpp) against an AIR handle (mini or recursive; the handle may
// have been created without a device).  conventions: 1 = the shipped proofs' (bit-reversed, unnormalised fold, unshifted
// remainder) with the bare draw as FRI challenge (round-1 proofs), 2 = the reference's (the FRI challenge is the draw times
// the layer offset), 0 = the older path's.  required_security_bits: cli/src/main.rs:66-67 (default 80 there).
// expected_options (nullable): the five ProofOptions the proof must carry.  positions_out (nullable): room for num_queries
// values; *n_positions is set.  n_friendly_layers: the N of FriendlyMerkleTree<N, _> (src/claims.rs:10: 22), read for
// SS_TREE_FRIENDLY only - the value the proof was made with (ssh_prove*'s argument of the same name).
int ssh_verify(ssh_air *air_h, int tree_kind, int coin_kind, const uint8_t seed[32], const uint8_t *proof, uint64_t proof_len,
               int conventions, uint32_t required_security_bits, const uint32_t *expected_options, uint64_t *positions_out,
               uint32_t *n_positions, uint32_t n_friendly_layers) {
    try {
        Air *air = reinterpret_cast<Air *>(air_h);
        Digest sd;
        memcpy(sd.data(), seed, 32);
        Conventions conv;
        if (!conventions) { conv.bitrev_commit = false; conv.fri_unnormalised = false; conv.remainder_unshifted = false; }
        conv.fri_alpha_times_offset = conventions == 2;       // 2: the shipped conventions plus the reference's FRI challenge scaling
        const WireProof w = parse_wire(proof, proof_len, tree_kind);
        ProofOptions exp;
        if (expected_options) {
            exp.num_queries = expected_options[0]; exp.lde_blowup_factor = expected_options[1]; exp.grinding_factor = expected_options[2];
            exp.fri_folding_factor = expected_options[3]; exp.fri_max_remainder_coeffs = expected_options[4];
        }
        const std::vector<uint64_t> pos = verify(w, *air, tree_kind, coin_kind, sd, conv, required_security_bits, expected_options ? &exp : nullptr, n_friendly_layers);
        if (positions_out && n_positions) { memcpy(positions_out, pos.data(), pos.size() * 8); *n_positions = (uint32_t)pos.size(); }
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}

// ---- the C++ coin, exposed for the CPU tests (tests/test_host_cpp.py)
typedef struct ssh_coin ssh_coin;
ssh_coin *ssh_coin_new(int kind, const uint8_t seed[32]) {
    Digest d;
    memcpy(d.data(), seed, 32);
    return reinterpret_cast<ssh_coin *>(new PublicCoin(kind, d));
}
void ssh_coin_free(ssh_coin *c) { delete reinterpret_cast<PublicCoin *>(c); }
static std::vector<Felt> felts_of(const uint64_t *v, uint32_t n) {
    std::vector<Felt> o(n);
    for (uint32_t i = 0; i < n; ++i) memcpy(o[i].data(), v + 4 * i, 32);
    return o;
}
int ssh_coin_op(ssh_coin *ch, int op, const uint8_t *bytes, uint64_t len, const uint64_t *felts, uint32_t nfelts, uint64_t arg,
                uint64_t *out, uint32_t *nout) {
    try {
        PublicCoin *c = reinterpret_cast<PublicCoin *>(ch);
        switch (op) {
        case 0: c->reseed_with_bytes(bytes, len); break;
        case 1: c->reseed_with_field_elements(felts_of(felts, nfelts)); break;
        case 2: c->reseed_with_field_element_vector(felts_of(felts, nfelts)); break;
        case 3: c->reseed_with_int(arg); break;
        case 4: { Felt f = c->draw(); memcpy(out, f.data(), 32); break; }
        case 5: { auto q = c->draw_queries((size_t)arg, len); memcpy(out, q.data(), 8 * q.size()); *nout = (uint32_t)q.size(); break; }
        case 6: memcpy(out, c->digest().data(), 32); out[4] = c->counter(); break;
        default: throw std::runtime_error("bad coin op");
        }
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}

}  // extern "C"
