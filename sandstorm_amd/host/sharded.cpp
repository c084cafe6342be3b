// sharded.cpp — one proof over several GPUs (sharded.hpp).  The mirror of sandstorm_amd/sharded_prover.py, stage by stage:
//
//   LDE of trace columns (N1/N2)          by COLUMN: column c on rank c % R            no exchange
//   row hashing (H1), constraints (Q1),   by ROW BLOCK: LDE rows [r N/R, (r+1) N/R)    owners send every rank its block (+ the rows
//   DEEP (D1)                             of every column + the halo behind them        behind it the constraints reach, wrapping)
//   Merkle sub-trees (H3/H4)              by LEAF BLOCK: leaf i is row bitrev(i)        digests stored at their LOCAL bit-reversed slot:
//                                                                                       chunk p of them is rank p's, one equal-split
//                                                                                       exchange, a stride-R comb on arrival; the R
//                                                                                       sub-tree roots all-gathered, top levels on hosts
//   extension columns, composition        ONE transform over the R ranks                owner scatters n / R blocks; per transform two
//   polynomial (Q2), DEEP's extension     (ss_ntt_shard_fp252: local stages on blocks,  equal-split all-to-alls (block <-> exchanged
//                                         the log2 R cross stages on the exchanged      layout); the halo rows from the next ranks
//                                         layout): every rank 1 / R of every transform
//   out-of-domain values                  base columns: by column owner; distributed    all-gather of the values / of R partial sums
//                                         coefficient blocks: P(y) = sum_r y^j(r) P_r(y^R)  per cell, combined on every host
//   FRI (F1): layers above 2^21 values    rank m holds the fold rows [m rows/R, ...)    block layout -> row layout: one all-to-all of
//                                         with all `fold` entries: folds and commits     1 / R of the layer; then the layer (<= 2^21)
//                                         its rows (sub-tree + digest routing as above)  gathered to rank 0, which finishes + grinds
//   openings                              rows / paths where they live                  gathered to rank 0 (kilobytes)
//
// Nothing is a sum over ranks: no all-reduce.  The coin runs on every rank in lock step up to the last distributed FRI layer.
#include "sharded.hpp"

#include <algorithm>
#include <chrono>
#include <string>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <stdexcept>

namespace ssh {

static void ok(ss_status s) {
    if (s != SS_OK) throw std::runtime_error(ss_last_error());
}
static uint32_t log2u(uint64_t v) {
    uint32_t l = 0;
    while ((1ull << l) < v) ++l;
    if ((1ull << l) != v) throw std::runtime_error("not a power of two");
    return l;
}
static uint64_t brev(uint64_t x, uint32_t bits) { uint64_t r = 0; for (uint32_t i = 0; i < bits; ++i) r |= ((x >> i) & 1ull) << (bits - 1 - i); return r; }

// ---------------------------------------------------------------------------------------------------------- transports
std::vector<std::vector<uint8_t>> Transport::all_gather_var(ss_ctx *ctx, const std::vector<uint8_t> &mine) {
    std::vector<uint8_t> len(8);
    const uint64_t n = mine.size();
    memcpy(len.data(), &n, 8);
    const std::vector<uint8_t> lens = all_gather(ctx, len);
    uint64_t most = 0;
    std::vector<uint64_t> ln(world);
    for (uint32_t p = 0; p < world; ++p) { memcpy(&ln[p], lens.data() + 8 * p, 8); most = std::max(most, ln[p]); }
    std::vector<std::vector<uint8_t>> out(world);
    if (!most) return out;
    std::vector<uint8_t> padded(mine);
    padded.resize(most, 0);
    const std::vector<uint8_t> all = all_gather(ctx, padded);
    for (uint32_t p = 0; p < world; ++p) out[p].assign(all.begin() + p * most, all.begin() + p * most + ln[p]);
    return out;
}

namespace {
// RCCL over the C ABI (ss_comm_*: grouped ncclSend / ncclRecv, ncclAllGather, on the context's stream)
class RcclTransport : public Transport {
public:
    RcclTransport(ss_ctx *ctx, const uint8_t id[128], uint32_t rank_, uint32_t world_) {
        rank = rank_; world = world_;
        ok(ss_comm_create(ctx, id, rank, world, &comm_));
    }
    ~RcclTransport() override { stage_.reset(); ss_comm_destroy(comm_); }
    void exchange(ss_ctx *, const std::vector<Message> &sends, const std::vector<Message> &recvs) override {
        std::vector<uint32_t> sp, rp;
        std::vector<const void *> sb;
        std::vector<void *> rb;
        std::vector<uint64_t> sl, rl;
        for (const Message &m : sends) { sp.push_back(m.peer); sb.push_back(m.ptr); sl.push_back(m.bytes); }
        for (const Message &m : recvs) { rp.push_back(m.peer); rb.push_back(m.ptr); rl.push_back(m.bytes); }
        ok(ss_comm_exchange(comm_, (uint32_t)sends.size(), sp.data(), sb.data(), sl.data(), (uint32_t)recvs.size(), rp.data(), rb.data(), rl.data()));
    }
    std::vector<uint8_t> all_gather(ss_ctx *ctx, const std::vector<uint8_t> &mine) override {
        const uint64_t n = mine.size();
        std::vector<uint8_t> out(n * world);
        if (!n) return out;
        // the host all-gathers are small and many (roots, out-of-domain values, openings): one staging buffer kept by the transport
        const size_t in_bytes = (n + 255) / 256 * 256, need = in_bytes + n * world;
        if (!stage_ || stage_->bytes() < need) stage_.reset(new DeviceBuffer(ctx, std::max<size_t>(need, 1 << 16)));
        uint8_t *d_in = stage_->u8(), *d_out = d_in + in_bytes;
        ok(ss_upload(ctx, d_in, mine.data(), n));
        ok(ss_comm_all_gather(comm_, d_in, n, d_out));
        ok(ss_download(ctx, out.data(), d_out, n * world));
        return out;
    }
private:
    ss_comm *comm_ = nullptr;
    std::unique_ptr<DeviceBuffer> stage_;
};
}  // namespace

std::unique_ptr<Transport> make_rccl_transport(ss_ctx *ctx, const uint8_t id[128], uint32_t rank, uint32_t world) {
    return std::unique_ptr<Transport>(new RcclTransport(ctx, id, rank, world));
}

namespace {
// The caller's collectives over host memory (TransportCallbacks): an exchange step is ONE all-to-all of bytes.  The messages for a
// peer are packed in list order - the order the receiver lists its receives from this rank in -, downloaded into one host buffer,
// exchanged, and uploaded to where the receives point.  A message to this rank itself is a device copy.
class CallbackTransport : public Transport {
public:
    CallbackTransport(const TransportCallbacks &cb, uint32_t rank_, uint32_t world_) : cb_(cb) {
        if (!cb.all_to_all || !cb.all_gather) throw std::runtime_error("callback transport: both callbacks are needed");
        if (!world_ || rank_ >= world_) throw std::runtime_error("callback transport: rank / world");
        rank = rank_; world = world_;
    }
    void exchange(ss_ctx *ctx, const std::vector<Message> &sends, const std::vector<Message> &recvs) override {
        std::vector<uint64_t> sb(world, 0), rb(world, 0), so(world + 1, 0), ro(world + 1, 0);
        for (const Message &m : sends) { if (m.peer >= world) throw std::runtime_error("callback transport: peer"); if (m.peer != rank) sb[m.peer] += m.bytes; }
        for (const Message &m : recvs) { if (m.peer >= world) throw std::runtime_error("callback transport: peer"); if (m.peer != rank) rb[m.peer] += m.bytes; }
        for (uint32_t p = 0; p < world; ++p) { so[p + 1] = so[p] + sb[p]; ro[p + 1] = ro[p] + rb[p]; }
        send_.resize(so[world]);
        recv_.resize(ro[world]);
        std::vector<uint64_t> at(so.begin(), so.end() - 1);
        for (const Message &m : sends)
            if (m.peer != rank && m.bytes) { ok(ss_download(ctx, send_.data() + at[m.peer], m.ptr, m.bytes)); at[m.peer] += m.bytes; }
        // this rank's messages to itself: matched in list order, copied on the device
        size_t k = 0;
        for (const Message &rv : recvs) {
            if (rv.peer != rank) continue;
            while (k < sends.size() && sends[k].peer != rank) ++k;
            if (k == sends.size() || sends[k].bytes != rv.bytes) throw std::runtime_error("callback transport: a receive without its matching send");
            ok(ss_dev_copy(ctx, rv.ptr, sends[k].ptr, rv.bytes));
            ++k;
        }
        if (cb_.all_to_all(cb_.user, send_.data(), sb.data(), recv_.data(), rb.data()) != 0) throw std::runtime_error("callback transport: all_to_all failed");
        at.assign(ro.begin(), ro.end() - 1);
        for (const Message &m : recvs)
            if (m.peer != rank && m.bytes) { ok(ss_upload(ctx, m.ptr, recv_.data() + at[m.peer], m.bytes)); at[m.peer] += m.bytes; }
        ok(ss_ctx_sync(ctx));                                        // the staging buffers are reused by the next step
    }
    std::vector<uint8_t> all_gather(ss_ctx *, const std::vector<uint8_t> &mine) override {
        std::vector<uint8_t> out(mine.size() * world);
        if (mine.empty()) return out;
        if (cb_.all_gather(cb_.user, mine.data(), mine.size(), out.data()) != 0) throw std::runtime_error("callback transport: all_gather failed");
        return out;
    }
private:
    TransportCallbacks cb_;
    std::vector<uint8_t> send_, recv_;
};
}  // namespace
std::unique_ptr<Transport> make_callback_transport(const TransportCallbacks &cb, uint32_t rank, uint32_t world) {
    return std::unique_ptr<Transport>(new CallbackTransport(cb, rank, world));
}

// Ranks as threads of one process: a step publishes what a rank offers, a barrier, every rank copies what is addressed to it
// (device -> device on its own stream: the contexts share an address space), a barrier.
class LocalGroup {
public:
    explicit LocalGroup(uint32_t world) : world_(world), sends_(world), bytes_(world) {}
    void barrier() {
        std::unique_lock<std::mutex> lk(m_);
        if (failed_) throw std::runtime_error("local transport: another rank failed");
        const uint64_t gen = gen_;
        if (++waiting_ == world_) { waiting_ = 0; ++gen_; cv_.notify_all(); }
        else cv_.wait(lk, [&] { return gen_ != gen || failed_; });
        if (failed_) throw std::runtime_error("local transport: another rank failed");
    }
    void fail() {                                // a rank that throws releases the others instead of leaving them at a barrier
        std::lock_guard<std::mutex> lk(m_);
        failed_ = true;
        cv_.notify_all();
    }
    uint32_t world_;
    std::vector<const std::vector<Message> *> sends_;
    std::vector<const std::vector<uint8_t> *> bytes_;
private:
    std::mutex m_;
    std::condition_variable cv_;
    uint32_t waiting_ = 0;
    uint64_t gen_ = 0;
    bool failed_ = false;
};
std::shared_ptr<LocalGroup> make_local_group(uint32_t world) { return std::make_shared<LocalGroup>(world); }
void local_group_fail(LocalGroup &g) { g.fail(); }

namespace {
class LocalTransport : public Transport {
public:
    LocalTransport(std::shared_ptr<LocalGroup> g, uint32_t rank_) : g_(std::move(g)) { rank = rank_; world = g_->world_; }
    void exchange(ss_ctx *ctx, const std::vector<Message> &sends, const std::vector<Message> &recvs) override {
        ok(ss_ctx_sync(ctx));                            // what this rank offers is complete
        g_->sends_[rank] = &sends;
        g_->barrier();
        std::vector<size_t> next(world, 0);              // per source: the next of its messages addressed to this rank
        for (const Message &rv : recvs) {
            const std::vector<Message> &from = *g_->sends_[rv.peer];
            size_t &k = next[rv.peer];
            while (k < from.size() && from[k].peer != rank) ++k;
            if (k == from.size() || from[k].bytes != rv.bytes) throw std::runtime_error("local transport: a receive without its matching send");
            ok(ss_dev_copy(ctx, rv.ptr, from[k].ptr, rv.bytes));
            ++k;
        }
        ok(ss_ctx_sync(ctx));                            // the copies are done before anybody reuses a send buffer
        g_->barrier();
    }
    std::vector<uint8_t> all_gather(ss_ctx *, const std::vector<uint8_t> &mine) override {
        g_->bytes_[rank] = &mine;
        g_->barrier();
        std::vector<uint8_t> out;
        for (uint32_t p = 0; p < world; ++p) {
            if (g_->bytes_[p]->size() != mine.size()) throw std::runtime_error("local transport: all_gather of unequal sizes");
            out.insert(out.end(), g_->bytes_[p]->begin(), g_->bytes_[p]->end());
        }
        g_->barrier();
        return out;
    }
private:
    std::shared_ptr<LocalGroup> g_;
};
}  // namespace
std::unique_ptr<Transport> make_local_transport(std::shared_ptr<LocalGroup> group, uint32_t rank) {
    return std::unique_ptr<Transport>(new LocalTransport(std::move(group), rank));
}

// ------------------------------------------------------------------------------------------------ self check
void transport_self_check(ss_ctx *ctx, Transport &comm, uint64_t bandwidth_bytes, double *gbps) {
    const uint32_t R = comm.world, r = comm.rank;
    auto len = [](uint32_t s, uint32_t d, uint32_t k) -> uint64_t { return 32ull * (1 + (7 * s + 3 * d + 11 * k) % 13) * (k ? 9 : 1); };
    auto byte = [](uint32_t s, uint32_t d, uint32_t k, uint64_t i) -> uint8_t { return (uint8_t)(131 * s + 17 * d + 5 * k + i + (i >> 8)); };
    // a rank that sees a wrong byte still enters every step (the others are in them); the verdicts are gathered at the end and
    // EVERY rank throws the same message
    std::string bad;
    auto complain = [&](const std::string &what) { if (bad.empty()) bad = "rank " + std::to_string(r) + " " + what; };
    uint64_t stot = 0, rtot = 0;
    for (uint32_t p = 0; p < R; ++p) for (uint32_t k = 0; k < 2; ++k) { stot += len(r, p, k); rtot += len(p, r, k); }
    std::vector<uint8_t> hs(stot), hr(rtot);
    DeviceBuffer ds(ctx, stot), dr(ctx, rtot);
    std::vector<Message> sends, recvs;
    uint64_t so = 0, ro = 0;
    for (uint32_t k = 0; k < 2; ++k)                         // message 0 of every pair, then message 1: the lists interleave the peers
        for (uint32_t p = 0; p < R; ++p) {
            const uint64_t ls = len(r, p, k), lr = len(p, r, k);
            for (uint64_t i = 0; i < ls; ++i) hs[so + i] = byte(r, p, k, i);
            sends.push_back({p, ds.u8() + so, ls});
            recvs.push_back({p, dr.u8() + ro, lr});
            so += ls; ro += lr;
        }
    ok(ss_upload(ctx, ds.u8(), hs.data(), stot));
    ok(ss_dev_zero(ctx, dr.u8(), rtot));
    comm.exchange(ctx, sends, recvs);
    ok(ss_download(ctx, hr.data(), dr.u8(), rtot));
    ro = 0;
    for (uint32_t k = 0; k < 2; ++k)
        for (uint32_t p = 0; p < R; ++p) {
            const uint64_t lr = len(p, r, k);
            for (uint64_t i = 0; i < lr; ++i)
                if (hr[ro + i] != byte(p, r, k, i)) {
                    complain("got a wrong byte in message " + std::to_string(k) + " from rank " + std::to_string(p) + " (offset " + std::to_string(i) + " of " +
                             std::to_string(lr) + ")");
                    break;
                }
            ro += lr;
        }
    std::vector<uint8_t> mine(33);
    for (size_t i = 0; i < mine.size(); ++i) mine[i] = byte(r, 0, 2, i);
    const std::vector<uint8_t> all = comm.all_gather(ctx, mine);
    if (all.size() != mine.size() * R) complain("got an all_gather of the wrong size");
    else
        for (uint32_t p = 0; p < R; ++p)
            for (size_t i = 0; i < mine.size(); ++i)
                if (all[p * mine.size() + i] != byte(p, 0, 2, i)) { complain("got a wrong all_gather slot " + std::to_string(p)); break; }
    std::vector<uint8_t> var(5 * r);                         // rank 0 contributes nothing
    for (size_t i = 0; i < var.size(); ++i) var[i] = byte(r, 1, 3, i);
    const std::vector<std::vector<uint8_t>> parts = comm.all_gather_var(ctx, var);
    for (uint32_t p = 0; p < R; ++p) {
        if (parts[p].size() != 5 * p) { complain("got an all_gather_var slot of the wrong length from rank " + std::to_string(p)); continue; }
        for (size_t i = 0; i < parts[p].size(); ++i) if (parts[p][i] != byte(p, 1, 3, i)) { complain("got wrong all_gather_var bytes from rank " + std::to_string(p)); break; }
    }
    std::string verdict;
    for (const std::vector<uint8_t> &v : comm.all_gather_var(ctx, std::vector<uint8_t>(bad.begin(), bad.end())))
        if (!v.empty()) verdict += (verdict.empty() ? "" : "; ") + std::string(v.begin(), v.end());
    if (!verdict.empty()) throw std::runtime_error("transport self check: " + verdict);
    if (gbps) *gbps = 0.0;
    if (bandwidth_bytes && R > 1) {
        const uint64_t per = bandwidth_bytes / 32 * 32;
        DeviceBuffer bs(ctx, per * R), br(ctx, per * R);
        ok(ss_dev_zero(ctx, bs.u8(), per * R));
        std::vector<Message> s2, r2;
        for (uint32_t p = 0; p < R; ++p) if (p != r) { s2.push_back({p, bs.u8() + per * p, per}); r2.push_back({p, br.u8() + per * p, per}); }
        comm.exchange(ctx, s2, r2);                          // untimed: connections come up at the first use of a pair
        ok(ss_ctx_sync(ctx));
        comm.all_gather(ctx, std::vector<uint8_t>(8, 0));    // every rank starts the timed step together
        const auto t0 = std::chrono::steady_clock::now();
        comm.exchange(ctx, s2, r2);
        ok(ss_ctx_sync(ctx));
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (gbps && dt > 0) *gbps = 2.0 * per * (R - 1) / dt / 1e9;
    }
}

// ------------------------------------------------------------------------------------------------ top levels on the host
std::array<uint8_t, 33> merge_nodes(int tree_kind, uint32_t n_friendly_layers, uint32_t depth, const std::array<uint8_t, 33> &left,
                                    const std::array<uint8_t, 33> &right) {
    std::array<uint8_t, 33> out{};
    uint8_t cat[64];
    memcpy(cat, left.data(), 32);
    memcpy(cat + 32, right.data(), 32);
    if (tree_kind == SS_TREE_KECCAK || tree_kind == SS_TREE_KECCAK_M20) {
        const Digest d = keccak256(cat, 64);
        memcpy(out.data(), d.data(), tree_kind == SS_TREE_KECCAK ? 32 : 20);       // masked: the first 20 bytes (hash/mod.rs:5-13)
        return out;
    }
    if (tree_kind != SS_TREE_FRIENDLY) throw std::runtime_error("merge_nodes: tree kind");
    if (depth < n_friendly_layers) {        // Pedersen; hash_boundary reads Blake2s digests as big-endian integers (mixed.rs:148-155)
        auto felt_of_be = [](const uint8_t *d) {
            uint64_t c[4];
            for (int k = 0; k < 4; ++k) { c[k] = 0; for (int j = 0; j < 8; ++j) c[k] |= (uint64_t)d[31 - (8 * k + j)] << (8 * j); }
            static const uint64_t P[4] = {1ull, 0ull, 0ull, 0x0800000000000011ull};
            auto geq = [&]() { for (int i = 3; i >= 0; --i) { if (c[i] != P[i]) return c[i] > P[i]; } return true; };
            while (geq()) { unsigned __int128 br = 0; for (int i = 0; i < 4; ++i) { unsigned __int128 t = (unsigned __int128)c[i] - P[i] - (uint64_t)br; c[i] = (uint64_t)t; br = (t >> 64) & 1; } }
            return felt_from_canonical(Felt{c[0], c[1], c[2], c[3]});
        };
        const Felt a = felt_of_be(left.data()), b = felt_of_be(right.data());
        Felt h;
        ok(ss_pedersen_hash_host(a.data(), b.data(), h.data()));
        const auto be = canonical_be_bytes(h);
        memcpy(out.data(), be.data(), 32);
        out[32] = 0;
        return out;
    }
    if (left[32] != 1 || right[32] != 1) throw std::runtime_error("merge_nodes: a Blake2s level above a Pedersen one");
    const Digest d = blake2s256(cat, 64);
    memcpy(out.data() + 12, d.data() + 12, 20);                                    // masked: the last 20 bytes (hash/mod.rs:15-23)
    out[32] = 1;
    return out;
}

// ---------------------------------------------------------------------------------------------------------- the prover
namespace {
struct Writer {
    std::vector<uint8_t> b;
    void u64(uint64_t v) { const uint8_t *p = (const uint8_t *)&v; b.insert(b.end(), p, p + 8); }
    void raw(const void *p, size_t n) { const uint8_t *q = (const uint8_t *)p; b.insert(b.end(), q, q + n); }
};
struct Reader {
    const std::vector<uint8_t> &b;
    size_t o = 0;
    uint64_t u64() { uint64_t v; memcpy(&v, b.data() + o, 8); o += 8; return v; }
    const uint8_t *take(size_t n) { const uint8_t *p = b.data() + o; o += n; return p; }
    bool done() const { return o >= b.size(); }
};
std::vector<uint64_t> flat(const std::vector<Felt> &v) {
    std::vector<uint64_t> o(4 * v.size());
    for (size_t i = 0; i < v.size(); ++i) memcpy(o.data() + 4 * i, v[i].data(), 32);
    return o;
}
Digest digest_of(const std::array<uint8_t, 33> &r) { Digest d; memcpy(d.data(), r.data(), 32); return d; }
}  // namespace

struct ShardedProver::Commitment {
    Buf leaves_own;                          // this rank's leaf block: B digests (or elements)
    uint8_t *leaves = nullptr;
    int leaf_kind = SS_LEAF_DIGEST;
    Buf nodes, tags;                         // the sub-tree
    std::vector<std::vector<std::array<uint8_t, 33>>> top;      // top levels, root first, down to the R sub-tree roots
    std::array<uint8_t, 33> root{};
};

// column blocks -> row blocks: LDE rows (r B + k) mod N, k < B + halo, of ALL ncols columns
std::vector<ShardedProver::Buf> ShardedProver::to_row_blocks(const std::map<uint32_t, Buf> &owned, uint32_t ncols, uint32_t first_col,
                                                             uint64_t N, uint64_t halo) {
    const uint32_t R = comm_.world, r = comm_.rank;
    const uint64_t B = N / R, len = B + halo;
    std::vector<Buf> out;
    if (R == 1) {                                // a group of one: the columns are their own row blocks
        for (uint32_t c = first_col; c < first_col + ncols; ++c) out.push_back(owned.at(c));
        return out;
    }
    std::vector<Message> sends, recvs;
    auto pieces = [&](uint32_t p, uint64_t &len1, uint64_t &len2) {          // rows [p B, p B + len): up to the end, then from row 0
        const uint64_t lo = p * B, hi = lo + len;
        len1 = hi <= N ? len : N - lo;
        len2 = len - len1;
    };
    for (uint32_t c = first_col; c < first_col + ncols; ++c) {
        const uint32_t o = owner(c);
        Buf mine = std::make_shared<DeviceBuffer>(ctx_, 32 * len);
        uint64_t l1, l2;
        if (o == r) {
            uint8_t *col = owned.at(c)->u8();
            for (uint32_t p = 0; p < R; ++p) {
                pieces(p, l1, l2);
                if (p == r) {
                    ok(ss_dev_copy(ctx_, mine->u8(), col + 32 * p * B, 32 * l1));
                    if (l2) ok(ss_dev_copy(ctx_, mine->u8() + 32 * l1, col, 32 * l2));
                } else {
                    sends.push_back({p, col + 32 * p * B, 32 * l1});
                    if (l2) sends.push_back({p, col, 32 * l2});
                }
            }
        } else {
            pieces(r, l1, l2);
            recvs.push_back({o, mine->u8(), 32 * l1});
            if (l2) recvs.push_back({o, mine->u8() + 32 * l1, 32 * l2});
        }
        out.push_back(mine);
    }
    comm_.exchange(ctx_, sends, recvs);
    return out;
}

std::unique_ptr<ShardedProver::Commitment> ShardedProver::commit(const std::vector<Buf> &blocks, uint64_t N, int order) {
    std::vector<const uint64_t *> cols;
    for (const Buf &b : blocks) cols.push_back(b->u64());
    return commit(cols, N, order);
}
std::unique_ptr<ShardedProver::Commitment> ShardedProver::commit(const std::vector<const uint64_t *> &blocks, uint64_t N, int order) {
    const uint32_t R = comm_.world;
    const uint64_t B = N / R;
    const uint32_t log_R = log2u(R), log_B = log2u(B);
    const bool single = blocks.size() == 1, friendly = claim_.tree_kind == SS_TREE_FRIENDLY;
    auto com = std::unique_ptr<Commitment>(new Commitment);
    com->leaf_kind = single ? SS_LEAF_FELT : SS_LEAF_DIGEST;
    // Row i = r B + k is leaf bitrev(i) (or i).  bitrev_{log N}(i) = bitrev_{log B}(k) << log R | bitrev_{log R}(r): with this rank's
    // digests at their LOCAL bit-reversed slot - which the row-hash kernel's scatter does for free - chunk p of them is exactly the
    // leaves rank p owns: one equal-split exchange (32 B per row), after which the chunk from rank s is the stride-R comb at
    // offset bitrev_{log R}(s) of the leaf block.
    Buf mine = std::make_shared<DeviceBuffer>(ctx_, 32 * B);
    if (R > 1 && order == SS_ORDER_BITREV && B < R) throw std::runtime_error("a tree of fewer than R^2 leaves over R ranks");
    if (single) {                               // raw-element leaves (merkle/mod.rs:113-117)
        if (order == SS_ORDER_BITREV) ok(ss_bitrev_permute32(ctx_, blocks[0], log_B, mine->u8()));
        else ok(ss_dev_copy(ctx_, mine->u8(), blocks[0], 32 * B));
    } else {
        const int row_hash = claim_.tree_kind == SS_TREE_KECCAK ? SS_HASH_KECCAK : claim_.tree_kind == SS_TREE_KECCAK_M20 ? SS_HASH_KECCAK_M20 : SS_HASH_BLAKE2S_M20;
        ok(ss_hash_rows_ex(ctx_, row_hash, blocks.data(), (uint32_t)blocks.size(), B, order, mine->u8()));      // the blocks' halo is not hashed
    }
    if (R == 1 || order != SS_ORDER_BITREV) {   // natural order: a rank's rows are its leaves
        com->leaves_own = mine;
    } else {
        const uint64_t chunk = B / R;
        Buf got = std::make_shared<DeviceBuffer>(ctx_, 32 * B);
        std::vector<Message> sends, recvs;
        for (uint32_t p = 0; p < R; ++p) {
            sends.push_back({p, mine->u8() + 32 * p * chunk, 32 * chunk});
            recvs.push_back({p, got->u8() + 32 * p * chunk, 32 * chunk});
        }
        comm_.exchange(ctx_, sends, recvs);
        com->leaves_own = std::make_shared<DeviceBuffer>(ctx_, 32 * B);
        for (uint32_t src = 0; src < R; ++src)
            ok(ss_dev_copy_2d(ctx_, com->leaves_own->u8() + 32 * brev(src, log_R), 32 * (size_t)R, got->u8() + 32 * src * chunk, 32, 32, chunk));
    }
    com->leaves = com->leaves_own->u8();
    // this rank's sub-tree: its root sits at depth log2 R of the whole tree
    com->nodes = std::make_shared<DeviceBuffer>(ctx_, 64 * B);
    if (friendly) com->tags = std::make_shared<DeviceBuffer>(ctx_, 2 * B);
    const uint32_t sub_friendly = claim_.n_friendly_layers > log_R ? claim_.n_friendly_layers - log_R : 0;
    std::array<uint8_t, 33> sub_root{};
    ok(ss_merkle_build(ctx_, claim_.tree_kind, sub_friendly, com->leaf_kind, com->leaves, B, com->nodes->u8(), com->tags ? com->tags->u8() : nullptr,
                       sub_root.data()));
    const std::vector<uint8_t> roots = comm_.all_gather(ctx_, std::vector<uint8_t>(sub_root.begin(), sub_root.end()));
    std::vector<std::array<uint8_t, 33>> level(R);
    for (uint32_t p = 0; p < R; ++p) memcpy(level[p].data(), roots.data() + 33 * p, 33);
    com->top.push_back(level);
    // the log2 R levels above them on the host (<= 7 hashes); a single-column friendly tree is Pedersen at every level (mod.rs:113-117)
    const uint32_t nf = (single && friendly) ? (1u << 30) : claim_.n_friendly_layers;
    uint32_t depth = log_R;
    while (com->top.front().size() > 1) {
        --depth;
        const auto &lvl = com->top.front();
        std::vector<std::array<uint8_t, 33>> up(lvl.size() / 2);
        for (size_t q = 0; q < up.size(); ++q) up[q] = merge_nodes(claim_.tree_kind, nf, depth, lvl[2 * q], lvl[2 * q + 1]);
        com->top.insert(com->top.begin(), up);
    }
    com->root = com->top.front()[0];
    return com;
}

// -> on rank 0: rows [nq x ncols felts], paths [nq x log N x 32], leaf digests [nq x 32] (hashed leaves), tags [nq x log N] (friendly trees)
void ShardedProver::open(const Commitment &com, const std::vector<Buf> &blocks, uint64_t N, const std::vector<uint64_t> &positions, int order,
                         std::vector<uint64_t> *rows, std::vector<uint8_t> *paths, std::vector<uint8_t> *leaves, std::vector<uint8_t> *tags) {
    std::vector<const uint64_t *> cols;
    for (const Buf &b : blocks) cols.push_back(b->u64());
    open(com, cols, N, positions, order, rows, paths, leaves, tags);
}
void ShardedProver::open(const Commitment &com, const std::vector<const uint64_t *> &blocks, uint64_t N, const std::vector<uint64_t> &positions, int order,
                         std::vector<uint64_t> *rows, std::vector<uint8_t> *paths, std::vector<uint8_t> *leaves, std::vector<uint8_t> *tags) {
    const uint32_t R = comm_.world, r = comm_.rank;
    const uint64_t B = N / R;
    const uint32_t log_N = log2u(N), log_R = log2u(R), log_B = log_N - log_R;
    const size_t ncols = blocks.size(), nq = positions.size();
    const bool hashed = com.leaf_kind == SS_LEAF_DIGEST, friendly = (bool)com.tags;
    // this rank's share: rows of its row block, paths of its leaf block
    Writer part;
    {
        std::vector<uint64_t> q_rows, k_rows, q_leaf, k_leaf;
        for (size_t q = 0; q < nq; ++q) {
            const uint64_t nat = order == SS_ORDER_BITREV ? brev(positions[q], log_N) : positions[q];
            if (nat / B == r) { q_rows.push_back(q); k_rows.push_back(nat % B); }
            if (positions[q] / B == r) { q_leaf.push_back(q); k_leaf.push_back(positions[q] % B); }
        }
        part.u64(q_rows.size());
        if (!q_rows.empty()) {
            std::vector<uint64_t> got(q_rows.size() * ncols * 4);
            ok(ss_gather_rows(ctx_, blocks.data(), (uint32_t)ncols, k_rows.data(), (uint32_t)k_rows.size(), got.data()));
            for (size_t t = 0; t < q_rows.size(); ++t) { part.u64(q_rows[t]); part.raw(got.data() + t * ncols * 4, ncols * 32); }
        }
        part.u64(q_leaf.size());
        if (!q_leaf.empty()) {
            std::vector<uint8_t> pth(q_leaf.size() * log_B * 32), tg(q_leaf.size() * log_B, 0), dg;
            if (log_B) ok(ss_merkle_open(ctx_, com.nodes->u8(), com.tags ? com.tags->u8() : nullptr, B, k_leaf.data(), (uint32_t)k_leaf.size(), pth.data(),
                                         friendly ? tg.data() : nullptr));
            if (hashed) {                    // only the opened leaves cross to the host
                dg.resize(q_leaf.size() * 32);
                const uint64_t *col = (const uint64_t *)com.leaves;
                ok(ss_gather_rows(ctx_, &col, 1, k_leaf.data(), (uint32_t)k_leaf.size(), (uint64_t *)dg.data()));
            }
            for (size_t t = 0; t < q_leaf.size(); ++t) {
                part.u64(q_leaf[t]);
                part.raw(pth.data() + t * log_B * 32, log_B * 32);
                part.raw(tg.data() + t * log_B, log_B);
                if (hashed) part.raw(dg.data() + 32 * t, 32);
            }
        }
    }
    const std::vector<std::vector<uint8_t>> parts = comm_.all_gather_var(ctx_, part.b);
    if (r != 0) return;
    rows->assign(nq * ncols * 4, 0);
    paths->assign(nq * log_N * 32, 0);
    leaves->assign(hashed ? nq * 32 : 0, 0);
    tags->assign(friendly ? nq * log_N : 0, 0);
    for (const std::vector<uint8_t> &pb : parts) {
        if (pb.empty()) continue;
        Reader rd{pb};
        for (uint64_t cnt = rd.u64(), t = 0; t < cnt; ++t) { const uint64_t q = rd.u64(); memcpy(rows->data() + q * ncols * 4, rd.take(ncols * 32), ncols * 32); }
        for (uint64_t cnt = rd.u64(), t = 0; t < cnt; ++t) {
            const uint64_t q = rd.u64();
            memcpy(paths->data() + q * log_N * 32, rd.take(log_B * 32), log_B * 32);
            const uint8_t *tg = rd.take(log_B);
            if (friendly) memcpy(tags->data() + q * log_N, tg, log_B);
            if (hashed) memcpy(leaves->data() + 32 * q, rd.take(32), 32);
        }
    }
    for (size_t q = 0; q < nq; ++q) {           // the top levels: siblings of the sub-tree root's ancestors
        uint64_t node = positions[q] / B;
        for (uint32_t lvl = 0; lvl < log_R; ++lvl) {
            const std::array<uint8_t, 33> &sib = com.top[log_R - lvl][node ^ 1];
            memcpy(paths->data() + (q * log_N + log_B + lvl) * 32, sib.data(), 32);
            if (friendly) (*tags)[q * log_N + log_B + lvl] = sib[32];
            node >>= 1;
        }
    }
}

bool ShardedProver::prove(const Digest &coin_seed, const std::map<uint32_t, uint64_t *> &my_base, const ShardedExtensionBuilder &build_extension,
                          uint64_t n, Proof *out) {
    Air &air = *claim_.air;
    const uint32_t R = comm_.world, r = comm_.rank;
    if (R & (R - 1)) throw std::runtime_error("the row blocks need a power-of-two number of ranks");
    const uint32_t log_n = log2u(n), lb = log2u(opt_.lde_blowup_factor), log_N = log_n + lb;
    const uint64_t N = n << lb, B = N / R;
    if (N % R || n / R < 1) throw std::runtime_error("more ranks than trace rows");
    const Felt g = felt_from_u64(conv_.lde_offset);
    const int order = conv_.bitrev_commit ? SS_ORDER_BITREV : SS_ORDER_NATURAL;
    const uint32_t nb = air.num_base_columns, ne = air.num_extension_columns;
    // rows behind a block that its constraints reach (wrap-around included); never more than the rest of the domain
    uint64_t halo = 0;
    for (auto &c : air.mask) halo = std::max<uint64_t>(halo, (uint64_t)c.second << lb);
    if (R > 1 && halo > N - B) throw std::runtime_error("the constraints reach further than the rest of the domain: fewer ranks for a trace this short");
    halo = std::min(halo, N - B);
    PublicCoin coin(claim_.coin_kind, coin_seed);
    Proof proof;
    proof.options = opt_;
    proof.tree_kind = claim_.tree_kind;
    proof.trace_len = n;

    std::map<uint32_t, Buf> coeffs;              // bit-reversed coefficients of the columns this rank extended
    auto extend = [&](const std::map<uint32_t, uint64_t *> &owned) {       // -> the whole extended columns (released once re-sharded)
        std::map<uint32_t, Buf> ev;
        std::vector<const uint64_t *> in;
        std::vector<uint64_t *> evp, cop;
        for (auto &kv : owned) {
            Buf e = std::make_shared<DeviceBuffer>(ctx_, 32 * N), c = std::make_shared<DeviceBuffer>(ctx_, 32 * n);
            coeffs[kv.first] = c;
            ev[kv.first] = e;
            in.push_back(kv.second); evp.push_back(e->u64()); cop.push_back(c->u64());
        }
        if (!in.empty()) ok(ss_lde_fp252(ctx_, in.data(), (uint32_t)in.size(), log_n, lb, g.data(), evp.data(), cop.data()));
        return ev;
    };
    const uint64_t nb_rows = n / R;              // rows of a trace-size block
    // Whole columns on their owners -> every rank's LDE row block of them, as ONE transform per column over the ranks: the owner
    // deals out blocks of n / R rows, the inverse and the forward transform run spread (spread_inverse / spread_forward), the
    // halo comes from the next ranks.  -> the blocks (with halo); the bit-reversed coefficient blocks are appended to `co`.
    auto spread_lde = [&](const std::map<uint32_t, uint64_t *> &mine, uint32_t first_col, uint32_t ncols, std::vector<Buf> *co) {
        std::vector<Buf> xb;
        std::vector<Message> sends, recvs;
        for (uint32_t c = first_col; c < first_col + ncols; ++c) {
            Buf blk = std::make_shared<DeviceBuffer>(ctx_, 32 * nb_rows);
            const uint32_t o = owner(c);
            if (o == r) {
                const uint8_t *col = (const uint8_t *)mine.at(c);
                for (uint32_t p = 0; p < R; ++p) {
                    if (p == r) ok(ss_dev_copy(ctx_, blk->u8(), col + 32 * p * nb_rows, 32 * nb_rows));
                    else sends.push_back({p, (void *)(col + 32 * p * nb_rows), 32 * nb_rows});
                }
            } else recvs.push_back({o, blk->u8(), 32 * nb_rows});
            xb.push_back(blk);
        }
        if (R > 1) comm_.exchange(ctx_, sends, recvs);
        std::vector<Buf> cb = spread_inverse(xb, log_n, nullptr);
        xb.clear();
        std::vector<Buf> ev = with_halo(spread_forward(cb, log_N, lb, &g), B, halo);
        co->insert(co->end(), cb.begin(), cb.end());
        return ev;
    };
    // the same from columns that ARRIVE as blocks of n / R rows (the extension trace made by build_extension_blocks): no scatter
    auto spread_lde_of_blocks = [&](const std::vector<uint64_t *> &blks, std::vector<Buf> *co) {
        std::vector<Buf> xb;
        for (uint64_t *p : blks) {
            Buf blk = std::make_shared<DeviceBuffer>(ctx_, 32 * nb_rows);
            ok(ss_dev_copy(ctx_, blk->u8(), p, 32 * nb_rows));
            xb.push_back(blk);
        }
        std::vector<Buf> cb = spread_inverse(xb, log_n, nullptr);
        xb.clear();
        std::vector<Buf> ev = with_halo(spread_forward(cb, log_N, lb, &g), B, halo);
        co->insert(co->end(), cb.begin(), cb.end());
        return ev;
    };
    // 2. base trace: R columns at a time, one per rank, extended whole on their owners and dealt out as row blocks; a few columns
    // left over (9 columns on 8 ranks: one) each as one transform over all the ranks - no rank extends two while the others wait
    for (uint32_t c = 0; c < nb; ++c)
        if ((owner(c) == r) != (my_base.count(c) == 1)) throw std::runtime_error("column c lives on rank c % R");
    // (a spread column costs a rank 1 / R of its butterflies and four all-to-alls of ~1 / R of it; a column on its owner a whole
    // LDE and one re-shard while R - L ranks idle: spread the L left-over columns when they are few, L <= R / 2)
    const uint32_t nb_owned = (R == 1 || nb % R > R / 2) ? nb : (nb / R) * R;          // columns 0 .. nb_owned: by owner; nb_owned .. nb: spread
    std::vector<Buf> spread_coeff_blocks;                            // coefficient blocks of the spread columns: nb_owned .. nb, then the extension's
    std::vector<Buf> base_blocks;
    {
        std::map<uint32_t, uint64_t *> owned_part;
        for (auto &kv : my_base) if (kv.first < nb_owned) owned_part[kv.first] = kv.second;
        base_blocks = to_row_blocks(extend(owned_part), nb_owned, 0, N, halo);
        if (nb_owned < nb) {
            std::vector<Buf> more = spread_lde(my_base, nb_owned, nb - nb_owned, &spread_coeff_blocks);
            base_blocks.insert(base_blocks.end(), more.begin(), more.end());
        }
    }
    auto base_com = commit(base_blocks, N, order);
    proof.base_root = base_com->root;
    coin.reseed_with_digest(digest_of(proof.base_root));
    // 3-4. challenges -> extension trace: every column ONE transform over the ranks
    for (uint32_t i = 0; i < air.num_challenges; ++i) proof.challenges.push_back(coin.draw());
    std::vector<Buf> blocks = base_blocks, ext_blocks;
    std::unique_ptr<Commitment> ext_com;
    if (ne) {
        if (ext_blocks_) {
            const std::vector<uint64_t *> blks = ext_blocks_(proof.challenges);
            if (blks.size() != ne) throw std::runtime_error("the extension builder returns one row block per extension column");
            ext_blocks = spread_lde_of_blocks(blks, &spread_coeff_blocks);
        } else {
            const std::map<uint32_t, uint64_t *> my_ext = build_extension(proof.challenges);
            for (uint32_t c = nb; c < nb + ne; ++c)
                if ((owner(c) == r) != (my_ext.count(c) == 1)) throw std::runtime_error("extension column c lives on rank c % R");
            ext_blocks = spread_lde(my_ext, nb, ne, &spread_coeff_blocks);
        }
        air.prepare_program(n, proof.challenges);        // (host work beside the queued transforms: Air::prepare_program)
        ext_com = commit(ext_blocks, N, order);
        proof.has_extension = true;
        proof.extension_root = ext_com->root;
        coin.reseed_with_digest(digest_of(proof.extension_root));
        blocks.insert(blocks.end(), ext_blocks.begin(), ext_blocks.end());
    }
    std::vector<const uint64_t *> block_ptrs;
    for (const Buf &b : blocks) block_ptrs.push_back(b->u64());
    // 5. the composition constraint on this rank's rows; its interpolation and the two column extensions over all the ranks
    proof.composition_coeff = coin.draw();
    AirProgramData pd = air.build_program(n, proof.challenges, proof.composition_coeff);
    const std::vector<uint64_t> consts = flat(pd.program.consts);
    ss_air_program prog;
    prog.code = pd.program.code.data(); prog.n_instr = pd.program.n_instr();
    prog.consts = consts.data(); prog.n_consts = (uint32_t)pd.program.consts.size();
    prog.d_tables = pd.d_tables; prog.table_desc = pd.table_desc.data(); prog.n_tables = (uint32_t)(pd.table_desc.size() / 2);
    prog.n_slots = pd.program.n_slots;
    Buf q_block = std::make_shared<DeviceBuffer>(ctx_, 32 * B);
    if (R == 1) ok(ss_eval_quotient(ctx_, &prog, block_ptrs.data(), (uint32_t)block_ptrs.size(), log_n, lb, g.data(), q_block->u64()));
    else ok(ss_eval_quotient_rows(ctx_, &prog, block_ptrs.data(), (uint32_t)block_ptrs.size(), log_n, lb, g.data(), r * B, B, B + halo, q_block->u64()));
    const uint32_t ncomp = conv_.composition_columns;
    if (ncomp != (1u << lb) || ncomp != 2) throw std::runtime_error("composition split implemented for blowup 2");
    // the N coefficients in bit-reversed order: positions < n are H0's (in its own bit-reversed order), the rest H1's - the split
    // is free, and rank r's block of the array is one of the two halves' blocks 2 r', 2 r' + 1: one exchange deals them out
    std::vector<Buf> comp_coeff_blocks;
    {
        std::vector<Buf> cb = spread_inverse({q_block}, log_N, &g);
        q_block.reset();
        if (R == 1) {
            for (uint32_t k = 0; k < ncomp; ++k) {
                Buf b = std::make_shared<DeviceBuffer>(ctx_, 32 * n);
                ok(ss_dev_copy(ctx_, b->u8(), cb[0]->u8() + 32 * n * k, 32 * n));
                comp_coeff_blocks.push_back(b);
            }
        } else {
            std::vector<Message> sends, recvs;
            for (uint32_t k = 0; k < ncomp; ++k) comp_coeff_blocks.push_back(std::make_shared<DeviceBuffer>(ctx_, 32 * nb_rows));
            const uint32_t mine_k = r / (R / 2), r2 = r % (R / 2);         // this rank's block: halves 2 r2, 2 r2 + 1 of H_{mine_k}
            for (uint32_t h = 0; h < 2; ++h) {
                const uint32_t dst = 2 * r2 + h;
                uint8_t *src = cb[0]->u8() + 32 * nb_rows * h;
                if (dst == r) ok(ss_dev_copy(ctx_, comp_coeff_blocks[mine_k]->u8(), src, 32 * nb_rows));
                else sends.push_back({dst, src, 32 * nb_rows});
            }
            for (uint32_t k = 0; k < ncomp; ++k) {
                const uint32_t src_rank = k * (R / 2) + r / 2;
                if (src_rank != r) recvs.push_back({src_rank, comp_coeff_blocks[k]->u8(), 32 * nb_rows});
            }
            comm_.exchange(ctx_, sends, recvs);
        }
    }
    std::vector<Buf> comp_blocks = spread_forward(comp_coeff_blocks, log_N, lb, &g);
    auto comp_com = commit(comp_blocks, N, order);
    proof.composition_root = comp_com->root;
    coin.reseed_with_digest(digest_of(proof.composition_root));
    // 6. out-of-domain point.  Base columns: their owner evaluates its cells.  A column whose coefficients are spread over the
    // ranks: block r of the bit-reversed coefficient array holds the coefficients j = R j' + bitrev(r), in the bit-reversed
    // order of j' - P(y) = sum_r y^bitrev(r) P_r(y^R), and P_r(y^R) at y = z w_n^off is the out-of-domain evaluation of the
    // block as a polynomial of n / R coefficients at the point z^R with the same offset.  Everybody learns everything.
    proof.z = coin.draw();
    const uint32_t nmask = (uint32_t)air.mask.size();
    std::vector<uint32_t> mask_col, mask_off;
    for (auto &c : air.mask) { mask_col.push_back(c.first); mask_off.push_back(c.second); }
    const uint32_t log_R = log2u(R), br_r = (uint32_t)brev(r, log_R);
    {
        Writer part;
        std::vector<uint32_t> cols_mine, cell_j, cell_col, cell_off;
        for (auto &kv : coeffs) cols_mine.push_back(kv.first);                    // ascending (std::map)
        for (uint32_t j = 0; j < nmask; ++j)
            if (mask_col[j] < nb_owned && owner(mask_col[j]) == r) {
                cell_j.push_back(j);
                cell_col.push_back((uint32_t)(std::find(cols_mine.begin(), cols_mine.end(), mask_col[j]) - cols_mine.begin()));
                cell_off.push_back(mask_off[j]);
            }
        part.u64(cell_j.size());
        if (!cell_j.empty()) {
            std::vector<const uint64_t *> cp;
            for (uint32_t c : cols_mine) cp.push_back(coeffs[c]->u64());
            std::vector<uint64_t> vals(4 * cell_j.size());
            ok(ss_ood_eval(ctx_, cp.data(), (uint32_t)cp.size(), log_n, cell_col.data(), cell_off.data(), (uint32_t)cell_j.size(), proof.z.data(), vals.data()));
            for (size_t t = 0; t < cell_j.size(); ++t) { part.u64(cell_j[t]); part.raw(vals.data() + 4 * t, 32); }
        }
        // the partial sums of the extension columns' cells and of the composition columns at z^ncomp
        std::vector<uint32_t> xcell_j, xcell_col, xcell_off;
        for (uint32_t j = 0; j < nmask; ++j)
            if (mask_col[j] >= nb_owned) { xcell_j.push_back(j); xcell_col.push_back(mask_col[j] - nb_owned); xcell_off.push_back(mask_off[j]); }
        const Felt zR = felt_pow(proof.z, R), zc = felt_pow(proof.z, ncomp), zcR = felt_pow(zc, R);
        std::vector<uint64_t> xvals(4 * xcell_j.size()), cvals(4 * ncomp);
        if (!xcell_j.empty()) {
            std::vector<const uint64_t *> cp;
            for (const Buf &b : spread_coeff_blocks) cp.push_back(b->u64());
            ok(ss_ood_eval(ctx_, cp.data(), (uint32_t)cp.size(), log_n - log_R, xcell_col.data(), xcell_off.data(), (uint32_t)xcell_j.size(), zR.data(), xvals.data()));
        }
        {
            std::vector<const uint64_t *> cp;
            for (const Buf &b : comp_coeff_blocks) cp.push_back(b->u64());
            ok(ss_poly_eval(ctx_, cp.data(), ncomp, log_n - log_R, zcR.data(), cvals.data()));
        }
        part.raw(xvals.data(), 8 * xvals.size());
        part.raw(cvals.data(), 8 * cvals.size());
        proof.ood_trace.assign(nmask, Felt{});
        proof.ood_composition.assign(ncomp, Felt{});
        const Felt w_n = root_of_unity(log_n);
        const std::vector<std::vector<uint8_t>> parts = comm_.all_gather_var(ctx_, part.b);
        for (uint32_t p = 0; p < R; ++p) {
            Reader rd{parts[p]};
            for (uint64_t cnt = rd.u64(), t = 0; t < cnt; ++t) { const uint64_t j = rd.u64(); memcpy(proof.ood_trace[j].data(), rd.take(32), 32); }
            const uint64_t e = brev(p, log_R);                       // rank p's blocks hold the coefficients j = e mod R
            for (size_t t = 0; t < xcell_j.size(); ++t) {
                Felt v;
                memcpy(v.data(), rd.take(32), 32);
                const Felt y = felt_mul(proof.z, felt_pow(w_n, xcell_off[t]));
                Felt &acc = proof.ood_trace[xcell_j[t]];
                acc = felt_add(acc, felt_mul(felt_pow(y, e), v));
            }
            for (uint32_t k = 0; k < ncomp; ++k) {
                Felt v;
                memcpy(v.data(), rd.take(32), 32);
                proof.ood_composition[k] = felt_add(proof.ood_composition[k], felt_mul(felt_pow(zc, e), v));
            }
        }
        (void)br_r;
        std::vector<Felt> all = proof.ood_trace;
        all.insert(all.end(), proof.ood_composition.begin(), proof.ood_composition.end());
        coin.reseed_with_field_elements(all);
    }
    // 7. DEEP composition on this rank's part of the trace-size sub-coset, then its extension over all the ranks: the values on
    // offset * <w_n> are those of Q(y) = P(offset y) on <w_n>, and P(offset w_2n^k) = Q(w_2n^k) (ss_deep_extend's argument)
    proof.deep_alpha = coin.draw();
    std::vector<Felt> dcoef;
    Felt cur = felt_from_u64(1);
    for (uint32_t i = 0; i < nmask + ncomp; ++i) { dcoef.push_back(cur); cur = felt_mul(cur, proof.deep_alpha); }
    const std::vector<uint64_t> ct = flat(std::vector<Felt>(dcoef.begin(), dcoef.begin() + nmask)), cc = flat(std::vector<Felt>(dcoef.begin() + nmask, dcoef.end()));
    const std::vector<uint64_t> ood_t = flat(proof.ood_trace), ood_c = flat(proof.ood_composition);
    const uint64_t cnt = n / R;
    Buf sub_block = std::make_shared<DeviceBuffer>(ctx_, 32 * cnt);
    std::vector<const uint64_t *> comp_ptrs;
    for (const Buf &b : comp_blocks) comp_ptrs.push_back(b->u64());
    ok(ss_deep_compose_rows(ctx_, block_ptrs.data(), (uint32_t)block_ptrs.size(), comp_ptrs.data(), ncomp, log_n, lb, g.data(), mask_col.data(), mask_off.data(),
                            nmask, ood_t.data(), ct.data(), ood_c.data(), cc.data(), proof.z.data(), r * cnt, cnt, sub_block->u64()));
    Buf layer;                                   // the current FRI layer: this rank's block of it (R == 1: all of it)
    if (R == 1) {
        layer = std::make_shared<DeviceBuffer>(ctx_, 32 * N);
        ok(ss_deep_extend(ctx_, sub_block->u64(), log_n, lb, g.data(), layer->u64()));
    } else {
        layer = spread_forward(spread_inverse({sub_block}, log_n, nullptr), log_N, lb, nullptr)[0];
    }
    sub_block.reset();
    // 8. FRI.  A layer above 2^fri_spread_min_log values is folded by all the ranks: rank m takes the fold rows [m rows / R, ...)
    // - all `fold` entries of a row, which sit len / fold apart: one all-to-all out of the block layout -, commits them (its
    // leaf block after the digest routing of commit()) and folds them into ITS block of the next layer.  Then the layer is
    // gathered to rank 0, which finishes as the single-device prover does.
    struct SpreadLayer { std::unique_ptr<Commitment> com; Buf rows_buf; std::vector<const uint64_t *> cols; uint64_t rows; };
    std::vector<SpreadLayer> spread_layers;
    const uint32_t fold = opt_.fri_folding_factor, log_fold = log2u(fold);
    uint32_t log_len = log_N;
    Felt fri_offset = g;
    uint64_t degree_bound = n;
    const char *min_log_env = getenv("SSH_FRI_SPREAD_MIN_LOG");                  // (a test knob: small proofs spread their layers too)
    const uint32_t spread_min_log = min_log_env ? (uint32_t)atoi(min_log_env) : 21u;
    while (R > 1 && degree_bound > opt_.fri_max_remainder_coeffs && log_len > spread_min_log && log_len >= log_fold + 2 * log_R && fold >= R && fold % R == 0) {
        const uint64_t len = 1ull << log_len, rows = len >> log_fold, rcnt = rows / R, Bl = len / R;
        SpreadLayer L;
        L.rows = rows;
        L.rows_buf = std::make_shared<DeviceBuffer>(ctx_, 32 * fold * rcnt);
        {   // block layout -> row layout: my block holds fold / R whole columns k of the layer; chunk m of column k goes to rank m's slot k
            std::vector<Message> sends, recvs;
            const uint32_t kper = fold / R;
            for (uint32_t k = 0; k < fold; ++k) {
                const uint32_t src_rank = k / kper;
                for (uint32_t m = 0; m < R; ++m) {
                    if (src_rank == r) {
                        uint8_t *src = layer->u8() + 32 * ((uint64_t)(k % kper) * rows + (uint64_t)m * rcnt);
                        if (m == r) ok(ss_dev_copy(ctx_, L.rows_buf->u8() + 32 * (uint64_t)k * rcnt, src, 32 * rcnt));
                        else sends.push_back({m, src, 32 * rcnt});
                    }
                }
                if (src_rank != r) recvs.push_back({src_rank, L.rows_buf->u8() + 32 * (uint64_t)k * rcnt, 32 * rcnt});
            }
            (void)Bl;
            comm_.exchange(ctx_, sends, recvs);
        }
        // committed row = the natural stride columns in bit-reversed column order (fri_commit_phase)
        for (uint32_t j = 0; j < fold; ++j) L.cols.push_back(L.rows_buf->u64() + 4 * rcnt * (conv_.bitrev_commit ? brev(j, log_fold) : j));
        L.com = commit(L.cols, rows, order);
        FriLayerProof lp;
        lp.root = L.com->root;
        lp.log_len = log_len;
        proof.fri_layers.push_back(lp);
        coin.reseed_with_digest(digest_of(lp.root));
        Felt alpha = coin.draw();
        if (conv_.fri_alpha_times_offset) alpha = felt_mul(alpha, fri_offset);
        proof.fri_alphas.push_back(alpha);
        Buf next = std::make_shared<DeviceBuffer>(ctx_, 32 * rcnt);
        ok(ss_fri_fold_rows(ctx_, L.rows_buf->u64(), log_len, fold, alpha.data(), fri_offset.data(), conv_.fri_unnormalised ? SS_FRI_UNNORMALISED : 0,
                            (uint64_t)r * rcnt, rcnt, next->u64()));
        spread_layers.push_back(std::move(L));
        layer = next;
        log_len -= log_fold;
        fri_offset = felt_pow(fri_offset, fold);
        degree_bound /= fold;
    }
    std::vector<FriLayerState> layers;
    std::vector<uint64_t> positions;
    const uint64_t len_now = 1ull << log_len, blk_now = len_now / R;
    if (r != 0) {
        if (R > 1) comm_.exchange(ctx_, {{0, layer->u8(), 32 * blk_now}}, {});
    } else {
        Buf whole = layer;
        if (R > 1) {
            whole = std::make_shared<DeviceBuffer>(ctx_, 32 * len_now);
            ok(ss_dev_copy(ctx_, whole->u8(), layer->u8(), 32 * blk_now));
            std::vector<Message> recvs;
            for (uint32_t p = 1; p < R; ++p) recvs.push_back({p, whole->u8() + 32 * p * blk_now, 32 * blk_now});
            comm_.exchange(ctx_, {}, recvs);
        }
        // 8-9. the rest of FRI, proof of work, query positions: on rank 0, as the single-device prover does them
        layers = fri_commit_phase_from(ctx_, claim_, conv_, opt_, coin, proof, whole, log_len, fri_offset, degree_bound);
        proof.pow_nonce = proof_of_work(ctx_, claim_, coin, opt_, have_nonce_, nonce_);
        coin.reseed_with_int(proof.pow_nonce);
        positions = coin.draw_queries(opt_.num_queries, N);
        proof.query_positions = positions;
    }
    layer.reset();
    {                                            // the positions to every rank
        Writer w;
        if (r == 0) for (uint64_t p : positions) w.u64(p);
        const std::vector<std::vector<uint8_t>> got = comm_.all_gather_var(ctx_, w.b);
        if (r != 0) {
            Reader rd{got[0]};
            while (!rd.done()) positions.push_back(rd.u64());
        }
    }
    // openings of the three trace commitments: rows from the row-block ranks, paths from the leaf-block ranks
    open(*base_com, base_blocks, N, positions, order, &proof.base_rows, &proof.base_paths, &proof.base_leaves, &proof.base_path_tags);
    if (ne) open(*ext_com, ext_blocks, N, positions, order, &proof.extension_rows, &proof.extension_paths, &proof.extension_leaves, &proof.extension_path_tags);
    open(*comp_com, comp_blocks, N, positions, order, &proof.composition_rows, &proof.composition_paths, &proof.composition_leaves, &proof.composition_path_tags);
    // the FRI layers the ranks folded: the same openings over their row matrices; position p of a layer -> row p >> log2 fold of it
    std::vector<uint64_t> fp = positions;
    for (size_t li = 0; li < spread_layers.size(); ++li) {
        SpreadLayer &L = spread_layers[li];
        std::vector<uint64_t> nxt;
        for (uint64_t q : fp) nxt.push_back(conv_.bitrev_commit ? (q >> log_fold) : (q % L.rows));
        std::sort(nxt.begin(), nxt.end());
        nxt.erase(std::unique(nxt.begin(), nxt.end()), nxt.end());
        fp = nxt;
        FriLayerProof lpo;
        open(*L.com, L.cols, L.rows, fp, order, &lpo.rows, &lpo.paths, &lpo.leaves, &lpo.path_tags);
        if (r == 0) {
            FriLayerProof &lp = proof.fri_layers[li];
            lp.positions = fp;
            lp.rows = std::move(lpo.rows); lp.paths = std::move(lpo.paths); lp.leaves = std::move(lpo.leaves); lp.path_tags = std::move(lpo.path_tags);
        }
    }
    if (r != 0) return false;
    fri_open_from(ctx_, conv_, opt_, proof, layers, fp, spread_layers.size());
    *out = std::move(proof);
    return true;
}

// ------------------------------------------------------------------------------------ one transform over the ranks
// block layout <-> exchanged layout (its own inverse): chunk m of this rank's buffer to rank m, chunk t from rank t
std::vector<ShardedProver::Buf> ShardedProver::exchange_layout(const std::vector<Buf> &in, uint64_t elems) {
    const uint32_t R = comm_.world, r = comm_.rank;
    const uint64_t chunk = elems / R;
    if (chunk * R != elems || !chunk) throw std::runtime_error("a transform of fewer than R^2 points over R ranks");
    std::vector<Buf> out;
    std::vector<Message> sends, recvs;
    for (const Buf &b : in) {
        Buf o = std::make_shared<DeviceBuffer>(ctx_, 32 * elems);
        for (uint32_t p = 0; p < R; ++p) {
            if (p == r) ok(ss_dev_copy(ctx_, o->u8() + 32 * p * chunk, b->u8() + 32 * p * chunk, 32 * chunk));
            else {
                sends.push_back({p, b->u8() + 32 * p * chunk, 32 * chunk});
                recvs.push_back({p, o->u8() + 32 * p * chunk, 32 * chunk});
            }
        }
        out.push_back(o);
    }
    comm_.exchange(ctx_, sends, recvs);
    return out;
}

// blocks of 2^log_n / R values over offset * <w_n>, natural order -> blocks of the bit-reversed coefficient array (inputs kept)
std::vector<ShardedProver::Buf> ShardedProver::spread_inverse(const std::vector<Buf> &blocks, uint32_t log_n, const Felt *offset) {
    const uint32_t R = comm_.world, r = comm_.rank;
    const uint64_t elems = (1ull << log_n) / R;
    std::vector<uint64_t *> ptrs;
    if (R == 1) {
        std::vector<Buf> out;
        for (const Buf &b : blocks) {
            Buf o = std::make_shared<DeviceBuffer>(ctx_, 32 * elems);
            ok(ss_dev_copy(ctx_, o->u8(), b->u8(), 32 * elems));
            out.push_back(o);
            ptrs.push_back(o->u64());
        }
        ok(ss_ntt_fp252(ctx_, ptrs.data(), (uint32_t)ptrs.size(), log_n, SS_NTT_INVERSE, offset ? offset->data() : nullptr, SS_ORDER_NATURAL, SS_ORDER_BITREV));
        return out;
    }
    const uint32_t log_R = log2u(R);
    std::vector<Buf> t = exchange_layout(blocks, elems);
    for (const Buf &b : t) ptrs.push_back(b->u64());
    ok(ss_ntt_shard_fp252(ctx_, ptrs.data(), (uint32_t)ptrs.size(), log_n, log_R, r, SS_NTT_INVERSE, offset ? offset->data() : nullptr, SS_NTT_PART_CROSS, 0, nullptr));
    std::vector<Buf> out = exchange_layout(t, elems);
    t.clear();
    ptrs.clear();
    for (const Buf &b : out) ptrs.push_back(b->u64());
    ok(ss_ntt_shard_fp252(ctx_, ptrs.data(), (uint32_t)ptrs.size(), log_n, log_R, r, SS_NTT_INVERSE, offset ? offset->data() : nullptr, SS_NTT_PART_LOCAL, 0, nullptr));
    return out;
}

// blocks of the bit-reversed coefficient array (2^(log_n - log_expand) / R coefficients each) -> blocks of the 2^log_n evaluations
// over offset * <w>
std::vector<ShardedProver::Buf> ShardedProver::spread_forward(const std::vector<Buf> &coeff_blocks, uint32_t log_n, uint32_t log_expand, const Felt *offset) {
    const uint32_t R = comm_.world, r = comm_.rank;
    const uint64_t elems = (1ull << log_n) / R;
    std::vector<Buf> out;
    std::vector<uint64_t *> src, dst;
    for (const Buf &b : coeff_blocks) {
        out.push_back(std::make_shared<DeviceBuffer>(ctx_, 32 * elems));
        src.push_back(b->u64());
        dst.push_back(out.back()->u64());
    }
    if (R == 1) {
        ok(ss_evaluate_fp252(ctx_, (const uint64_t *const *)src.data(), (uint32_t)src.size(), log_n - log_expand, log_expand, offset ? offset->data() : nullptr, dst.data()));
        return out;
    }
    const uint32_t log_R = log2u(R);
    ok(ss_ntt_shard_fp252(ctx_, src.data(), (uint32_t)src.size(), log_n, log_R, r, SS_NTT_FORWARD, offset ? offset->data() : nullptr, SS_NTT_PART_LOCAL, log_expand, dst.data()));
    std::vector<Buf> t = exchange_layout(out, elems);
    out.clear();
    dst.clear();
    for (const Buf &b : t) dst.push_back(b->u64());
    ok(ss_ntt_shard_fp252(ctx_, dst.data(), (uint32_t)dst.size(), log_n, log_R, r, SS_NTT_FORWARD, offset ? offset->data() : nullptr, SS_NTT_PART_CROSS, 0, nullptr));
    return exchange_layout(t, elems);
}

// rows [r B, (r + 1) B + halo) mod N of vectors held as blocks of B rows: the rows behind a block come from the next ranks
std::vector<ShardedProver::Buf> ShardedProver::with_halo(const std::vector<Buf> &blocks, uint64_t B, uint64_t halo) {
    const uint32_t R = comm_.world, r = comm_.rank;
    if (R == 1 || !halo) return blocks;
    std::vector<Buf> out;
    std::vector<Message> sends, recvs;
    for (const Buf &b : blocks) {
        Buf o = std::make_shared<DeviceBuffer>(ctx_, 32 * (B + halo));
        ok(ss_dev_copy(ctx_, o->u8(), b->u8(), 32 * B));
        for (uint64_t d = 1, left = halo; left; ++d) {
            const uint64_t take = std::min(B, left);
            recvs.push_back({(uint32_t)((r + d) % R), o->u8() + 32 * (B + (d - 1) * B), 32 * take});
            sends.push_back({(uint32_t)((r + R - d % R) % R), b->u8(), 32 * take});
            left -= take;
        }
        out.push_back(o);
    }
    comm_.exchange(ctx_, sends, recvs);
    return out;
}

}  // namespace ssh
