#include "prover.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
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

// ------------------------------------------------------------------ device objects
DeviceBuffer::DeviceBuffer(ss_ctx *ctx, size_t bytes) : ctx_(ctx), bytes_(bytes) { ok(ss_dev_alloc(ctx, bytes, &ptr_)); }
DeviceBuffer::~DeviceBuffer() { if (ptr_) ss_dev_free(ctx_, ptr_); }

Matrix Matrix::alloc(ss_ctx *ctx, uint32_t ncols, uint64_t nrows) {
    Matrix m;
    m.nrows = nrows;
    for (uint32_t c = 0; c < ncols; ++c) {
        auto b = std::make_shared<DeviceBuffer>(ctx, 32 * nrows);
        m.cols.push_back(b->u64());
        m.owned.push_back(b);
    }
    return m;
}

std::unique_ptr<MerkleTree> MerkleTree::from_matrix(ss_ctx *ctx, int tree_kind, uint32_t n_friendly, const Matrix &m, int order) {
    auto t = std::unique_ptr<MerkleTree>(new MerkleTree);
    t->ctx_ = ctx; t->tree_kind_ = tree_kind; t->n_ = m.nrows;
    t->nodes_.reset(new DeviceBuffer(ctx, 64 * m.nrows));
    if (tree_kind == SS_TREE_FRIENDLY) t->tags_.reset(new DeviceBuffer(ctx, 2 * m.nrows));
    const int row_hash = tree_kind == SS_TREE_KECCAK ? SS_HASH_KECCAK : tree_kind == SS_TREE_KECCAK_M20 ? SS_HASH_KECCAK_M20 : SS_HASH_BLAKE2S_M20;
    if (m.num_cols() == 1) {            // single column: the column becomes the leaves (merkle/mod.rs:113-117)
        ok(ss_merkle_build_ex(ctx, tree_kind, n_friendly, SS_LEAF_FELT, m.cols[0], m.nrows, order, t->nodes_->u8(),
                              t->tags_ ? t->tags_->u8() : nullptr, t->root_.data()));
    } else {
        t->leaves_.reset(new DeviceBuffer(ctx, 32 * m.nrows));
        ok(ss_hash_rows_ex(ctx, row_hash, (const uint64_t *const *)m.cols.data(), m.num_cols(), m.nrows, order, t->leaves_->u8()));
        ok(ss_merkle_build(ctx, tree_kind, n_friendly, SS_LEAF_DIGEST, t->leaves_->u8(), m.nrows, t->nodes_->u8(),
                           t->tags_ ? t->tags_->u8() : nullptr, t->root_.data()));
    }
    return t;
}
std::vector<uint8_t> MerkleTree::prove(const std::vector<uint64_t> &idx, std::vector<uint8_t> *tags) const {
    std::vector<uint8_t> out(idx.size() * log2u(n_) * 32);
    if (tags) tags->assign(idx.size() * log2u(n_), 0);
    ok(ss_merkle_open(ctx_, nodes_->u8(), tags_ ? tags_->u8() : nullptr, n_, idx.data(), (uint32_t)idx.size(), out.data(),
                      tags && tags_ ? tags->data() : nullptr));
    return out;
}

std::vector<uint8_t> MerkleTree::leaf_digests(const std::vector<uint64_t> &idx) const {
    std::vector<uint8_t> out;
    if (!leaves_ || idx.empty()) return out;
    out.resize(idx.size() * 32);
    const uint64_t *col = leaves_->u64();           // a digest is one 32-byte entry of a single "column"
    ok(ss_gather_rows(ctx_, &col, 1, idx.data(), (uint32_t)idx.size(), (uint64_t *)out.data()));
    return out;
}

void MerkleTree::queue_openings(GatherBatch &batch, const std::vector<uint64_t> &idx, std::vector<uint8_t> *paths, std::vector<uint8_t> *tags,
                                std::vector<uint8_t> *leaves) const {
    const uint32_t log_n = log2u(n_);
    // the siblings along each path, leaf level first (ss_merkle_open): node numbers of the heap-ordered node array
    std::vector<uint64_t> sib(idx.size() * log_n);
    for (size_t q = 0; q < idx.size(); ++q) {
        if (idx[q] >= n_) throw std::runtime_error("leaf index out of range");
        uint64_t k = n_ + idx[q];
        for (uint32_t l = 0; l < log_n; ++l) { sib[q * log_n + l] = k ^ 1ull; k >>= 1; }
    }
    if (tags) {
        tags->assign(idx.size() * log_n, 0);
        if (tags_) batch.entries(tags_->u8(), 1, sib, tags);
    }
    if (paths) batch.entries(nodes_->u8(), 32, std::move(sib), paths);
    if (leaves) {
        leaves->clear();
        if (leaves_ && !idx.empty()) batch.entries(leaves_->u8(), 32, idx, leaves);
    }
}

void GatherBatch::rows(const std::vector<uint64_t *> &cols, const std::vector<uint64_t> &idx, std::vector<uint64_t> *out) {
    out->assign(idx.size() * cols.size() * 4, 0);
    if (idx.empty() || cols.empty()) return;
    idx_.push_back(idx);
    cols_.emplace_back(cols.begin(), cols.end());
    jobs_.push_back(ss_gather_job{cols_.back().data(), (uint32_t)cols.size(), 32u, idx_.back().data(), (uint32_t)idx.size(), out->data()});
}
void GatherBatch::entries(const void *d_array, uint32_t entry_bytes, std::vector<uint64_t> idx, std::vector<uint8_t> *out) {
    out->assign(idx.size() * entry_bytes, 0);
    if (idx.empty()) return;
    idx_.push_back(std::move(idx));
    cols_.push_back(std::vector<const void *>{d_array});
    jobs_.push_back(ss_gather_job{cols_.back().data(), 1u, entry_bytes, idx_.back().data(), (uint32_t)idx_.back().size(), out->data()});
}
void GatherBatch::run() {
    if (!jobs_.empty()) ok(ss_gather_batch(ctx_, jobs_.data(), (uint32_t)jobs_.size()));
    jobs_.clear(); idx_.clear(); cols_.clear();
}

static std::vector<uint64_t> flat(const std::vector<Felt> &v) {
    std::vector<uint64_t> o(4 * v.size());
    for (size_t i = 0; i < v.size(); ++i) memcpy(o.data() + 4 * i, v[i].data(), 32);
    return o;
}

Felt Air::composition_at(uint64_t, const std::vector<Felt> &, const Felt &, const Felt &, const std::vector<Felt> &) {
    throw std::runtime_error("the " + name + " AIR has no verifier side");
}

// ------------------------------------------------------------- FRI, proof of work
// (free functions: the single-device prover below and the sharded one of sharded.cpp run the same code on rank 0)
static uint64_t brev_bits(uint64_t x, uint32_t bits) { uint64_t r = 0; for (uint32_t i = 0; i < bits; ++i) r |= ((x >> i) & 1ull) << (bits - 1 - i); return r; }
static Digest digest_of_root(const std::array<uint8_t, 33> &r) { Digest d; memcpy(d.data(), r.data(), 32); return d; }

std::vector<FriLayerState> fri_commit_phase(ss_ctx *ctx, const Claim &claim, const Conventions &conv, const ProofOptions &opt, PublicCoin &coin,
                                            Proof &proof, std::shared_ptr<DeviceBuffer> deep, uint32_t log_N, uint64_t n) {
    return fri_commit_phase_from(ctx, claim, conv, opt, coin, proof, deep, log_N, felt_from_u64(conv.lde_offset), n);
}
// the same from a later layer on (the sharded prover folds the first layers on the ranks): `evals` = that layer, 2^log_len values
// over offset * <w>, of a polynomial of degree < degree_bound
std::vector<FriLayerState> fri_commit_phase_from(ss_ctx *ctx, const Claim &claim, const Conventions &conv, const ProofOptions &opt, PublicCoin &coin,
                                                 Proof &proof, std::shared_ptr<DeviceBuffer> evals, uint32_t log_len, Felt offset, uint64_t degree_bound) {
    const int order = conv.bitrev_commit ? SS_ORDER_BITREV : SS_ORDER_NATURAL;
    const uint32_t fold = opt.fri_folding_factor, log_fold = log2u(fold);
    std::vector<FriLayerState> layers;
    while (degree_bound > opt.fri_max_remainder_coeffs) {
        const uint64_t rows = 1ull << (log_len - log_fold);
        FriLayerState L;
        L.evals = evals;
        L.matrix.nrows = rows;
        // committed row r = entries fold*r .. fold*r+fold-1 of the bit-reversed vector = natural row bitrev(r),
        // its entry j at x_r * w_fold^bitrev(j): the natural stride columns, re-ordered
        for (uint32_t j = 0; j < fold; ++j)
            L.matrix.cols.push_back(evals->u64() + 4 * rows * (conv.bitrev_commit ? brev_bits(j, log_fold) : j));
        L.tree = MerkleTree::from_matrix(ctx, claim.tree_kind, claim.n_friendly_layers, L.matrix, order);
        FriLayerProof lp;
        lp.root = L.tree->root();
        lp.log_len = log_len;
        proof.fri_layers.push_back(lp);
        coin.reseed_with_digest(digest_of_root(lp.root));
        Felt alpha = coin.draw();
        if (conv.fri_alpha_times_offset) alpha = felt_mul(alpha, offset);     // challenge = draw * layer offset
        proof.fri_alphas.push_back(alpha);
        auto next = std::make_shared<DeviceBuffer>(ctx, 32 * rows);
        ok(ss_fri_fold_ex(ctx, evals->u64(), log_len, fold, alpha.data(), offset.data(),
                          conv.fri_unnormalised ? SS_FRI_UNNORMALISED : 0, next->u64()));     // natural order in memory
        layers.push_back(std::move(L));
        evals = next;
        log_len -= log_fold;
        offset = felt_pow(offset, fold);
        degree_bound /= fold;
    }
    uint64_t *e = evals->u64();
    const Felt rem_offset = conv.remainder_unshifted ? felt_from_u64(1) : offset;
    ok(ss_ntt_fp252(ctx, &e, 1, log_len, SS_NTT_INVERSE, rem_offset.data(), SS_ORDER_NATURAL, SS_ORDER_NATURAL));
    std::vector<uint64_t> rem(4ull << log_len);
    ok(ss_download(ctx, rem.data(), e, rem.size() * 8));
    const uint64_t keep = degree_bound ? degree_bound : 1;
    for (uint64_t i = 4 * keep; i < rem.size(); ++i) if (rem[i]) throw std::runtime_error("FRI remainder exceeds its degree bound");
    for (uint64_t i = 0; i < keep; ++i) { Felt f; memcpy(f.data(), rem.data() + 4 * i, 32); proof.fri_remainder.push_back(f); }
    coin.reseed_with_field_element_vector(proof.fri_remainder);
    return layers;
}

uint64_t proof_of_work(ss_ctx *ctx, const Claim &claim, PublicCoin &coin, const ProofOptions &opt, bool have_nonce, uint64_t nonce) {
    if (have_nonce) {
        if (!verify_proof_of_work(claim.coin_kind, coin.digest(), opt.grinding_factor, nonce))
            throw std::runtime_error("the supplied proof-of-work nonce is not valid for this transcript");
        return nonce;
    }
    uint64_t found = 0;
    if (opt.grinding_factor) ok(ss_pow_grind(ctx, claim.coin_kind, coin.digest().data(), opt.grinding_factor, &found));
    return found;
}

void fri_open(ss_ctx *ctx, const Conventions &conv, const ProofOptions &opt, Proof &proof, std::vector<FriLayerState> &layers,
              const std::vector<uint64_t> &positions, GatherBatch *batch) {
    fri_open_from(ctx, conv, opt, proof, layers, positions, 0, batch);
}
// layers[k] is the proof's layer first + k; `positions`: the query positions as folded down to layer `first`'s index space
void fri_open_from(ss_ctx *ctx, const Conventions &conv, const ProofOptions &opt, Proof &proof, std::vector<FriLayerState> &layers_from,
                   const std::vector<uint64_t> &positions, size_t first, GatherBatch *batch) {
    GatherBatch own(ctx);
    GatherBatch &gb = batch ? *batch : own;
    const uint32_t log_fold = log2u(opt.fri_folding_factor);
    std::vector<uint64_t> p = positions;
    for (size_t li = first; li < first + layers_from.size(); ++li) {
        FriLayerState &layer = layers_from[li - first];
        const uint32_t row_bits = proof.fri_layers[li].log_len - log_fold;
        const uint64_t rows = 1ull << row_bits;
        std::set<uint64_t> s;
        for (uint64_t q : p) s.insert(conv.bitrev_commit ? (q >> log_fold) : (q % rows));
        p.assign(s.begin(), s.end());
        std::vector<uint64_t> nat_rows = p;
        if (conv.bitrev_commit) for (auto &r : nat_rows) r = brev_bits(r, row_bits);
        proof.fri_layers[li].positions = p;
        gb.rows(layer.matrix.cols, nat_rows, &proof.fri_layers[li].rows);
        layer.tree->queue_openings(gb, p, &proof.fri_layers[li].paths, &proof.fri_layers[li].path_tags, &proof.fri_layers[li].leaves);
    }
    if (!batch) own.run();
}

// ------------------------------------------------------------------------- prove
Proof Prover::prove(const Digest &coin_seed, const Matrix &base_trace, const ExtensionBuilder &build_extension) {
    Air &air = *claim_.air;
    const uint64_t n = base_trace.nrows;
    const uint32_t log_n = log2u(n), lb = log2u(opt_.lde_blowup_factor), log_N = log_n + lb;
    const uint64_t N = n << lb;
    const Felt g = felt_from_u64(conv_.lde_offset);
    PublicCoin coin(claim_.coin_kind, coin_seed);
    Proof proof;
    proof.options = opt_;
    proof.tree_kind = claim_.tree_kind;
    proof.trace_len = n;
    // SSH_TIMING=1: wall clock per stage (with a device sync at each boundary) on stderr
    const bool timing = getenv("SSH_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char *stage) {
        if (!timing) return;
        ss_ctx_sync(ctx_);
        auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[ssh timing] %-28s %9.3f ms\n", stage, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    const int order = conv_.bitrev_commit ? SS_ORDER_BITREV : SS_ORDER_NATURAL;
    auto brev = [](uint64_t x, uint32_t bits) { uint64_t r = 0; for (uint32_t i = 0; i < bits; ++i) r |= ((x >> i) & 1ull) << (bits - 1 - i); return r; };
    auto commit = [&](const Matrix &m) { return MerkleTree::from_matrix(ctx_, claim_.tree_kind, claim_.n_friendly_layers, m, order); };
    auto digest_of = [](const std::array<uint8_t, 33> &r) { Digest d; memcpy(d.data(), r.data(), 32); return d; };

    // 2. base trace: interpolate, extend, commit
    Matrix base_lde = Matrix::alloc(ctx_, base_trace.num_cols(), N), base_co = Matrix::alloc(ctx_, base_trace.num_cols(), n);
    if (feed_.wait) {                        // the columns as they land (ColumnFeed): one transform pair per column, in arrival order
        if (feed_.order.size() != base_trace.num_cols()) throw std::runtime_error("the column feed's order must name every base column once");
        std::vector<uint8_t> seen(base_trace.num_cols(), 0);
        for (uint32_t c : feed_.order) {
            if (c >= base_trace.num_cols() || seen[c]++) throw std::runtime_error("the column feed's order must name every base column once");
            feed_.wait(c);
            const uint64_t *in = base_trace.cols[c];
            ok(ss_lde_fp252(ctx_, &in, 1, log_n, lb, g.data(), &base_lde.cols[c], &base_co.cols[c]));
        }
    } else {
        ok(ss_lde_fp252(ctx_, (const uint64_t *const *)base_trace.cols.data(), base_trace.num_cols(), log_n, lb, g.data(),
                        base_lde.cols.data(), base_co.cols.data()));
    }
    mark("base lde");
    auto base_tree = commit(base_lde);
    mark("base commit");
    proof.base_root = base_tree->root();
    coin.reseed_with_digest(digest_of(proof.base_root));

    // 3-4. challenges -> extension trace
    for (uint32_t i = 0; i < air.num_challenges; ++i) proof.challenges.push_back(coin.draw());
    std::vector<uint64_t *> lde_cols = base_lde.cols, coeff_cols = base_co.cols;
    Matrix ext_lde, ext_co;
    std::unique_ptr<MerkleTree> ext_tree;
    if (air.num_extension_columns) {
        Matrix ext = build_extension(proof.challenges);
        ext_lde = Matrix::alloc(ctx_, ext.num_cols(), N);
        ext_co = Matrix::alloc(ctx_, ext.num_cols(), n);
        ok(ss_lde_fp252(ctx_, (const uint64_t *const *)ext.cols.data(), ext.num_cols(), log_n, lb, g.data(), ext_lde.cols.data(),
                        ext_co.cols.data()));
        // the transforms are queued, not waited for: the host lowers the composition program for these challenges meanwhile (all of
        // it but the powers of the composition coefficient, drawn after the commitment below: Air::prepare_program)
        air.prepare_program(n, proof.challenges);
        mark("extension lde");
        ext_tree = commit(ext_lde);
        mark("extension commit");
        proof.has_extension = true;
        proof.extension_root = ext_tree->root();
        coin.reseed_with_digest(digest_of(proof.extension_root));
        lde_cols.insert(lde_cols.end(), ext_lde.cols.begin(), ext_lde.cols.end());
        coeff_cols.insert(coeff_cols.end(), ext_co.cols.begin(), ext_co.cols.end());
    }

    // 5. composition constraint on the LDE domain; its coefficients in bit-reversed order split
    //    for free into H0 (first half) and H1 (second half)
    proof.composition_coeff = coin.draw();
    AirProgramData pd = air.build_program(n, proof.challenges, proof.composition_coeff);
    std::vector<uint64_t> consts = flat(pd.program.consts);
    ss_air_program prog;
    prog.code = pd.program.code.data(); prog.n_instr = pd.program.n_instr();
    prog.consts = consts.data(); prog.n_consts = (uint32_t)pd.program.consts.size();
    prog.d_tables = pd.d_tables; prog.table_desc = pd.table_desc.data(); prog.n_tables = (uint32_t)(pd.table_desc.size() / 2);
    prog.n_slots = pd.program.n_slots;
    mark("program build");
    DeviceBuffer comp_evals(ctx_, 32 * N);
    ok(ss_eval_quotient(ctx_, &prog, (const uint64_t *const *)lde_cols.data(), (uint32_t)lde_cols.size(), log_n, lb, g.data(), comp_evals.u64()));
    mark("quotient");
    uint64_t *ce = comp_evals.u64();
    ok(ss_ntt_fp252(ctx_, &ce, 1, log_N, SS_NTT_INVERSE, g.data(), SS_ORDER_NATURAL, SS_ORDER_BITREV));
    const uint32_t ncomp = conv_.composition_columns;
    if (ncomp != (1u << lb) || ncomp != 2) throw std::runtime_error("composition split implemented for blowup 2");
    std::vector<uint64_t *> comp_coeffs;
    for (uint32_t k = 0; k < ncomp; ++k) comp_coeffs.push_back(ce + 4 * n * k);
    Matrix comp_lde = Matrix::alloc(ctx_, ncomp, N);
    ok(ss_evaluate_fp252(ctx_, (const uint64_t *const *)comp_coeffs.data(), ncomp, log_n, lb, g.data(), comp_lde.cols.data()));
    mark("composition lde");
    auto comp_tree = commit(comp_lde);
    mark("composition commit");
    proof.composition_root = comp_tree->root();
    coin.reseed_with_digest(digest_of(proof.composition_root));

    // 6. out-of-domain point
    proof.z = coin.draw();
    std::vector<uint32_t> mask_col, mask_off;
    for (auto &c : air.mask) { mask_col.push_back(c.first); mask_off.push_back(c.second); }
    const uint32_t nmask = (uint32_t)air.mask.size();
    std::vector<uint64_t> ood_t(4 * nmask), ood_c(4 * ncomp);
    ok(ss_ood_eval(ctx_, (const uint64_t *const *)coeff_cols.data(), (uint32_t)coeff_cols.size(), log_n, mask_col.data(), mask_off.data(),
                   nmask, proof.z.data(), ood_t.data()));
    const Felt zc = felt_pow(proof.z, ncomp);
    ok(ss_poly_eval(ctx_, (const uint64_t *const *)comp_coeffs.data(), ncomp, log_n, zc.data(), ood_c.data()));
    for (uint32_t j = 0; j < nmask; ++j) { Felt f; memcpy(f.data(), ood_t.data() + 4 * j, 32); proof.ood_trace.push_back(f); }
    for (uint32_t k = 0; k < ncomp; ++k) { Felt f; memcpy(f.data(), ood_c.data() + 4 * k, 32); proof.ood_composition.push_back(f); }
    // DEEP's denominator tables depend on z alone: the device builds them while this thread hashes the out-of-domain values into the
    // coin (the Cairo coin chains one host Pedersen hash per value: ~2 ms at the recursive layout's 135 values)
    ok(ss_deep_prepare(ctx_, ncomp, log_n, g.data(), proof.z.data()));
    {
        std::vector<Felt> all = proof.ood_trace;
        all.insert(all.end(), proof.ood_composition.begin(), proof.ood_composition.end());
        coin.reseed_with_field_elements(all);
    }

    mark("ood");
    // 7. DEEP composition: coefficients are powers of one alpha (src/lib.rs:102-116)
    proof.deep_alpha = coin.draw();
    std::vector<Felt> coeffs;
    Felt cur = felt_from_u64(1);
    for (uint32_t i = 0; i < nmask + ncomp; ++i) { coeffs.push_back(cur); cur = felt_mul(cur, proof.deep_alpha); }
    std::vector<uint64_t> ct = flat(std::vector<Felt>(coeffs.begin(), coeffs.begin() + nmask));
    std::vector<uint64_t> cc = flat(std::vector<Felt>(coeffs.begin() + nmask, coeffs.end()));
    auto deep = std::make_shared<DeviceBuffer>(ctx_, 32 * N);
    ok(ss_deep_compose(ctx_, (const uint64_t *const *)lde_cols.data(), (uint32_t)lde_cols.size(), (const uint64_t *const *)comp_lde.cols.data(), ncomp,
                       log_n, lb, g.data(), mask_col.data(), mask_off.data(), nmask, ood_t.data(), ct.data(), ood_c.data(), cc.data(),
                       proof.z.data(), deep->u64()));

    mark("deep");
    // 8. FRI
    std::vector<FriLayerState> layers = fri_commit_phase(ctx_, claim_, conv_, opt_, coin, proof, deep, log_N, n);

    mark("fri");
    // 9. proof of work, queries, openings
    proof.pow_nonce = proof_of_work(ctx_, claim_, coin, opt_, have_nonce_, nonce_);
    coin.reseed_with_int(proof.pow_nonce);
    proof.query_positions = coin.draw_queries(opt_.num_queries, N);
    const auto &pos = proof.query_positions;
    // a position is an index into the COMMITTED order; the matrices themselves are in natural order
    std::vector<uint64_t> nat = pos;
    if (conv_.bitrev_commit) for (auto &q : nat) q = brev(q, log_N);
    GatherBatch gb(ctx_);                       // every opening of the proof in one round trip to the device
    gb.rows(base_lde.cols, nat, &proof.base_rows);
    base_tree->queue_openings(gb, pos, &proof.base_paths, &proof.base_path_tags, &proof.base_leaves);
    if (ext_tree) {
        gb.rows(ext_lde.cols, nat, &proof.extension_rows);
        ext_tree->queue_openings(gb, pos, &proof.extension_paths, &proof.extension_path_tags, &proof.extension_leaves);
    }
    gb.rows(comp_lde.cols, nat, &proof.composition_rows);
    comp_tree->queue_openings(gb, pos, &proof.composition_paths, &proof.composition_path_tags, &proof.composition_leaves);
    fri_open(ctx_, conv_, opt_, proof, layers, pos, &gb);
    gb.run();
    mark("pow + openings");
    return proof;
}

// ------------------------------------------------------------------ serialisation
// A flat little-endian dump for the Python tests, transcript values included (the reference's
// own wire format is serialize_wire below).
namespace {
struct W {
    std::vector<uint8_t> b;
    void u32(uint32_t v) { for (int i = 0; i < 4; ++i) b.push_back((uint8_t)(v >> (8 * i))); }
    void u64(uint64_t v) { for (int i = 0; i < 8; ++i) b.push_back((uint8_t)(v >> (8 * i))); }
    void raw(const void *p, size_t n) { const uint8_t *q = (const uint8_t *)p; b.insert(b.end(), q, q + n); }
    void felts(const std::vector<Felt> &v) { u32((uint32_t)v.size()); for (auto &f : v) raw(f.data(), 32); }
    void u64s(const std::vector<uint64_t> &v) { u32((uint32_t)v.size()); raw(v.data(), 8 * v.size()); }
    void bytes(const std::vector<uint8_t> &v) { u32((uint32_t)v.size()); raw(v.data(), v.size()); }
};
}  // namespace
std::vector<uint8_t> Proof::serialize() const {
    W w;
    w.u64(trace_len);
    w.raw(base_root.data(), 33); w.u32(has_extension ? 1 : 0); w.raw(extension_root.data(), 33); w.raw(composition_root.data(), 33);
    w.felts(challenges); w.raw(composition_coeff.data(), 32); w.raw(z.data(), 32); w.raw(deep_alpha.data(), 32);
    w.felts(ood_trace); w.felts(ood_composition); w.felts(fri_alphas); w.felts(fri_remainder);
    w.u64(pow_nonce); w.u64s(query_positions);
    w.u64s(base_rows); w.u64s(extension_rows); w.u64s(composition_rows);
    w.bytes(base_paths); w.bytes(extension_paths); w.bytes(composition_paths);
    w.u32((uint32_t)fri_layers.size());
    for (auto &l : fri_layers) { w.raw(l.root.data(), 33); w.u32(l.log_len); w.u64s(l.positions); w.u64s(l.rows); w.bytes(l.paths); }
    return w.b;
}

// ---- the reference's wire format (layout: sandstorm_amd/wire.py)
namespace {
struct Wire {
    std::vector<uint8_t> b;
    void u8(uint8_t v) { b.push_back(v); }
    void u64(uint64_t v) { for (int i = 0; i < 8; ++i) b.push_back((uint8_t)(v >> (8 * i))); }
    bool friendly = false;                          // FriendlyMerkleTree: MixedMerkleDigest / FriendlyMerkleTreeProof encodings
    void digest(const uint8_t *d) { u64(32); b.insert(b.end(), d, d + 32); }
    void pedersen(const uint8_t *be_bytes) { for (int i = 31; i >= 0; --i) b.push_back(be_bytes[i]); }     // Fp, little-endian canonical
    void mixed(const uint8_t *d, uint8_t tag) {     // crypto/src/merkle/mixed.rs:46-61
        u8(tag);
        if (tag == 0) pedersen(d); else digest(d);
    }
    void root(const uint8_t *d33) { if (friendly) mixed(d33, d33[32]); else digest(d33); }
    void fp(const Felt &mont) {                     // 32-byte little-endian canonical value
        const auto be = canonical_be_bytes(mont);
        for (int i = 31; i >= 0; --i) b.push_back(be[i]);
    }
    void fp_limbs(const uint64_t *limbs) { Felt f; memcpy(f.data(), limbs, 32); fp(f); }
    void fp_vec(const std::vector<Felt> &v) { u64(v.size()); for (auto &f : v) fp(f); }
    void fp_vec(const std::vector<uint64_t> &limbs) { u64(limbs.size() / 4); for (size_t i = 0; i < limbs.size(); i += 4) fp_limbs(&limbs[i]); }
    // openings of `npos` positions: rows (npos x ncols felts), paths (npos x depth x 32, leaf level first),
    // leaves (npos x 32 row digests; empty for a single-column tree)
    void openings(const std::vector<uint64_t> &rows, const std::vector<uint8_t> &paths, const std::vector<uint8_t> &leaves, size_t npos,
                  const std::vector<uint8_t> &tags = std::vector<uint8_t>()) {
        u64(npos);
        if (!npos) return;
        const size_t depth = paths.size() / (32 * npos), ncols = rows.size() / (4 * npos);
        if (!depth || paths.size() != npos * depth * 32) throw std::runtime_error("wire: malformed authentication paths");
        const bool single = ncols == 1;
        if (!single && leaves.size() != 32 * npos) throw std::runtime_error("wire: row digests missing");
        if (friendly && !single && tags.size() != npos * depth) throw std::runtime_error("wire: path tags missing");
        for (size_t q = 0; q < npos; ++q) {
            const uint8_t *path = paths.data() + q * depth * 32;
            u8(single ? 1 : 0);
            u64(depth - 1);
            for (size_t l = 1; l < depth; ++l) {
                if (!friendly) digest(path + 32 * l);
                else if (single) pedersen(path + 32 * l);                 // MerkleView<PedersenDigest, Fp> (merkle/mod.rs:172)
                else mixed(path + 32 * l, tags[q * depth + l]);
            }
            if (single) {
                // the sibling's leaf slot holds the element as big-endian Montgomery bytes
                Felt sib{};
                for (int k = 0; k < 4; ++k)
                    for (int j = 0; j < 8; ++j) sib[k] |= (uint64_t)path[31 - (8 * k + j)] << (8 * j);
                fp(sib);
                fp_limbs(&rows[4 * q]);
            } else {
                digest(path);
                digest(leaves.data() + 32 * q);
            }
        }
    }
};
}  // namespace
std::vector<uint8_t> Proof::serialize_wire() const {
    const uint32_t o[5] = {options.num_queries, options.lde_blowup_factor, options.grinding_factor, options.fri_folding_factor,
                           options.fri_max_remainder_coeffs};
    Wire w;
    w.friendly = tree_kind == SS_TREE_FRIENDLY;
    for (uint32_t v : o) { if (v > 255) throw std::runtime_error("wire: proof options are single bytes"); w.u8((uint8_t)v); }
    w.u64(trace_len);
    w.root(base_root.data());
    w.u8(has_extension ? 1 : 0);
    if (has_extension) w.root(extension_root.data());
    w.root(composition_root.data());
    w.u64(fri_layers.size());
    for (auto &l : fri_layers) {
        w.fp_vec(l.rows);
        w.openings(l.rows, l.paths, l.leaves, l.positions.size(), l.path_tags);
        w.root(l.root.data());
    }
    w.fp_vec(fri_remainder);
    w.u64(pow_nonce);
    w.fp_vec(base_rows); w.fp_vec(extension_rows); w.fp_vec(composition_rows);
    const size_t nq = query_positions.size();
    w.openings(base_rows, base_paths, base_leaves, nq, base_path_tags);
    w.openings(extension_rows, extension_paths, extension_leaves, has_extension ? nq : 0, extension_path_tags);
    w.openings(composition_rows, composition_paths, composition_leaves, nq, composition_path_tags);
    w.fp_vec(ood_trace); w.fp_vec(ood_composition);
    return w.b;
}

}  // namespace ssh
