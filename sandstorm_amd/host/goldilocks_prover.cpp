// goldilocks_prover.cpp — the 64-bit field's claim (cli/src/main.rs:103-133: p = 2^64 - 2^32 + 1, challenges in Fq3) in the C++ host:
// the mirror of sandstorm_amd/goldilocks.py's Prover, stage by stage and array for array (tests hold the two to each other: the
// same proof arrays from the same statement).  PARITY UNPINNED, as the Python module's header says: the reference instantiates this
// claim from un-vendored parts; Options.hash = "sha256" takes the parts it NAMES (MatrixMerkleTreeImpl<Sha256HashFn> trees, a coin
// with SHA-256 inside), "blake2s" this library's own choice.  Every stage is a HIP kernel behind the C ABI (the *_gl64* entry
// points); what the layout decides - the extension columns, the lowered composition program for the drawn challenges - comes
// from the caller through two callbacks, as the 252-bit Prover takes its extension builder.
#include "goldilocks_prover.hpp"

#include <algorithm>
#include <cstring>
#include <stdexcept>

namespace ssh {
namespace gl {

static void ok(ss_status s) {
    if (s != SS_OK) throw std::runtime_error(ss_last_error());
}

// ---- the field on the host (a handful of values per proof)
static const uint64_t P = 0xFFFFFFFF00000001ull;
static uint64_t mulm(uint64_t a, uint64_t b) { return (uint64_t)((unsigned __int128)a * b % P); }
static uint64_t addm(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a + b) % P); }
static uint64_t powm(uint64_t a, uint64_t e) { uint64_t r = 1; for (; e; e >>= 1) { if (e & 1) r = mulm(r, a); a = mulm(a, a); } return r; }
static uint64_t invm(uint64_t a) { return powm(a, P - 2); }
Fq3 mul3(const Fq3 &a, const Fq3 &b) {                     // X^3 = 2
    const uint64_t d0 = mulm(a[0], b[0]), d1 = addm(mulm(a[0], b[1]), mulm(a[1], b[0]));
    const uint64_t d2 = addm(addm(mulm(a[0], b[2]), mulm(a[1], b[1])), mulm(a[2], b[0]));
    const uint64_t d3 = addm(mulm(a[1], b[2]), mulm(a[2], b[1])), d4 = mulm(a[2], b[2]);
    return Fq3{addm(d0, addm(d3, d3)), addm(d1, addm(d4, d4)), d2};
}
static Fq3 pow3(Fq3 a, uint64_t e) { Fq3 r{1, 0, 0}; for (; e; e >>= 1) { if (e & 1) r = mul3(r, a); a = mul3(a, a); } return r; }

// ---- SHA-256 (FIPS 180-4) for the coin's host side; the trees' SHA-256 is the device's (csrc/hash.hip)
static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
Digest sha256(const uint8_t *msg, size_t len) {
    static const uint32_t K[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u,
        0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
        0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u,
        0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
        0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
        0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    std::vector<uint8_t> m(msg, msg + len);
    m.push_back(0x80);
    while (m.size() % 64 != 56) m.push_back(0);
    for (int i = 7; i >= 0; --i) m.push_back((uint8_t)(((uint64_t)len * 8) >> (8 * i)));
    for (size_t off = 0; off < m.size(); off += 64) {
        uint32_t w[64];
        for (int t = 0; t < 16; ++t) w[t] = ((uint32_t)m[off + 4 * t] << 24) | ((uint32_t)m[off + 4 * t + 1] << 16) | ((uint32_t)m[off + 4 * t + 2] << 8) | m[off + 4 * t + 3];
        for (int t = 16; t < 64; ++t)
            w[t] = w[t - 16] + (rotr(w[t - 15], 7) ^ rotr(w[t - 15], 18) ^ (w[t - 15] >> 3)) + w[t - 7] + (rotr(w[t - 2], 17) ^ rotr(w[t - 2], 19) ^ (w[t - 2] >> 10));
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int t = 0; t < 64; ++t) {
            const uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[t] + w[t];
            const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    Digest out;
    for (int i = 0; i < 8; ++i) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
    return out;
}

// ---- the coin (goldilocks.py Coin over coin.PublicCoin's Solidity shape: solidity.rs:54-118 with H = Keccak-256 or SHA-256)
namespace {
struct Coin {
    Digest digest;
    uint64_t counter = 0;
    bool sha;
    Coin(const Digest &seed, bool sha_) : digest(seed), sha(sha_) {}
    Digest h(const std::vector<uint8_t> &m) const { return sha ? sha256(m.data(), m.size()) : keccak256(m.data(), m.size()); }
    void reseed_with_bytes(const uint8_t *data, size_t len) {
        std::vector<uint8_t> m(digest.begin(), digest.end());                  // be32(digest + 1)
        for (int i = 31; i >= 0; --i) { if (++m[i] != 0) break; }
        m.insert(m.end(), data, data + len);
        digest = h(m);
        counter = 0;
    }
    void reseed_with_digest(const Digest &d) { reseed_with_bytes(d.data(), 32); }
    void reseed_with_fq3s(const uint64_t *vals, size_t n3) {                   // little-endian 8-byte coordinates
        std::vector<uint8_t> b(8 * n3);
        for (size_t i = 0; i < n3; ++i) for (int k = 0; k < 8; ++k) b[8 * i + k] = (uint8_t)(vals[i] >> (8 * k));
        reseed_with_bytes(b.data(), b.size());
    }
    void reseed_with_int(uint64_t v) {
        uint8_t b[8];
        for (int k = 0; k < 8; ++k) b[k] = (uint8_t)(v >> (8 * (7 - k)));
        reseed_with_bytes(b, 8);
    }
    Digest draw_bytes() {
        std::vector<uint8_t> m(digest.begin(), digest.end());
        uint8_t c[32] = {0};
        for (int k = 0; k < 8; ++k) c[31 - k] = (uint8_t)(counter >> (8 * k));
        m.insert(m.end(), c, c + 32);
        ++counter;
        return h(m);
    }
    uint64_t draw_felt() {                                                     // the first of the draw's four big-endian words below p
        for (;;) {
            const Digest d = draw_bytes();
            for (int k = 0; k < 4; ++k) {
                uint64_t v = 0;
                for (int j = 0; j < 8; ++j) v = (v << 8) | d[8 * k + j];
                if (v < P) return v;
            }
        }
    }
    Fq3 draw_fq3() { Fq3 r; r[0] = draw_felt(); r[1] = draw_felt(); r[2] = draw_felt(); return r; }
    std::vector<uint64_t> draw_queries(size_t max_n, uint64_t domain) {
        std::vector<uint64_t> vals;
        while (vals.size() < max_n) {
            const Digest d = draw_bytes();
            for (int k = 0; k < 4 && vals.size() < max_n; ++k) {
                uint64_t v = 0;
                for (int j = 0; j < 8; ++j) v = (v << 8) | d[8 * k + j];
                vals.push_back(v % domain);
            }
        }
        std::sort(vals.begin(), vals.end());
        vals.erase(std::unique(vals.begin(), vals.end()), vals.end());
        return vals;
    }
};
void be64(std::vector<uint8_t> &b, uint64_t v) { for (int k = 7; k >= 0; --k) b.push_back((uint8_t)(v >> (8 * k))); }
}  // namespace

Digest transcript_seed(const Digest &seed, const Options &opt, uint64_t trace_len, const Digest &statement_digest) {
    std::vector<uint8_t> b(seed.begin(), seed.end());
    for (uint64_t v : {(uint64_t)opt.num_queries, (uint64_t)opt.log_blowup, (uint64_t)opt.grinding, (uint64_t)opt.fold, (uint64_t)opt.max_remainder, trace_len}) be64(b, v);
    b.insert(b.end(), statement_digest.begin(), statement_digest.end());
    if (opt.sha256) { const char *name = "sha256"; b.insert(b.end(), name, name + 6); }
    return keccak256(b.data(), b.size());
}

namespace {
struct Dev {                                                // a device buffer of the context's pool
    ss_ctx *ctx;
    void *p = nullptr;
    Dev(ss_ctx *c, size_t bytes) : ctx(c) { ok(ss_dev_alloc(c, bytes ? bytes : 8, &p)); }
    ~Dev() { if (p) ss_dev_free(ctx, p); }
    Dev(const Dev &) = delete;
    Dev &operator=(const Dev &) = delete;
    uint64_t *u64() const { return (uint64_t *)p; }
    uint8_t *u8() const { return (uint8_t *)p; }
};
using DevP = std::unique_ptr<Dev>;
struct Commitment { DevP nodes; Digest root; };
}  // namespace

Proof prove(ss_ctx *ctx, const Options &opt, const Digest &seed, const Digest &statement_digest, const std::vector<const uint64_t *> &base_cols,
            uint64_t n, const std::vector<std::pair<uint32_t, uint32_t>> &mask, uint32_t num_challenges, uint32_t num_ext,
            const ExtensionBuilder &build_extension, const ProgramBuilder &build_program) {
    if (!n || (n & (n - 1))) throw std::runtime_error("trace length: a power of two");
    if (opt.log_blowup != 1) throw std::runtime_error("the composition split (even / odd coefficients) is written for blowup 2");
    uint32_t log_n = 0;
    while ((1ull << log_n) < n) ++log_n;
    const uint32_t lb = opt.log_blowup;
    const uint64_t N = n << lb, OFFSET = 7;
    const int row_hash = opt.sha256 ? SS_HASH_SHA256 : SS_HASH_BLAKE2S, tree = opt.sha256 ? SS_TREE_SHA256 : SS_TREE_BLAKE2S;
    Coin coin(transcript_seed(seed, opt, n, statement_digest), opt.sha256);
    Proof proof;
    proof.trace_len = n;

    auto commit = [&](const std::vector<const uint64_t *> &segs, uint32_t seg_len, uint64_t rows) {
        Dev dig(ctx, 32 * rows);
        Commitment c;
        c.nodes.reset(new Dev(ctx, 64 * rows));
        ok(ss_hash_rows_gl64(ctx, row_hash, segs.data(), (uint32_t)segs.size(), seg_len, rows, dig.u8()));
        uint8_t root[33];
        ok(ss_merkle_build(ctx, tree, 0, SS_LEAF_DIGEST, dig.p, rows, c.nodes->u8(), nullptr, root));
        memcpy(c.root.data(), root, 32);
        return c;
    };
    auto open = [&](const std::vector<const uint64_t *> &segs, uint32_t seg_len, const Commitment &c, uint64_t rows, const std::vector<uint64_t> &positions) {
        Opening o;
        o.width = (uint32_t)segs.size() * seg_len;
        o.rows.assign(positions.size() * o.width, 0);
        ok(ss_gather_rows_gl64(ctx, segs.data(), (uint32_t)segs.size(), seg_len, rows, positions.data(), (uint32_t)positions.size(), o.rows.data()));
        uint32_t depth = 0;
        while ((1ull << depth) < rows) ++depth;
        o.depth = depth;
        o.paths.assign(positions.size() * depth * 32, 0);
        ok(ss_merkle_open(ctx, c.nodes->u8(), nullptr, rows, positions.data(), (uint32_t)positions.size(), o.paths.data(), nullptr));
        return o;
    };
    std::vector<DevP> keep;                                 // every column of the proof lives to its end
    auto extend = [&](const std::vector<const uint64_t *> &cols, std::vector<const uint64_t *> *ev, std::vector<const uint64_t *> *co) {
        std::vector<uint64_t *> e, c;
        for (size_t i = 0; i < cols.size(); ++i) {
            keep.emplace_back(new Dev(ctx, 8 * N)); e.push_back(keep.back()->u64());
            keep.emplace_back(new Dev(ctx, 8 * n)); c.push_back(keep.back()->u64());
        }
        if (!cols.empty()) ok(ss_lde_gl64(ctx, cols.data(), (uint32_t)cols.size(), log_n, lb, OFFSET, e.data(), c.data()));
        ev->insert(ev->end(), e.begin(), e.end());
        co->insert(co->end(), c.begin(), c.end());
    };
    // 1-2. base and extension traces
    std::vector<const uint64_t *> trace_ev, trace_co;
    extend(base_cols, &trace_ev, &trace_co);
    const size_t nbase = base_cols.size();
    Commitment base_com = commit(std::vector<const uint64_t *>(trace_ev.begin(), trace_ev.begin() + nbase), 1, N);
    proof.base_root = base_com.root;
    coin.reseed_with_digest(proof.base_root);
    std::vector<Fq3> challenges;
    for (uint32_t i = 0; i < num_challenges; ++i) challenges.push_back(coin.draw_fq3());
    Commitment ext_com;
    if (num_ext) {
        const std::vector<const uint64_t *> ext_cols = build_extension(challenges);
        if (ext_cols.size() != num_ext) throw std::runtime_error("the extension builder returned another number of coordinate columns");
        extend(ext_cols, &trace_ev, &trace_co);
        ext_com = commit(std::vector<const uint64_t *>(trace_ev.begin() + nbase, trace_ev.end()), 1, N);
        proof.has_ext = true;
        proof.ext_root = ext_com.root;
        coin.reseed_with_digest(proof.ext_root);
    }
    // 3-4. composition: the lowered program over the LDE domain, then H = H0(x^2) + x H1(x^2) as six coordinate columns
    const Fq3 alpha = coin.draw_fq3();
    const ProgramData pd = build_program(challenges, alpha);
    ss_air_program prog;
    memset(&prog, 0, sizeof prog);
    static const uint64_t no_consts[3] = {0, 0, 0};
    static const uint32_t no_desc[2] = {0, 0};
    prog.code = pd.code.data(); prog.n_instr = (uint32_t)(pd.code.size() / 2);
    prog.consts = pd.consts.empty() ? no_consts : pd.consts.data(); prog.n_consts = (uint32_t)(pd.consts.size() / 3);
    prog.d_tables = pd.d_tables; prog.table_desc = pd.table_desc.empty() ? no_desc : pd.table_desc.data(); prog.n_tables = (uint32_t)(pd.table_desc.size() / 2);
    prog.n_slots = pd.n_slots;
    std::vector<const uint64_t *> comp_co, comp_ev;
    {
        Dev q(ctx, 24 * N);
        ok(ss_eval_quotient_gl64x3(ctx, &prog, trace_ev.data(), (uint32_t)trace_ev.size(), log_n, lb, OFFSET, q.u64()));
        // de-interleave the three coordinates, interpolate each over the coset (bit-reversed coefficients: the even ones first)
        std::vector<uint64_t *> qc;
        std::vector<DevP> qbuf;
        for (int t = 0; t < 3; ++t) {
            qbuf.emplace_back(new Dev(ctx, 8 * N));
            qc.push_back(qbuf.back()->u64());
            ok(ss_dev_copy_2d(ctx, qc[t], 8, q.u8() + 8 * t, 24, 8, N));
        }
        ok(ss_ntt_gl64(ctx, qc.data(), 3, log_n + lb, SS_NTT_INVERSE, OFFSET, SS_ORDER_NATURAL, SS_ORDER_BITREV));
        for (int half = 0; half < 2; ++half)
            for (int t = 0; t < 3; ++t) {
                keep.emplace_back(new Dev(ctx, 8 * n));
                ok(ss_dev_copy(ctx, keep.back()->p, qc[t] + half * n, 8 * n));
                comp_co.push_back(keep.back()->u64());
            }
        for (const uint64_t *c : comp_co) {                 // zero-padded to N in bit-reversed order, then the forward transform
            keep.emplace_back(new Dev(ctx, 8 * N));
            uint64_t *padded = keep.back()->u64();
            ok(ss_dev_zero(ctx, padded, 8 * N));
            ok(ss_dev_copy_2d(ctx, padded, 8u << lb, c, 8, 8, n));
            ok(ss_ntt_gl64(ctx, &padded, 1, log_n + lb, SS_NTT_FORWARD, OFFSET, SS_ORDER_BITREV, SS_ORDER_NATURAL));
            comp_ev.push_back(padded);
        }
    }
    Commitment comp_com = commit(comp_ev, 1, N);
    proof.comp_root = comp_com.root;
    coin.reseed_with_digest(proof.comp_root);
    // 5. out-of-domain evaluations
    const Fq3 z = coin.draw_fq3(), zc = pow3(z, 2);
    std::vector<uint32_t> mc, mo;
    for (auto &c : mask) { mc.push_back(c.first); mo.push_back(c.second); }
    proof.ood_trace.assign(3 * mask.size(), 0);
    ok(ss_ood_eval_gl64x3(ctx, trace_co.data(), (uint32_t)trace_co.size(), log_n, mc.data(), mo.data(), (uint32_t)mask.size(), z.data(), proof.ood_trace.data()));
    const uint32_t six[6] = {0, 1, 2, 3, 4, 5}, zeros[6] = {0, 0, 0, 0, 0, 0};
    proof.ood_comp.assign(18, 0);
    ok(ss_ood_eval_gl64x3(ctx, comp_co.data(), 6, log_n, six, zeros, 6, zc.data(), proof.ood_comp.data()));
    {
        std::vector<uint64_t> all(proof.ood_trace);
        all.insert(all.end(), proof.ood_comp.begin(), proof.ood_comp.end());
        coin.reseed_with_fq3s(all.data(), all.size());
    }
    // 6. DEEP composition
    const Fq3 gamma = coin.draw_fq3();
    std::vector<uint64_t> coefs;
    {
        Fq3 cur{1, 0, 0};
        for (size_t i = 0; i < mask.size() + 6; ++i) { coefs.insert(coefs.end(), cur.begin(), cur.end()); cur = mul3(cur, gamma); }
    }
    DevP layer(new Dev(ctx, 24 * N));
    ok(ss_deep_compose_gl64x3(ctx, trace_ev.data(), (uint32_t)trace_ev.size(), comp_ev.data(), 6, log_n, lb, OFFSET, mc.data(), mo.data(), (uint32_t)mask.size(),
                              proof.ood_trace.data(), coefs.data(), proof.ood_comp.data(), coefs.data() + 3 * mask.size(), z.data(), zc.data(), layer->u64()));
    // 7. FRI
    uint32_t n_layers = 0;
    for (uint64_t len = N; len > ((uint64_t)opt.max_remainder << lb); len /= opt.fold) {
        if (len % opt.fold) throw std::runtime_error("a FRI layer's length is not a multiple of the folding factor");
        ++n_layers;
    }
    uint32_t log_fold = 0;
    while ((1u << log_fold) < opt.fold) ++log_fold;
    struct Layer { DevP evals; Commitment com; uint64_t rows; std::vector<const uint64_t *> segs; };
    std::vector<Layer> layers;
    uint32_t ll = log_n + lb;
    uint64_t off = OFFSET;
    for (uint32_t i = 0; i < n_layers; ++i) {
        Layer L;
        L.rows = (1ull << ll) / opt.fold;
        for (uint32_t k = 0; k < opt.fold; ++k) L.segs.push_back(layer->u64() + 3 * k * L.rows);
        L.com = commit(L.segs, 3, L.rows);
        coin.reseed_with_digest(L.com.root);
        const Fq3 a = coin.draw_fq3();
        DevP next(new Dev(ctx, 24 * L.rows));
        ok(ss_fri_fold_gl64x3(ctx, layer->u64(), ll, opt.fold, a.data(), off, 0, next->u64()));
        FriLayer fl;
        fl.root = L.com.root; fl.log_len = ll;
        proof.fri_layers.push_back(fl);
        L.evals = std::move(layer);
        layers.push_back(std::move(L));
        layer = std::move(next);
        ll -= log_fold;
        off = powm(off, opt.fold);
    }
    // remainder: interpolate the last layer (tiny) on the host
    {
        const uint64_t L = 1ull << ll;
        std::vector<uint64_t> last(3 * L);
        ok(ss_download(ctx, last.data(), layer->p, 24 * L));
        const uint64_t winv = invm(powm(7, (P - 1) >> ll)), oinv = invm(off), linv = invm(L % P);
        proof.remainder.assign(3 * L, 0);
        for (int t = 0; t < 3; ++t)
            for (uint64_t k = 0; k < L; ++k) {               // coefficient k = (1/L) sum_j v_j (w^-k)^j * off^-k
                uint64_t acc = 0, cur = 1;
                const uint64_t wk = powm(winv, k);
                for (uint64_t j = 0; j < L; ++j) { acc = addm(acc, mulm(last[3 * j + t], cur)); cur = mulm(cur, wk); }
                proof.remainder[3 * k + t] = mulm(mulm(acc, linv), powm(oinv, k));
            }
        for (uint64_t k = L >> lb; k < L; ++k)
            for (int t = 0; t < 3; ++t)
                if (proof.remainder[3 * k + t]) throw std::runtime_error("the FRI remainder is not of low degree: the DEEP composition is not the polynomial it must be");
        coin.reseed_with_fq3s(proof.remainder.data(), proof.remainder.size());
    }
    // 8. proof of work, queries, openings
    proof.pow_nonce = 0;
    if (opt.grinding) ok(ss_pow_grind(ctx, SS_COIN_SOLIDITY, coin.digest.data(), opt.grinding, &proof.pow_nonce));
    coin.reseed_with_int(proof.pow_nonce);
    const std::vector<uint64_t> positions = coin.draw_queries(opt.num_queries, N);
    proof.base = open(std::vector<const uint64_t *>(trace_ev.begin(), trace_ev.begin() + nbase), 1, base_com, N, positions);
    if (num_ext) proof.ext = open(std::vector<const uint64_t *>(trace_ev.begin() + nbase, trace_ev.end()), 1, ext_com, N, positions);
    proof.comp = open(comp_ev, 1, comp_com, N, positions);
    std::vector<uint64_t> pos = positions;
    for (size_t i = 0; i < layers.size(); ++i) {
        for (uint64_t &p : pos) p %= layers[i].rows;
        std::sort(pos.begin(), pos.end());
        pos.erase(std::unique(pos.begin(), pos.end()), pos.end());
        proof.fri_layers[i].opening = open(layers[i].segs, 3, layers[i].com, layers[i].rows, pos);
    }
    ok(ss_ctx_sync(ctx));
    return proof;
}

}  // namespace gl
}  // namespace ssh
