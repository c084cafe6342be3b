// airs.cpp — the small valid "mini" AIR used by the end-to-end tests (mirror of tests/mini_air.py)
// and the layout-SHAPED synthetic AIR that bench.py drives at sizes for which no valid trace exists
// (mirror of sandstorm_amd/synthetic_air.py).  The real `recursive` layout is in air_recursive.cpp.
#include <algorithm>
#include <cstring>
#include <random>
#include <set>
#include <stdexcept>

#include "prover.hpp"

namespace ssh {

static void ok(ss_status s) {
    if (s != SS_OK) throw std::runtime_error(ss_last_error());
}

Felt Air::composition_at(uint64_t, const std::vector<Felt> &, const Felt &, const Felt &, const std::vector<Felt> &) {
    throw std::runtime_error("the " + name + " AIR has no verifier side");
}

// ---------------------------------------------------------------------------- mini
class MiniAir : public Air {
public:
    explicit MiniAir(ss_ctx *ctx) : ctx_(ctx) {
        name = "mini"; num_base_columns = 2; num_extension_columns = 1; num_challenges = 1;
        mask = {{0, 0}, {0, 1}, {1, 0}, {1, 1}, {2, 0}, {2, 1}};
    }
    // the composition constraint (mirror of tests/mini_air.py::composition); table 0 = 1 / (X^n - 1)
    static int graph(Graph &g, uint64_t n, const std::vector<Felt> &ch, const Felt &alpha) {
        uint32_t log_n = 0;
        while ((1ull << log_n) < n) ++log_n;
        const Felt w = root_of_unity(log_n);
        const int X = g.x();
        const int last = g.sub(X, g.constant(felt_pow(w, n - 1)));
        const int inv_all = g.table(0);
        const int inv_first = g.inv(g.sub(X, g.constant_u64(1)));
        const int c0 = g.trace(0, 0), c0n = g.trace(0, 1), c1 = g.trace(1, 0), c1n = g.trace(1, 1), e0 = g.trace(2, 0), e0n = g.trace(2, 1);
        const int gamma = g.constant(ch[0]);
        const int ks[5] = {
            g.mul(g.mul(g.sub(c0n, c1), last), inv_all),
            g.mul(g.mul(g.sub(g.sub(c1n, g.mul(c0, c1)), c0), last), inv_all),
            g.mul(g.sub(c0, g.constant_u64(1)), inv_first),
            g.mul(g.mul(g.sub(e0n, g.mul(e0, g.add(gamma, c0n))), last), inv_all),
            g.mul(g.sub(e0, g.add(gamma, c0)), inv_first)};
        int total = -1;
        Felt ap = felt_from_u64(1);
        for (int k = 0; k < 5; ++k) {
            const int term = g.mul(ks[k], g.constant(ap));
            total = total < 0 ? term : g.add(total, term);
            ap = felt_mul(ap, alpha);
        }
        return total;
    }
    Felt composition_at(uint64_t n, const std::vector<Felt> &ch, const Felt &alpha, const Felt &z, const std::vector<Felt> &ood) override {
        Graph g;
        const int root = graph(g, n, ch, alpha);
        return evaluate(g, root, z, [&](uint32_t c, uint32_t o) {
            for (size_t j = 0; j < mask.size(); ++j) if (mask[j].first == c && mask[j].second == o) return ood[j];
            throw std::runtime_error("trace cell outside the mask");
        }, [&](uint32_t) { return felt_inv(felt_sub(felt_pow(z, n), felt_from_u64(1))); });
    }
    AirProgramData build_program(uint64_t n, const std::vector<Felt> &ch, const Felt &alpha) override {
        Graph g;
        const int total = graph(g, n, ch, alpha);
        AirProgramData pd;
        pd.program = lower(g, total);
        // table 0 = 1/(X^n - 1) on the blowup-2 coset: x_i^n = 3^n * (-1)^i
        const Felt gn = felt_pow(felt_from_u64(3), n), one = felt_from_u64(1);
        const Felt t0 = felt_inv(felt_sub(gn, one)), t1 = felt_inv(felt_sub(felt_neg(gn), one));
        tables_.reset(new DeviceBuffer(ctx_, 64));
        uint64_t host[8];
        memcpy(host, t0.data(), 32); memcpy(host + 4, t1.data(), 32);
        ok(ss_upload(ctx_, tables_->u64(), host, 64));
        pd.d_tables = tables_->u64();
        pd.table_desc = {0, 1};
        return pd;
    }
private:
    ss_ctx *ctx_;
    std::unique_ptr<DeviceBuffer> tables_;
};
std::unique_ptr<Air> make_mini_air(ss_ctx *ctx) { return std::unique_ptr<Air>(new MiniAir(ctx)); }

// ----------------------------------------------------------------------- synthetic
namespace {
const std::vector<std::vector<uint32_t>> RECURSIVE_MASK = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15},
    {0, 1, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32, 33, 64, 65, 88, 90, 92, 94, 96, 97, 120, 122, 124, 126},
    {0, 1},
    {0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 12, 13, 16, 26, 27, 42, 43, 58, 74, 75, 91, 122, 123, 154, 202, 522, 523, 1034, 1035, 2058},
    {0, 1, 2, 3},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 12, 28, 44, 60, 76, 92, 108, 124, 1021, 1023, 1025, 1027, 2045},
    {0, 1, 2, 3, 4, 5, 7, 9, 11, 13, 17, 25, 768, 772, 784, 788, 1004, 1008, 1022, 1024},
    {0, 1}, {0, 1}, {0, 1, 2, 5}};
const uint32_t STARKNET_CELLS[10] = {16, 5, 4, 9, 2, 60, 4, 56, 105, 8};
const uint32_t STARKNET_MAXOFF[10] = {15, 511, 256, 256, 255, 33158, 3, 1009, 32763, 15};
constexpr uint32_t N_POINT_ZEROFIERS = 10;
}  // namespace

class SyntheticAir : public Air {
public:
    SyntheticAir(ss_ctx *ctx, const std::string &layout, uint32_t log_n, uint32_t lb, uint64_t lde_offset) : ctx_(ctx), log_n_(log_n), lb_(lb) {
        const uint64_t n = 1ull << log_n, N = n << lb;
        std::vector<uint32_t> periods, periodic;
        if (layout == "recursive") {
            num_base_columns = 7; num_extension_columns = 3; ncons_ = 93;
            periods = {1, 2, 4, 16, 32, 128, 1024, 2048}; periodic = {2048, 2048};
            for (uint32_t c = 0; c < RECURSIVE_MASK.size(); ++c) for (uint32_t o : RECURSIVE_MASK[c]) if (o < n) mask.push_back({c, o});
        } else if (layout == "starknet") {
            num_base_columns = 9; num_extension_columns = 1; ncons_ = 195;
            periods = {1, 2, 4, 8, 16, 64, 128, 256, 512, 1024, 16384, 32768};
            periodic = {512, 512, 32768, 32768, 512, 512, 512, 64, 32};
            for (uint32_t c = 0; c < 10; ++c) {
                std::set<uint32_t> offs;
                std::mt19937 rng(1000 + c);
                for (uint32_t o = 0; o < std::max(1u, STARKNET_CELLS[c] / 2) && o <= STARKNET_MAXOFF[c]; ++o) offs.insert(o);
                offs.insert(STARKNET_MAXOFF[c]);
                while (offs.size() < STARKNET_CELLS[c] && offs.size() <= STARKNET_MAXOFF[c]) offs.insert(rng() % (STARKNET_MAXOFF[c] + 1));
                for (uint32_t o : offs) if (o < n) mask.push_back({c, o});
            }
        } else {
            throw std::runtime_error("unknown layout " + layout);
        }
        name = "synthetic-" + layout;
        num_challenges = 6;
        std::sort(mask.begin(), mask.end());
        n_zero_ = (uint32_t)periods.size(); n_per_ = (uint32_t)periodic.size();
        uint64_t off = 0;
        auto push = [&](uint64_t len) { uint32_t ll = 0; while ((1ull << ll) < len) ++ll; desc_.push_back((uint32_t)off); desc_.push_back(ll); off += len; };
        for (uint32_t p : periods) push(std::min<uint64_t>(N, (uint64_t)p << lb));
        for (uint32_t l : periodic) push(std::min<uint64_t>(N, (uint64_t)l << lb));
        const uint64_t random_len = off;
        for (uint32_t k = 0; k < N_POINT_ZEROFIERS; ++k) push(N);
        tables_.reset(new DeviceBuffer(ctx, 32 * off));
        std::vector<uint64_t> host(4 * random_len);
        std::mt19937_64 rng(7);
        for (uint64_t i = 0; i < random_len; ++i) { for (int k = 0; k < 4; ++k) host[4 * i + k] = rng(); host[4 * i + 3] &= (1ull << 59) - 1; }
        ok(ss_upload(ctx, tables_->u64(), host.data(), host.size() * 8));
        const Felt g = felt_from_u64(lde_offset), wn = root_of_unity(log_n);
        for (uint32_t k = 0; k < N_POINT_ZEROFIERS; ++k) {
            const Felt c = felt_pow(wn, (k * 7919ull) % n);
            ok(ss_inverse_table(ctx, log_n + lb, g.data(), c.data(), tables_->u64() + 4 * (random_len + (uint64_t)k * N)));
        }
        ok(ss_ctx_sync(ctx));
    }
    AirProgramData build_program(uint64_t n, const std::vector<Felt> &ch, const Felt &alpha) override {
        if (n != (1ull << log_n_)) throw std::runtime_error("synthetic AIR built for another trace length");
        Graph g;
        std::vector<int> cells, chs;
        for (auto &c : mask) cells.push_back(g.trace(c.first, c.second));
        for (auto &c : ch) chs.push_back(g.constant(c));
        std::mt19937_64 r(0xC0FFEE);
        auto pick = [&]() { return cells[r() % cells.size()]; };
        // composition = sum_k alpha^k C_k / Z_k, evaluated as sum_Z (1/Z) * (sum_{k: Z_k = Z} alpha^k C_k):
        // constraints sharing a zerofier are summed before the ONE multiplication by its inverse table
        // (195 constraints, 22 distinct zerofiers for starknet) — the same field element, a third fewer muls
        std::vector<std::pair<int, int>> groups;     // (zerofier-inverse table node, partial sum), first-use order
        Felt ap = felt_from_u64(1);
        for (uint32_t k = 0; k < ncons_; ++k) {
            const int a = pick(), b = pick(), c = pick();
            int body;
            if (k % 7 == 3) body = g.sub(g.mul(g.add(a, chs[k % chs.size()]), g.sub(b, g.table(n_zero_ + k % n_per_))), c);
            else if (k % 11 == 5) body = g.sub(g.mul(a, b), g.mul(c, chs[k % chs.size()]));
            else { Felt rc = {r(), r(), r(), r() & ((1ull << 59) - 1)}; body = g.add(g.sub(g.mul(a, b), c), g.constant(rc)); }
            const int zer = (k % 9 == 8) ? g.table(n_zero_ + n_per_ + (k / 9) % N_POINT_ZEROFIERS) : g.table(k % n_zero_);
            const int term = g.mul(body, g.constant(ap));
            auto it = std::find_if(groups.begin(), groups.end(), [&](const std::pair<int, int> &p) { return p.first == zer; });
            if (it == groups.end()) groups.push_back({zer, term});
            else it->second = g.add(it->second, term);
            ap = felt_mul(ap, alpha);
        }
        int total = -1;
        for (auto &gr : groups) {
            const int term = g.mul(gr.second, gr.first);
            total = total < 0 ? term : g.add(total, term);
        }
        int sum = cells[0];                          // every mask cell is read at least once
        for (size_t i = 1; i < cells.size(); ++i) sum = g.add(sum, cells[i]);
        total = g.add(total, g.mul(sum, g.table(0)));
        AirProgramData pd;
        pd.program = lower(g, total);
        pd.d_tables = tables_->u64();
        pd.table_desc = desc_;
        return pd;
    }
private:
    ss_ctx *ctx_;
    uint32_t log_n_, lb_, ncons_ = 0, n_zero_ = 0, n_per_ = 0;
    std::vector<uint32_t> desc_;
    std::unique_ptr<DeviceBuffer> tables_;
};
std::unique_ptr<Air> make_synthetic_air(ss_ctx *ctx, const std::string &layout, uint32_t log_n, uint32_t log_blowup, uint64_t lde_offset) {
    return std::unique_ptr<Air>(new SyntheticAir(ctx, layout, log_n, log_blowup, lde_offset));
}

}  // namespace ssh
