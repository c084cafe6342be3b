// mini_air_lib.cpp — TEST INFRASTRUCTURE: the small valid "mini" AIR of the end-to-end tests (mirror of tests/mini_air.py) as an `Air` of
// the C++ host, built into tests/_build/libsandstorm_test_air.so (tests/mini_air_host.py) and handed to the product's prover
// through the `ssh_air` handle.  The product library holds the layouts' AIRs only (host/air_recursive.cpp, air_starknet.cpp).
#include <algorithm>
#include <cstring>
#include <random>
#include <set>
#include <stdexcept>

#include "../../sandstorm_amd/host/prover.hpp"

namespace ssh {

static void ok(ss_status s) {
    if (s != SS_OK) throw std::runtime_error(ss_last_error());
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
}  // namespace ssh

extern "C" int sst_mini_air_create(ss_ctx *ctx, void **out) {          // -> an `ssh_air` handle (ssh_air_destroy frees it)
    try { *out = new ssh::MiniAir(ctx); return 0; } catch (const std::exception &) { return 1; }
}
