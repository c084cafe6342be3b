// air_layout.hpp — what the layout AIRs (air_recursive.cpp, air_starknet.cpp) share: the expression wrapper over the
// hash-consed Graph, domains as zerofier factor lists, the table registry (periodic columns, periodic zerofier
// multipliers, full-length inverse tables), the composition's grouping by domain, the lowering / out-of-domain
// evaluation entry points of `Air`, and the curve helpers the periodic columns are made of.
// Mirror of sandstorm_amd/layouts/recursive.py (Domain, Tables, composition, mask, make_air).
#pragma once
#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "prover.hpp"
#include "public_input.hpp"

namespace ssh {
namespace layout {

inline void ok(ss_status s) {
    if (s != SS_OK) throw std::runtime_error(ss_last_error());
}

// ---- expression wrapper over the hash-consed Graph
struct E {
    Graph *g;
    int id;
};
inline E operator+(const E &a, const E &b) { return E{a.g, a.g->add(a.id, b.id)}; }
inline E operator-(const E &a, const E &b) { return E{a.g, a.g->sub(a.id, b.id)}; }
inline E operator*(const E &a, const E &b) { return E{a.g, a.g->mul(a.id, b.id)}; }

// ---- domains: multiplier prod(num) / prod(den), factor (p, e) = X^p - g^e
struct Factor {
    uint64_t p, e;
    bool operator<(const Factor &o) const { return p != o.p ? p < o.p : e < o.e; }
    bool operator==(const Factor &o) const { return p == o.p && e == o.e; }
};
struct Domain { std::vector<Factor> num, den; };

struct TableSpec {
    int kind;                               // 0 periodic column number e, 2 periodic multiplier, 3 full-length inverse 1 / (X - g^e)
    std::vector<Factor> num, den;           // kind 2
    uint64_t e = 0;
    bool operator<(const TableSpec &o) const {
        if (kind != o.kind) return kind < o.kind;
        if (e != o.e) return e < o.e;
        if (num != o.num) return num < o.num;
        return den < o.den;
    }
};

// ---- the curve y^2 = x^3 + x + beta (builtins/src/utils.rs:134-181)
struct Pt { Felt x, y; };
inline Pt ec_double(const Pt &p) {
    const Felt xx = felt_mul(p.x, p.x);
    const Felt lam = felt_mul(felt_add(felt_add(felt_add(xx, xx), xx), felt_from_u64(1)), felt_inv(felt_add(p.y, p.y)));
    const Felt x3 = felt_sub(felt_mul(lam, lam), felt_add(p.x, p.x));
    return Pt{x3, felt_sub(felt_mul(lam, felt_sub(p.x, x3)), p.y)};
}
Felt pedersen_coord(int point, int which);                   // builtins/src/pedersen/constants.rs:5-30: P0 (shift point), P1..P4
std::vector<Felt> pedersen_column(int which);                // the 512 values of a Pedersen periodic column
std::vector<Felt> interpolate(std::vector<Felt> values);     // coefficients of the interpolant over <w_m>, m a power of two

// ---- a layout's AIR: the derived class supplies the constraints (composition) and its periodic columns
class LayoutAir : public Air {
public:
    AirProgramData build_program(uint64_t n, const std::vector<Felt> &ch, const Felt &alpha) override;
    void prepare_program(uint64_t n, const std::vector<Felt> &ch) override;
    Felt composition_at(uint64_t n, const std::vector<Felt> &ch, const Felt &alpha, const Felt &z, const std::vector<Felt> &ood) override;
    // flat description of the tables for host-side checks: per table kind, e, #num, (p, e)..., #den, (p, e)...
    std::vector<uint64_t> describe_tables() const;

private:
    // the program lowered for `ch` with a placeholder composition coefficient, and where its powers sit among the constants
    struct Prepared { bool valid = false; uint64_t n = 0; std::vector<Felt> ch; AirProgramData pd; std::vector<uint32_t> alpha_slot; } prepared_;
protected:
    LayoutAir(ss_ctx *ctx, const AirPublicInput &pi, uint32_t log_n, uint32_t lb, uint64_t lde_offset)
        : ctx_(ctx), pi_(pi), log_n_(log_n), lb_(lb), offset_(lde_offset), n_(1ull << log_n), g_(root_of_unity(log_n)) {}
    // registers the periodic columns and every table the composition refers to (their set and order do not depend on the
    // challenges), collects the mask, builds the tables on the device when there is one; call at the end of the constructor
    void finish_construction();

    virtual int composition(Graph &g, const std::vector<Felt> &ch, const Felt &alpha) = 0;
    virtual size_t num_periodic_columns() const = 0;
    virtual std::vector<Felt> column_values(size_t column) const = 0;    // over one period
    virtual uint64_t column_period(size_t column) const = 0;             // in trace rows

    Domain every(uint64_t k) const { return Domain{{}, {{n_ / k, 0}}}; }
    Domain every_except_last(uint64_t k) const { return Domain{{{1, n_ - k}}, {{n_ / k, 0}}}; }
    Domain row_from_end(uint64_t k) const { return Domain{{}, {{1, n_ - k}}}; }
    Factor F(uint64_t d, uint64_t k = 0, uint64_t m = 1) const { return Factor{n_ / d, k * n_ / m}; }   // X^(n/d) - g^(k n / m)

    int table_index(const TableSpec &s);
    E multiplier(Graph &g, const Domain &d);
    E column(Graph &g, size_t c) { return E{&g, g.table((uint32_t)c)}; }

    // sum_i alpha^i numerator_i * multiplier(domain_i), grouped by domain in first-use order
    class Composer {
    public:
        Composer(LayoutAir &air, Graph &g, const Felt &alpha) : air_(air), g_(g), alpha_(alpha), apow_(felt_from_u64(1)) {}
        void add(const std::string &domain_name, const Domain &d, const E &numerator);
        int total();
    private:
        struct Group { std::string name; Domain d; int sum; };
        LayoutAir &air_;
        Graph &g_;
        Felt alpha_, apow_;
        uint64_t count_ = 0;
        std::vector<Group> groups_;
    };

    Felt table_value_at(const TableSpec &s, const Felt &x) const;       // the function a table tabulates
    void build_tables();

    ss_ctx *ctx_;
    AirPublicInput pi_;
    uint32_t log_n_, lb_;
    uint64_t offset_, n_;
    Felt g_;
    std::vector<TableSpec> specs_;
    std::map<TableSpec, int> table_ix_;
    std::vector<uint32_t> desc_;
    std::unique_ptr<DeviceBuffer> tables_;
    mutable std::map<size_t, std::vector<Felt>> column_coeffs_;
};

// compute_public_memory_quotient (layouts/src/utils.rs:14-46) and compute_diluted_cumulative_value (utils.rs:48-108)
Felt public_memory_quotient(const AirPublicInput &pi, const Felt &z, const Felt &alpha, uint64_t trace_len, uint64_t public_memory_step);
Felt diluted_cumulative_value(const Felt &z, const Felt &alpha);          // 16 bits, spacing 4

}  // namespace layout
}  // namespace ssh
