// air_layout.cpp — see air_layout.hpp
#include "air_layout.hpp"
#include <map>

namespace ssh {
namespace layout {

namespace {
// builtins/src/pedersen/constants.rs:5-30 (canonical little-endian limbs)
const uint64_t PEDERSEN_POINTS[5][2][4] = {
    {{0x551fde4050ca6804ull, 0x716b0b1022947733ull, 0x00ee1b87eb599f16ull, 0x049ee3eba8c16007ull}, {0xd0405d266e10268aull, 0x4e621062c0e056c1ull, 0xf346d49d06ea0ed3ull, 0x03ca0cfe4b3bc6ddull}},
    {{0x1080d17957ebe47bull, 0x8fa8120b6d56eb0cull, 0x969c748655fca9e5ull, 0x0234287dcbaffe7full}, {0x6ed0268ee89e5615ull, 0x940135dd7a6c94ccull, 0x1e889527d41f4e39ull, 0x03b056f100f96fb2ull}},
    {{0xb7a6932dba8aa378ull, 0x99099ec1de5e3018ull, 0x3f9dab2656558f33ull, 0x04fa56f376c83db3ull}, {0x5168f4e80ff5b54dull, 0x562761f92a7a23b4ull, 0x8113e0c0e47e4401ull, 0x03fa0984c931c9e3ull}},
    {{0x3aa372f0bd2d6997ull, 0x40c690c74709e90full, 0x764910f75b45f74bull, 0x04ba4cc166be8decull}, {0x48151f27b24b219cull, 0xcac5c59a5ce5ae7cull, 0x4b971e46c4ede85full, 0x0040301cf5c1751full}},
    {{0xd36ff12c49a58202ull, 0x2ca65048d53fb325ull, 0x6e44cca8f61a63bbull, 0x054302dcb0e6cc1cull}, {0x879dcc77e99c2426ull, 0xce98ad783c25561aull, 0xb348046268d8ae25ull, 0x01b77b3e37d13504ull}},
};
}  // namespace

Felt pedersen_coord(int point, int which) {
    Felt c;
    memcpy(c.data(), PEDERSEN_POINTS[point][which], 32);
    return felt_from_canonical(c);
}

// builtins/src/pedersen/periodic.rs:1211-1250: 2^i P1 (i < 248), 2^i P2 (i < 4), the last one repeated up to 256; then P3, P4
std::vector<Felt> pedersen_column(int which) {
    std::vector<Felt> out;
    for (int e = 0; e < 2; ++e) {
        std::vector<Pt> half;
        Pt acc{pedersen_coord(1 + 2 * e, 0), pedersen_coord(1 + 2 * e, 1)};
        for (int i = 0; i < 248; ++i) { half.push_back(acc); acc = ec_double(acc); }
        acc = Pt{pedersen_coord(2 + 2 * e, 0), pedersen_coord(2 + 2 * e, 1)};
        for (int i = 0; i < 4; ++i) { half.push_back(acc); acc = ec_double(acc); }
        for (int i = 0; i < 4; ++i) half.push_back(half[251]);
        for (auto &p : half) out.push_back(which ? p.y : p.x);
    }
    return out;
}

// plain O(m log m) inverse transform
std::vector<Felt> interpolate(std::vector<Felt> a) {
    const size_t m = a.size();
    uint32_t lg = 0;
    while ((1ull << lg) < m) ++lg;
    for (size_t i = 0; i < m; ++i) {
        size_t r = 0;
        for (uint32_t b = 0; b < lg; ++b) r |= ((i >> b) & 1) << (lg - 1 - b);
        if (r > i) std::swap(a[i], a[r]);
    }
    const Felt w_inv = felt_inv(root_of_unity(lg));
    for (size_t len = 2; len <= m; len <<= 1) {
        const Felt wl = felt_pow(w_inv, m / len);
        for (size_t s = 0; s < m; s += len) {
            Felt w = felt_from_u64(1);
            for (size_t k = 0; k < len / 2; ++k) {
                const Felt u = a[s + k], v = felt_mul(a[s + k + len / 2], w);
                a[s + k] = felt_add(u, v); a[s + k + len / 2] = felt_sub(u, v);
                w = felt_mul(w, wl);
            }
        }
    }
    const Felt m_inv = felt_inv(felt_from_u64(m));
    for (auto &v : a) v = felt_mul(v, m_inv);
    return a;
}

Felt public_memory_quotient(const AirPublicInput &pi, const Felt &z, const Felt &a, uint64_t trace_len, uint64_t public_memory_step) {
    const uint64_t s = trace_len / public_memory_step, count = pi.public_memory.size();
    Felt den = felt_from_u64(1);
    const MemoryEntry *pad = nullptr;
    for (auto &e : pi.public_memory) {
        den = felt_mul(den, felt_sub(z, felt_add(felt_mul(a, felt_from_canonical(e.value)), felt_from_u64(e.address))));
        if (!pad && e.address == 1) pad = &e;
    }
    if (!pad) throw std::runtime_error("public memory has no entry at address 1");
    den = felt_mul(den, felt_pow(felt_sub(z, felt_add(felt_mul(a, felt_from_canonical(pad->value)), felt_from_u64(1))), s - count));
    return felt_mul(felt_pow(z, s), felt_inv(den));
}

Felt diluted_cumulative_value(const Felt &dz, const Felt &da) {
    const Felt one = felt_from_u64(1), mult = felt_from_u64(16);
    Felt diff_x = felt_from_u64(14), p = felt_add(dz, one), q = one, x = one;
    for (int i = 1; i < 16; ++i) {
        x = felt_add(x, diff_x);
        diff_x = felt_mul(diff_x, mult);
        const Felt xp = felt_mul(x, p), y = felt_add(p, felt_mul(dz, xp));
        q = felt_add(q, felt_add(felt_mul(q, y), felt_mul(x, xp)));
        p = felt_mul(p, y);
    }
    return felt_add(p, felt_mul(q, da));
}

// ---- LayoutAir
void LayoutAir::finish_construction() {
    for (size_t c = 0; c < num_periodic_columns(); ++c) { TableSpec s; s.kind = 0; s.e = c; table_index(s); }
    std::vector<Felt> ch(num_challenges, felt_from_u64(2));
    Graph g;
    const int root = composition(g, ch, felt_from_u64(17));
    std::set<std::pair<uint32_t, uint32_t>> cells;
    std::vector<char> seen(g.nodes().size(), 0);
    std::vector<int> stack{root};
    while (!stack.empty()) {
        const int id = stack.back(); stack.pop_back();
        if (seen[id]) continue;
        seen[id] = 1;
        const Node &nd = g.nodes()[id];
        if (nd.kind == NodeKind::Trace) cells.insert({nd.p0, nd.p1});
        if (nd.a >= 0) stack.push_back(nd.a);
        if (nd.b >= 0) stack.push_back(nd.b);
    }
    mask.assign(cells.begin(), cells.end());
    if (ctx_) build_tables();
}

AirProgramData LayoutAir::build_program(uint64_t n, const std::vector<Felt> &ch, const Felt &alpha) {
    if (n != n_) throw std::runtime_error("this " + name + " AIR was built for another trace length");
    if (prepared_.valid && prepared_.n == n && prepared_.ch == ch) {
        // built ahead (prepare_program): the powers of the composition coefficient are the only constants it could not know
        AirProgramData pd = prepared_.pd;
        Felt apow = felt_from_u64(1);
        for (uint32_t slot : prepared_.alpha_slot) {
            if (slot != 0xffffffffu) pd.program.consts[slot] = apow;
            apow = felt_mul(apow, alpha);
        }
        return pd;
    }
    Graph g;
    const int root = composition(g, ch, alpha);
    AirProgramData pd;
    pd.program = lower(g, root);
    pd.d_tables = tables_ ? tables_->u64() : nullptr;
    pd.table_desc = desc_;
    return pd;
}

// The graph names every value a proof brings by SYMBOL (Graph::runtime_constant), the k-th power of the composition coefficient as
// "alpha^" k and nothing else of it: the code and all other constants are decided by the challenges.  Lowered here with the
// coefficient 1; build_program writes the real powers over their slots (a power that no instruction reads has none).
void LayoutAir::prepare_program(uint64_t n, const std::vector<Felt> &ch) {
    prepared_.valid = false;
    if (n != n_) return;
    Graph g;
    const int root = composition(g, ch, felt_from_u64(1));
    prepared_.pd.program = lower(g, root);
    prepared_.pd.d_tables = tables_ ? tables_->u64() : nullptr;
    prepared_.pd.table_desc = desc_;
    std::map<uint32_t, uint32_t> slot_of;                     // graph constant -> program constant
    const std::vector<uint32_t> &gi = prepared_.pd.program.const_graph_index;
    for (uint32_t k = 0; k < gi.size(); ++k) slot_of[gi[k]] = k;
    prepared_.alpha_slot.clear();
    for (uint64_t k = 0;; ++k) {
        const int ix = g.symbol_const_index(Graph::sym("alpha^", k));
        if (ix < 0) break;
        auto it = slot_of.find((uint32_t)ix);
        prepared_.alpha_slot.push_back(it == slot_of.end() ? 0xffffffffu : it->second);
    }
    prepared_.n = n; prepared_.ch = ch;
    prepared_.valid = true;
}

Felt LayoutAir::composition_at(uint64_t n, const std::vector<Felt> &ch, const Felt &alpha, const Felt &z, const std::vector<Felt> &ood) {
    if (n != n_) throw std::runtime_error("this " + name + " AIR was built for another trace length");
    Graph g;
    const int root = composition(g, ch, alpha);
    std::map<std::pair<uint32_t, uint32_t>, Felt> cell;
    for (size_t j = 0; j < mask.size(); ++j) cell[mask[j]] = ood[j];
    return evaluate(g, root, z, [&](uint32_t c, uint32_t o) { return cell.at({c, o}); }, [&](uint32_t t) { return table_value_at(specs_.at(t), z); });
}

std::vector<uint64_t> LayoutAir::describe_tables() const {
    std::vector<uint64_t> out{specs_.size()};
    for (auto &s : specs_) {
        out.push_back((uint64_t)s.kind); out.push_back(s.e);
        out.push_back(s.num.size());
        for (auto &f : s.num) { out.push_back(f.p); out.push_back(f.e); }
        out.push_back(s.den.size());
        for (auto &f : s.den) { out.push_back(f.p); out.push_back(f.e); }
    }
    return out;
}

int LayoutAir::table_index(const TableSpec &s) {
    auto it = table_ix_.find(s);
    if (it != table_ix_.end()) return it->second;
    const int ix = (int)specs_.size();
    specs_.push_back(s);
    table_ix_[s] = ix;
    return ix;
}

E LayoutAir::multiplier(Graph &g, const Domain &d) {
    TableSpec per; per.kind = 2;
    for (auto &f : d.num) if (f.p > 1) per.num.push_back(f);
    for (auto &f : d.den) if (f.p > 1) per.den.push_back(f);
    bool have = false;
    E expr{&g, -1};
    if (!per.num.empty() || !per.den.empty()) { expr = E{&g, g.table((uint32_t)table_index(per))}; have = true; }
    for (auto &f : d.num) if (f.p == 1) {
        // g^e is a per-size value: symbol by the row it names, counted from whichever end is nearer (n-independent)
        const uint64_t row_sym = f.e > n_ / 2 ? Graph::sym("row from end", n_ - f.e) : Graph::sym("row", f.e);
        const E lin = E{&g, g.x()} - E{&g, g.runtime_constant(row_sym, felt_pow(g_, f.e))};
        expr = have ? expr * lin : lin; have = true;
    }
    for (auto &f : d.den) if (f.p == 1) {
        TableSpec inv; inv.kind = 3; inv.e = f.e;
        const E t{&g, g.table((uint32_t)table_index(inv))};
        expr = have ? expr * t : t; have = true;
    }
    return expr;
}

void LayoutAir::Composer::add(const std::string &domain_name, const Domain &d, const E &numerator) {
    const E term = numerator * E{&g_, g_.runtime_constant(Graph::sym("alpha^", count_++), apow_)};
    auto it = std::find_if(groups_.begin(), groups_.end(), [&](const Group &p) { return p.name == domain_name; });
    if (it == groups_.end()) groups_.push_back(Group{domain_name, d, term.id});
    else it->sum = g_.add(it->sum, term.id);
    apow_ = felt_mul(apow_, alpha_);
}

int LayoutAir::Composer::total() {
    int total = -1;
    for (auto &gr : groups_) {
        const E term = E{&g_, gr.sum} * air_.multiplier(g_, gr.d);
        total = total < 0 ? term.id : g_.add(total, term.id);
    }
    return total;
}

Felt LayoutAir::table_value_at(const TableSpec &s, const Felt &x) const {
    if (s.kind == 0) {
        auto it = column_coeffs_.find(s.e);
        if (it == column_coeffs_.end()) it = column_coeffs_.emplace(s.e, interpolate(column_values(s.e))).first;
        const Felt arg = felt_pow(x, n_ / column_period(s.e));
        Felt acc = felt_from_u64(0);
        for (size_t k = it->second.size(); k-- > 0;) acc = felt_add(felt_mul(acc, arg), it->second[k]);
        return acc;
    }
    if (s.kind == 3) return felt_inv(felt_sub(x, felt_pow(g_, s.e)));
    Felt num = felt_from_u64(1), den = felt_from_u64(1);
    for (auto &f : s.num) num = felt_mul(num, felt_sub(felt_pow(x, f.p), felt_pow(g_, f.e)));
    for (auto &f : s.den) den = felt_mul(den, felt_sub(felt_pow(x, f.p), felt_pow(g_, f.e)));
    return felt_mul(num, felt_inv(den));
}

void LayoutAir::build_tables() {
    const uint64_t N = n_ << lb_;
    std::vector<uint64_t> lengths;
    uint64_t off = 0;
    for (auto &s : specs_) {
        uint64_t len = 0;
        if (s.kind == 0) len = column_period(s.e) << lb_;
        else if (s.kind == 3) len = N;
        else { for (auto &f : s.num) len = std::max(len, N / f.p); for (auto &f : s.den) len = std::max(len, N / f.p); }
        uint32_t ll = 0;
        while ((1ull << ll) < len) ++ll;
        desc_.push_back((uint32_t)off); desc_.push_back(ll);
        lengths.push_back(len);
        off += len;
    }
    tables_.reset(new DeviceBuffer(ctx_, 32 * off));
    const Felt offset = felt_from_u64(offset_), w = root_of_unity(log_n_ + lb_), one = felt_from_u64(1);
    for (size_t t = 0; t < specs_.size(); ++t) {
        const TableSpec &s = specs_[t];
        uint64_t *dst = tables_->u64() + 4ull * desc_[2 * t];
        if (s.kind == 3) {
            const Felt c = felt_pow(g_, s.e);
            ok(ss_inverse_table(ctx_, log_n_ + lb_, offset.data(), c.data(), dst));
            continue;
        }
        std::vector<Felt> host(lengths[t]);
        if (s.kind == 0) {
            const std::vector<Felt> coeffs = interpolate(column_values(s.e));
            const uint64_t p = n_ / column_period(s.e);
            const Felt step = felt_pow(w, p);
            Felt x = felt_pow(offset, p);
            for (auto &v : host) {
                Felt acc = felt_from_u64(0);
                for (size_t k = coeffs.size(); k-- > 0;) acc = felt_add(felt_mul(acc, x), coeffs[k]);
                v = acc;
                x = felt_mul(x, step);
            }
        } else {
            // prod(X^p - c) / prod(X^p - c') along x_i = offset * w^i: every power advances by its own step; one batch inversion
            struct Run { Felt cur, step, c; };
            auto runs = [&](const std::vector<Factor> &fs) {
                std::vector<Run> r;
                for (auto &f : fs) r.push_back(Run{felt_pow(offset, f.p), felt_pow(w, f.p), felt_pow(g_, f.e)});
                return r;
            };
            std::vector<Run> num = runs(s.num), den = runs(s.den);
            std::vector<Felt> dens(host.size());
            for (size_t i = 0; i < host.size(); ++i) {
                Felt a = one, b = one;
                for (auto &r : num) { a = felt_mul(a, felt_sub(r.cur, r.c)); r.cur = felt_mul(r.cur, r.step); }
                for (auto &r : den) { b = felt_mul(b, felt_sub(r.cur, r.c)); r.cur = felt_mul(r.cur, r.step); }
                host[i] = a; dens[i] = b;
            }
            std::vector<Felt> prefix(host.size());
            Felt run = one;
            for (size_t i = 0; i < dens.size(); ++i) { prefix[i] = run; run = felt_mul(run, dens[i]); }
            Felt inv = felt_inv(run);
            for (size_t i = dens.size(); i-- > 0;) {
                host[i] = felt_mul(host[i], felt_mul(inv, prefix[i]));
                inv = felt_mul(inv, dens[i]);
            }
        }
        ok(ss_upload(ctx_, dst, host.data(), host.size() * 32));
    }
    ok(ss_ctx_sync(ctx_));
}

}  // namespace layout

std::vector<uint64_t> layout_air_tables(const Air &air) {
    const layout::LayoutAir *r = dynamic_cast<const layout::LayoutAir *>(&air);
    if (!r) throw std::runtime_error("not a layout AIR");
    return r->describe_tables();
}

}  // namespace ssh
