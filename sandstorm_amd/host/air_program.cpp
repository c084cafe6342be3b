#include "air_program.hpp"

#include <algorithm>
#include <functional>
#include <stdexcept>

#include "../../include/sandstorm_hip.h"

namespace ssh {

int Graph::intern(NodeKind k, int a, int b, uint32_t p0, uint32_t p1) {
    auto key = std::make_tuple((int)k, a, b, p0, p1);
    auto it = pool_.find(key);
    if (it != pool_.end()) return it->second;
    Node n; n.kind = k; n.a = a; n.b = b; n.p0 = p0; n.p1 = p1;
    nodes_.push_back(n);
    return pool_[key] = (int)nodes_.size() - 1;
}
int Graph::x() { return intern(NodeKind::X, -1, -1, 0, 0); }
int Graph::constant(const Felt &m) {
    auto it = const_ix_.find(m);
    int ix;
    if (it == const_ix_.end()) { ix = (int)consts_.size(); consts_.push_back(m); const_ix_[m] = ix; } else ix = it->second;
    return intern(NodeKind::Const, -1, -1, (uint32_t)ix, 0);
}
int Graph::runtime_constant(uint64_t symbol, const Felt &m) {
    auto it = const_sym_.find(symbol);
    int ix;
    if (it == const_sym_.end()) { ix = (int)consts_.size(); consts_.push_back(m); const_sym_[symbol] = ix; }
    else { ix = it->second; if (!(consts_[ix] == m)) throw std::runtime_error("runtime constant: one symbol, two values"); }
    return intern(NodeKind::Const, -1, -1, (uint32_t)ix, 1);
}
int Graph::trace(uint32_t col, uint32_t off) { return intern(NodeKind::Trace, -1, -1, col, off); }
int Graph::table(uint32_t index) { return intern(NodeKind::Table, -1, -1, index, 0); }
int Graph::add(int a, int b) { if (a > b) std::swap(a, b); return intern(NodeKind::Add, a, b, 0, 0); }
int Graph::mul(int a, int b) { if (a > b) std::swap(a, b); return intern(NodeKind::Mul, a, b, 0, 0); }
int Graph::sub(int a, int b) { return intern(NodeKind::Sub, a, b, 0, 0); }
int Graph::inv(int a) { return intern(NodeKind::Inv, a, -1, 0, 0); }

namespace {
struct Lowerer {
    const Graph &g;
    Program prog;
    std::vector<int> uses;
    std::map<int, uint32_t> slot_of;
    std::vector<uint32_t> free_slots;
    std::map<uint32_t, uint32_t> const_ix;          // graph constant index -> program constant index (no merging by value)

    explicit Lowerer(const Graph &gr) : g(gr), uses(gr.nodes().size(), 0) {}

    bool leaf(int n) const { auto k = g.nodes()[n].kind; return k == NodeKind::X || k == NodeKind::Const || k == NodeKind::Trace || k == NodeKind::Table; }
    uint32_t alloc_slot() {
        if (!free_slots.empty()) { uint32_t s = free_slots.back(); free_slots.pop_back(); return s; }
        return prog.n_slots++;
    }
    void emit(uint32_t op, uint32_t dst, uint32_t kind = 0, uint32_t payload = 0) {
        prog.code.push_back(op | (dst << 8) | (kind << 12));
        prog.code.push_back(payload);
    }
    // (kind, payload) when usable in place
    bool operand(int n, uint32_t &kind, uint32_t &payload) {
        const Node &nd = g.nodes()[n];
        switch (nd.kind) {
        case NodeKind::X: kind = SS_SRC_X; payload = 0; return true;
        case NodeKind::Const: {
            auto it = const_ix.find(nd.p0);
            if (it == const_ix.end()) { payload = (uint32_t)prog.consts.size(); prog.consts.push_back(g.constants()[nd.p0]); prog.const_graph_index.push_back(nd.p0); const_ix[nd.p0] = payload; } else payload = it->second;
            kind = SS_SRC_CONST; return true; }
        case NodeKind::Trace: kind = SS_SRC_TRACE; payload = (nd.p0 << 24) | nd.p1; return true;
        case NodeKind::Table: kind = SS_SRC_TABLE; payload = nd.p0; return true;
        default: break;
        }
        auto it = slot_of.find(n);
        if (it == slot_of.end()) return false;
        kind = SS_SRC_SLOT; payload = it->second;
        return true;
    }
    void consume(int n) {
        if (leaf(n)) return;
        if (--uses[n] == 0) { auto it = slot_of.find(n); if (it != slot_of.end()) { free_slots.push_back(it->second); slot_of.erase(it); } }
    }
    void emit_op(uint32_t opc, uint32_t dst, int n) { uint32_t k, p; operand(n, k, p); emit(opc, dst, k, p); }

    void gen(int n, uint32_t dst) {
        uint32_t k, p;
        if (operand(n, k, p)) { emit(SS_OP_MOV, dst, k, p); consume(n); return; }
        const Node &nd = g.nodes()[n];
        if (nd.kind == NodeKind::Inv) {
            gen(nd.a, dst);
            emit(SS_OP_INV, dst);
        } else {
            const int l = nd.a, r = nd.b;
            const uint32_t opc = nd.kind == NodeKind::Add ? SS_OP_ADD : nd.kind == NodeKind::Sub ? SS_OP_SUB : SS_OP_MUL;
            const uint32_t ropc = nd.kind == NodeKind::Sub ? (uint32_t)SS_OP_RSUB : opc;
            if (l == r && !operand(l, k, p)) {
                --uses[l];                                   // both references served by one evaluation
                gen(l, dst);
                emit(opc, dst, SS_SRC_ACC, dst);
            } else if (operand(r, k, p)) {
                gen(l, dst);
                emit_op(opc, dst, r);
                consume(r);
            } else if (operand(l, k, p)) {
                gen(r, dst);
                emit_op(ropc, dst, l);
                consume(l);
            } else {
                gen(l, dst);
                if (operand(r, k, p)) { emit(opc, dst, k, p); consume(r); }
                else if (dst + 1 < 4) { gen(r, dst + 1); emit(opc, dst, SS_SRC_ACC, dst + 1); }
                else {
                    const uint32_t s = alloc_slot();
                    emit(SS_OP_ST, dst, 0, s);
                    gen(r, dst);
                    emit(ropc, dst, SS_SRC_SLOT, s);
                    free_slots.push_back(s);
                }
            }
        }
        if (--uses[n] > 0) {                                 // more parents: park it
            const uint32_t s = alloc_slot();
            slot_of[n] = s;
            emit(SS_OP_ST, dst, 0, s);
        }
    }
};
}  // namespace

Program lower(const Graph &g, int root) {
    Lowerer L(g);
    // reference counts over the nodes reachable from root
    std::vector<char> seen(g.nodes().size(), 0);
    std::vector<int> stack{root};
    while (!stack.empty()) {
        int n = stack.back(); stack.pop_back();
        if (seen[n]) continue;
        seen[n] = 1;
        const Node &nd = g.nodes()[n];
        for (int c : {nd.a, nd.b}) if (c >= 0) { ++L.uses[c]; stack.push_back(c); }
    }
    ++L.uses[root];
    L.gen(root, 0);
    L.emit(SS_OP_OUT, 0);
    return L.prog;
}

Felt evaluate(const Graph &g, int root, const Felt &x, const std::function<Felt(uint32_t, uint32_t)> &trace_at,
              const std::function<Felt(uint32_t)> &table_at) {
    const auto &nodes = g.nodes();
    std::vector<Felt> val(nodes.size());
    std::vector<char> done(nodes.size(), 0);
    std::vector<int> stack{root};
    while (!stack.empty()) {                            // iterative post-order: children first
        const int id = stack.back();
        if (done[id]) { stack.pop_back(); continue; }
        const Node &nd = nodes[id];
        const bool need_a = nd.a >= 0 && !done[nd.a], need_b = nd.b >= 0 && !done[nd.b];
        if (need_a) stack.push_back(nd.a);
        if (need_b) stack.push_back(nd.b);
        if (need_a || need_b) continue;
        switch (nd.kind) {
        case NodeKind::X: val[id] = x; break;
        case NodeKind::Const: val[id] = g.constants()[nd.p0]; break;
        case NodeKind::Trace: val[id] = trace_at(nd.p0, nd.p1); break;
        case NodeKind::Table: val[id] = table_at(nd.p0); break;
        case NodeKind::Add: val[id] = felt_add(val[nd.a], val[nd.b]); break;
        case NodeKind::Sub: val[id] = felt_sub(val[nd.a], val[nd.b]); break;
        case NodeKind::Mul: val[id] = felt_mul(val[nd.a], val[nd.b]); break;
        case NodeKind::Inv: val[id] = felt_inv(val[nd.a]); break;
        }
        done[id] = 1;
        stack.pop_back();
    }
    return val[root];
}

}  // namespace ssh
