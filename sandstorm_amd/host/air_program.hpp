// air_program.hpp — Expr DAG and its lowering to the constraint-evaluation program of the
// C ABI (ss_air_program).  C++ counterpart of the reference's `Expr` tree
// (layouts/src/recursive/air.rs:61-1200, ministark::expression::Expr) and of what a Rust
// `hip` feature would do once per (layout, trace length): walk
// `composition_constraint(..).reuse_shared_nodes()` and emit code for the 4-accumulator machine.
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <tuple>
#include <vector>

#include "coin.hpp"

namespace ssh {

enum class NodeKind { X, Const, Trace, Table, Add, Sub, Mul, Inv };

struct Node {
    NodeKind kind;
    int a = -1, b = -1;          // children (node ids)
    uint32_t p0 = 0, p1 = 0;     // Trace: col, row offset; Table: index; Const: index into the graph's constants
};

// Hash-consed expression graph: structurally equal nodes are the same node, so shared
// sub-expressions are shared by construction.
class Graph {
public:
    int x();
    int constant(const Felt &mont);                                   // a STRUCTURAL constant of the AIR: interned by value
    int constant_u64(uint64_t v) { return constant(felt_from_u64(v)); }
    // A constant whose value is only known per proof (a challenge, a hint from the public input, a power of the
    // composition coefficient, a power of the trace generator): interned by SYMBOL, never by value, so that the shape of
    // the graph - and with it the lowered program - is the same for every statement of a layout even when two such
    // values (or one of them and a structural constant) happen to coincide.  The compiled constraint kernels
    // (csrc/quotient_gen_*.hip, generated from the lowered program) rely on that.
    int runtime_constant(uint64_t symbol, const Felt &mont);
    static constexpr uint64_t sym(const char *name, uint64_t index = 0) {
        uint64_t h = 0xcbf29ce484222325ull;
        for (const char *c = name; *c; ++c) h = (h ^ (uint64_t)(unsigned char)*c) * 0x100000001b3ull;
        return (h ^ index) * 0x100000001b3ull;
    }
    int trace(uint32_t col, uint32_t row_offset);
    int table(uint32_t index);
    int add(int a, int b);
    int sub(int a, int b);
    int mul(int a, int b);
    int inv(int a);
    // the index in constants() of a runtime constant's value (-1: the graph never named that symbol)
    int symbol_const_index(uint64_t symbol) const { auto it = const_sym_.find(symbol); return it == const_sym_.end() ? -1 : it->second; }
    const std::vector<Node> &nodes() const { return nodes_; }
    const std::vector<Felt> &constants() const { return consts_; }
private:
    int intern(NodeKind k, int a, int b, uint32_t p0, uint32_t p1);
    std::vector<Node> nodes_;
    std::vector<Felt> consts_;
    std::map<std::tuple<int, int, int, uint32_t, uint32_t>, int> pool_;
    std::map<Felt, int> const_ix_;
    std::map<uint64_t, int> const_sym_;
};

struct Program {
    std::vector<uint32_t> code;      // 2 words per instruction
    std::vector<Felt> consts;
    std::vector<uint32_t> const_graph_index;     // consts[k] is the graph's constants()[const_graph_index[k]] (a program built ahead of its
                                                 // composition coefficient has the powers of it patched in: LayoutAir::prepare_program)
    uint32_t n_slots = 0;
    uint32_t n_instr() const { return (uint32_t)(code.size() / 2); }
};

// code leaving `root` in accumulator 0 followed by OUT
Program lower(const Graph &g, int root);

// Direct evaluation of the DAG at one point (the definition the lowered program must reproduce; the verifier's side of
// the out-of-domain identity).  trace_at(col, row offset) and table_at(index) supply the leaves.
Felt evaluate(const Graph &g, int root, const Felt &x, const std::function<Felt(uint32_t, uint32_t)> &trace_at,
              const std::function<Felt(uint32_t)> &table_at);

}  // namespace ssh
