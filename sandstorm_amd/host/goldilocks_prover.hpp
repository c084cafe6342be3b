// goldilocks_prover.hpp — the 64-bit field's claim in the C++ host (goldilocks_prover.cpp; mirror of sandstorm_amd/goldilocks.py).
#pragma once
#include <array>
#include <cstdint>
#include <functional>
#include <memory>
#include <utility>
#include <vector>

#include "coin.hpp"
#include "../../include/sandstorm_hip.h"

namespace ssh {
namespace gl {

using Fq3 = std::array<uint64_t, 3>;                        // c0 + c1 X + c2 X^2, X^3 = 2, coordinates < p = 2^64 - 2^32 + 1
Fq3 mul3(const Fq3 &a, const Fq3 &b);
Digest sha256(const uint8_t *msg, size_t len);              // FIPS 180-4 (the coin's host side when Options.sha256)

struct Options {                                            // goldilocks.py Options: the CLI's defaults (cli/src/main.rs:51-60)
    uint32_t num_queries = 65, log_blowup = 1, grinding = 16, fold = 8, max_remainder = 16;
    bool sha256 = false;                                    // the parts cli/src/main.rs:119-120 names: SHA-256 trees and a SHA-256 coin
};
struct Opening {
    uint32_t width = 0, depth = 0;
    std::vector<uint64_t> rows;                             // [positions][width]
    std::vector<uint8_t> paths;                             // [positions][depth][32]
};
struct FriLayer {
    Digest root{};
    uint32_t log_len = 0;
    Opening opening;
};
struct Proof {
    uint64_t trace_len = 0, pow_nonce = 0;
    Digest base_root{}, ext_root{}, comp_root{};
    bool has_ext = false;
    std::vector<uint64_t> ood_trace, ood_comp, remainder;   // [cells][3], [6][3], [len][3]
    std::vector<FriLayer> fri_layers;
    Opening base, ext, comp;
};
// what the layout decides for a proof, from the caller: the extension trace's coordinate columns (device, n values each) for the
// drawn challenges, and the lowered composition program (ss_eval_quotient_gl64x3's format) for them and the composition coefficient
using ExtensionBuilder = std::function<std::vector<const uint64_t *>(const std::vector<Fq3> &challenges)>;
struct ProgramData {
    std::vector<uint32_t> code;                             // two words per instruction
    std::vector<uint64_t> consts;                           // three values per constant
    uint32_t n_slots = 0;
    const uint64_t *d_tables = nullptr;
    std::vector<uint32_t> table_desc;
};
using ProgramBuilder = std::function<ProgramData(const std::vector<Fq3> &challenges, const Fq3 &alpha)>;

Digest transcript_seed(const Digest &seed, const Options &opt, uint64_t trace_len, const Digest &statement_digest);
Proof prove(ss_ctx *ctx, const Options &opt, const Digest &seed, const Digest &statement_digest, const std::vector<const uint64_t *> &base_cols,
            uint64_t n, const std::vector<std::pair<uint32_t, uint32_t>> &mask, uint32_t num_challenges, uint32_t num_ext,
            const ExtensionBuilder &build_extension, const ProgramBuilder &build_program);

}  // namespace gl
}  // namespace ssh
