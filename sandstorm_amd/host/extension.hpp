// extension.hpp — Trace::build_extension_columns above the C ABI ("next" row X1).
// The reference fills the extension trace with sequential host loops between two device phases
// (layouts/src/recursive/trace.rs:699-814, layouts/src/starknet/trace.rs:997-1100); here the auxiliary columns
// stay in HBM and the loops are ss_permutation_product / ss_diluted_aggregate calls.
#pragma once
#include <functional>
#include <string>
#include <vector>

#include "prover.hpp"

namespace ssh {

struct TraceColumns {                       // what the reference's trace object keeps next to the base matrix (device pointers)
    const uint64_t *npc = nullptr;          // npc_column: program-order accesses [a, v, a, v, ...]
    const uint64_t *memory = nullptr;       // memory_column: address-order accesses [a', v', ...]
    const uint64_t *range_check = nullptr;  // range_check_column
    const uint64_t *diluted_unordered = nullptr, *diluted_ordered = nullptr;   // recursive layout only
    uint64_t trace_len = 0;
};

// -> recursive: [diluted_check_aggregate, diluted_check_permutation, mem_and_rc_permutation]; starknet: [permutation_column].
// check: throw where the reference asserts that a permutation product closes to one (trace.rs:734, 757-760).
Matrix build_extension_columns(ss_ctx *ctx, const std::string &layout, const TraceColumns &cols,
                               const std::vector<Felt> &challenges, bool check = true);

// The same columns as ROW BLOCKS over `world` devices (one process per GPU): `cols` points at this rank's rows
// [rank n / world, (rank + 1) n / world) of the auxiliary columns (cols.trace_len = n, the whole trace's length), the result holds
// the same rows of the extension columns.  Every rank calls it with the same challenges; all_gather is entered ONCE per call with
// (number of products + 4) * 32 bytes: every rank's bytes in rank order (a Transport's all_gather, MPI_Allgather ...).
struct BlockGather {
    uint32_t rank = 0, world = 1;
    std::function<std::vector<uint8_t>(const std::vector<uint8_t> &)> all_gather;
};
Matrix build_extension_blocks(ss_ctx *ctx, const std::string &layout, const TraceColumns &cols, const std::vector<Felt> &challenges,
                              const BlockGather &gather, bool check = true);

}  // namespace ssh
