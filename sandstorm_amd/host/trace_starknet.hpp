// trace_starknet.hpp — base-trace generation of the `starknet` layout on the host: ExecutionTrace::new
// (layouts/src/starknet/trace.rs:98-987) with the instance traces of its six builtins
// (builtins/src/{pedersen,range_check,ecdsa,bitwise,ec_op,poseidon}/mod.rs) and the pools of layouts/src/utils.rs.
// Mirror: sandstorm_amd/layouts/starknet.py::base_trace, which is pinned to the reference's own proof of its bootloader
// run; tests/test_layout_starknet.py compares the two cell for cell on that run.
#pragma once
#include <functional>
#include "trace_recursive.hpp"
#include "../../include/sandstorm_hip.h"

namespace ssh {

struct EcdsaInstance { uint32_t index; U256 pubkey_x, message, r, w; };
struct EcOpInstance { uint32_t index; U256 p_x, p_y, q_x, q_y, m; };
struct PoseidonInstance { uint32_t index; U256 input[3]; };
struct StarknetPrivateInput : PrivateInput {
    std::vector<EcdsaInstance> ecdsa;
    std::vector<EcOpInstance> ec_op;
    std::vector<PoseidonInstance> poseidon;
};

// -> the 9 base columns (Montgomery felts): flags | Pedersen partial sum x | y | suffix | slope | memory pool | sorted memory |
// range check / diluted check / bitwise / Poseidon partial rounds | auxiliary / ECDSA / EC op / Poseidon full rounds
std::vector<std::vector<Felt>> starknet_base_trace(const RegisterStates &states, const std::vector<U256> &memory,
                                                   const std::vector<uint8_t> &present, const AirPublicInput &pi,
                                                   const StarknetPrivateInput &priv);

// the same into caller-owned columns of 16 * states.size() felts each (every cell is written)
// column_done (optional): called with c as soon as no section will write column c again - flags after the CPU cells, the four
// Pedersen columns after their builtin, range check and auxiliary after Poseidon, the memory pool after the gap fillers, the
// sorted memory last - so that an upload of c can leave while the rest is still being generated (host/prover.hpp ColumnFeed)
void starknet_base_trace_into(Felt *const out[9], const RegisterStates &states, const std::vector<U256> &memory,
                              const std::vector<uint8_t> &present, const AirPublicInput &pi, const StarknetPrivateInput &priv,
                              const std::function<void(int)> *column_done = nullptr);

// the same 9 columns made in HBM by the device (csrc/trace.hip through the ss_trace_* entry points; host/device_trace.hpp): the raw
// files' bytes go up as they are - 25 MB where the host-made columns are 4.8 GB at 2^20 steps -, the host only counts the range-check
// pool and traces the DISTINCT builtin instances.  memory / present: memory.bin as read_memory gives it (the plans read it).
void starknet_base_trace_device(ss_ctx *ctx, uint64_t *const d_cols[9], const uint8_t *trace_bin, uint64_t trace_len, const uint8_t *memory_bin,
                                uint64_t memory_len, const std::vector<U256> &memory, const std::vector<uint8_t> &present, const AirPublicInput &pi,
                                const StarknetPrivateInput &priv);

// shared with the AIR (air_starknet.cpp): StarkWare's Hades round constants, the curve's generator and beta
const std::vector<std::array<Felt, 3>> &poseidon_round_keys();
void starknet_curve(Felt &generator_x, Felt &generator_y, Felt &beta);

}  // namespace ssh
