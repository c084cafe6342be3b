// public_input.hpp — AirPublicInput -> public-coin seed on the host:
// CairoAuxInput::public_input_elements (src/input.rs:10-150) and CairoPublicCoin::from_public_input
// (src/lib.rs:145-167).  Mirror: sandstorm_amd/public_input.py.
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "coin.hpp"

namespace ssh {

using U256 = std::array<uint64_t, 4>;           // plain 256-bit integer, little-endian limbs

struct Segment { bool present = false; uint32_t begin_addr = 0, stop_ptr = 0; };
struct MemoryEntry { uint32_t address = 0; U256 value{}; };       // value: canonical integer < p

struct AirPublicInput {                         // binary/src/lib.rs:296-318
    std::string layout;                         // "recursive" | "starknet"
    uint16_t rc_min = 0, rc_max = 0;
    uint64_t n_steps = 0;
    // order: program, execution, output, pedersen, range_check, ecdsa, bitwise, ec_op, poseidon
    Segment segments[9];
    std::vector<MemoryEntry> public_memory;
};

std::vector<U256> public_input_elements(const AirPublicInput &pi, int coin_kind);      // src/input.rs:141-149
Digest public_coin_seed(const AirPublicInput &pi, int coin_kind);                       // src/lib.rs:145-167

}  // namespace ssh
