// trace_recursive.hpp — base-trace generation of the `recursive` layout on the host ("next"
// row X1): ExecutionTrace::new (layouts/src/recursive/trace.rs:95-660) with the builtins' instance traces
// (builtins/src/{pedersen,bitwise,range_check}/mod.rs) and the pools of layouts/src/utils.rs.
// Mirror: sandstorm_amd/layouts/recursive.py::base_trace, against which tests/test_layout_recursive.py checks it cell
// for cell; that module also holds the 93 constraints the trace is validated with.
#pragma once
#include <functional>
#include <cstdint>
#include <vector>

#include <cstring>

#include "public_input.hpp"
#include "../../include/sandstorm_hip.h"

namespace ssh {

struct RegisterState { uint64_t ap, fp, pc; };                 // binary/src/lib.rs:50-56

std::vector<RegisterState> read_register_states(const uint8_t *data, size_t len);        // trace.bin
// trace.bin read where it lies: its records ARE (ap, fp, pc) as little-endian 8-byte integers, and the generators read every record
// once or twice - a copy of the file into a vector (24 MB at 2^20 steps, fresh pages every call) cost 10-17 ms per proof before the
// first column could be made.  Also what a vector of states converts to; the bytes must outlive the view.
struct RegisterStates {
    const uint8_t *data = nullptr;
    size_t count = 0;
    RegisterStates() = default;
    RegisterStates(const uint8_t *bytes, size_t len);                                     // refuses a length that is not whole records
    RegisterStates(const std::vector<RegisterState> &v) : data(reinterpret_cast<const uint8_t *>(v.data())), count(v.size()) {}
    size_t size() const { return count; }
    RegisterState operator[](size_t i) const { RegisterState s; memcpy(&s, data + sizeof(RegisterState) * i, sizeof s); return s; }
};
static_assert(sizeof(RegisterState) == 24, "a trace.bin record");
// memory.bin -> memory[address] (canonical 256-bit words); present[address] = 0 for cells the run never touched
void read_memory(const uint8_t *data, size_t len, std::vector<U256> &memory, std::vector<uint8_t> &present);

struct PedersenInstance { uint32_t index; U256 a, b; };
struct RangeCheckInstance { uint32_t index; U256 value; };
struct BitwiseInstance { uint32_t index; U256 x, y; };
struct PrivateInput {                                           // air-private-input.json
    std::vector<PedersenInstance> pedersen;
    std::vector<RangeCheckInstance> range_check;
    std::vector<BitwiseInstance> bitwise;
};

// -> the 7 base columns (Montgomery felts, 16 rows per cycle): flags | diluted unordered / bitwise | diluted ordered |
// memory pool | sorted memory | range check / Pedersen partial sums | auxiliary / Pedersen suffixes, slopes
std::vector<std::vector<Felt>> recursive_base_trace(const RegisterStates &states, const std::vector<U256> &memory,
                                                    const std::vector<uint8_t> &present, const AirPublicInput &pi,
                                                    const PrivateInput &priv);

// the same into the caller's 7 columns of 16 * cycles felts each (every cell is written)
// column_done (optional): called with c as soon as no section will write column c again (the diluted pair after bitwise - the first
// section: it needs no CPU cell -, flags after the CPU cells, auxiliary after Pedersen, range check after its builtin, the memory pool
// after the gap fillers, sorted memory last)
void recursive_base_trace_into(Felt *const out[7], const RegisterStates &states, const std::vector<U256> &memory,
                               const std::vector<uint8_t> &present, const AirPublicInput &pi, const PrivateInput &priv,
                               const std::function<void(int)> *column_done = nullptr);

// the same 7 columns made in HBM by the device (csrc/trace.hip through the ss_trace_* entry points; host/device_trace.hpp): the raw
// files' bytes go up as they are, the host only counts the range-check pool and traces the DISTINCT builtin instances.
// memory / present: memory.bin as read_memory gives it (the plans read it).
void recursive_base_trace_device(ss_ctx *ctx, uint64_t *const d_cols[7], const uint8_t *trace_bin, uint64_t trace_len, const uint8_t *memory_bin,
                                 uint64_t memory_len, const std::vector<U256> &memory, const std::vector<uint8_t> &present, const AirPublicInput &pi,
                                 const PrivateInput &priv);

}  // namespace ssh
