#include "public_input.hpp"

#include <cstring>
#include <stdexcept>

#include "../../include/sandstorm_hip.h"

namespace ssh {

namespace {
enum { SEG_PROGRAM, SEG_EXECUTION, SEG_OUTPUT, SEG_PEDERSEN, SEG_RANGE_CHECK, SEG_ECDSA, SEG_BITWISE, SEG_EC_OP, SEG_POSEIDON };
const char *SEG_NAME[9] = {"program", "execution", "output", "pedersen", "range_check", "ecdsa", "bitwise", "ec_op", "poseidon"};

U256 u(uint64_t v) { return U256{v, 0, 0, 0}; }
U256 name_code(const std::string &s) {          // Layout::sharp_code: the layout name as a big-endian integer
    U256 r{};
    for (size_t i = 0; i < s.size(); ++i) {
        const size_t bit = 8 * (s.size() - 1 - i);
        r[bit / 64] |= (uint64_t)(uint8_t)s[i] << (bit % 64);
    }
    return r;
}
std::array<uint8_t, 32> be_bytes(const U256 &v) {
    std::array<uint8_t, 32> o;
    for (int i = 0; i < 4; ++i) for (int b = 0; b < 8; ++b) o[i * 8 + b] = (uint8_t)(v[3 - i] >> (56 - 8 * b));
    return o;
}
void push_segment(std::vector<U256> &out, const AirPublicInput &pi, int k) {
    if (!pi.segments[k].present) throw std::runtime_error("the " + pi.layout + " layout needs the " + SEG_NAME[k] + " segment");
    out.push_back(u(pi.segments[k].begin_addr));
    out.push_back(u(pi.segments[k].stop_ptr));
}
Felt pedersen(const Felt &a, const Felt &b) {
    Felt o;
    if (ss_pedersen_hash_host(a.data(), b.data(), o.data()) != SS_OK) throw std::runtime_error(ss_last_error());
    return o;
}
}  // namespace

std::vector<U256> public_input_elements(const AirPublicInput &pi, int coin_kind) {
    if (pi.layout != "recursive" && pi.layout != "starknet") throw std::runtime_error("layout " + pi.layout + " is not implemented");
    if (!pi.n_steps) throw std::runtime_error("n_steps must be positive");
    std::vector<U256> v;
    // base values (src/input.rs:10-48)
    uint64_t log_steps = 0;
    while ((pi.n_steps >> (log_steps + 1)) != 0) ++log_steps;
    v.push_back(u(log_steps)); v.push_back(u(pi.rc_min)); v.push_back(u(pi.rc_max)); v.push_back(name_code(pi.layout));
    for (int k : {SEG_PROGRAM, SEG_EXECUTION, SEG_OUTPUT, SEG_PEDERSEN, SEG_RANGE_CHECK}) push_segment(v, pi, k);
    // layout specific values (src/input.rs:50-116)
    if (pi.layout == "starknet") { for (int k : {SEG_ECDSA, SEG_BITWISE, SEG_EC_OP, SEG_POSEIDON}) push_segment(v, pi, k); }
    else push_segment(v, pi, SEG_BITWISE);
    const MemoryEntry *padding = nullptr;       // public_memory_padding(): the entry at address 1
    for (auto &e : pi.public_memory) if (e.address == 1) { padding = &e; break; }
    if (!padding) throw std::runtime_error("public memory has no entry at address 1");
    v.push_back(u(padding->address)); v.push_back(padding->value); v.push_back(u(1));
    // main memory page: size, hash of (address, value) pairs (src/input.rs:118-139)
    v.push_back(u(pi.public_memory.size()));
    if (coin_kind == SS_COIN_SOLIDITY) {        // CanonicalKeccak256HashFn::hash_elements
        std::vector<uint8_t> msg;
        for (auto &e : pi.public_memory) {
            for (const U256 &x : {u(e.address), e.value}) { auto b = be_bytes(x); msg.insert(msg.end(), b.begin(), b.end()); }
        }
        const Digest d = keccak256(msg.data(), msg.size());
        U256 h{};
        for (int i = 0; i < 4; ++i) for (int b = 0; b < 8; ++b) h[3 - i] |= (uint64_t)d[i * 8 + b] << (56 - 8 * b);
        v.push_back(h);
    } else {                                    // PedersenHashFn::hash_elements, digest = canonical big-endian bytes
        Felt cur = felt_from_u64(0);
        uint64_t n = 0;
        for (auto &e : pi.public_memory) {
            cur = pedersen(cur, felt_from_u64(e.address));
            cur = pedersen(cur, felt_from_canonical(e.value));
            n += 2;
        }
        const auto d = canonical_be_bytes(pedersen(cur, felt_from_u64(n)));
        U256 h{};
        for (int i = 0; i < 4; ++i) for (int b = 0; b < 8; ++b) h[3 - i] |= (uint64_t)d[i * 8 + b] << (56 - 8 * b);
        v.push_back(h);
    }
    return v;
}

Digest public_coin_seed(const AirPublicInput &pi, int coin_kind) {
    std::vector<uint8_t> seed;
    for (const U256 &x : public_input_elements(pi, coin_kind)) { auto b = be_bytes(x); seed.insert(seed.end(), b.begin(), b.end()); }
    return coin_kind == SS_COIN_SOLIDITY ? keccak256(seed.data(), seed.size()) : blake2s256(seed.data(), seed.size());
}

}  // namespace ssh
