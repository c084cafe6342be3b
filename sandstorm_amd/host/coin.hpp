// coin.hpp — host-side Fiat-Shamir coins (C++), byte-compatible with the reference's
// SolidityVerifierPublicCoin (crypto/src/public_coin/solidity.rs:36-161) and
// CairoVerifierPublicCoin (crypto/src/public_coin/cairo.rs:42-174).  Host code in the
// reference too; uses the C ABI's host hashes (ss_keccak256_host, ss_pedersen_hash_host).
#pragma once
#include <array>
#include <cstddef>
#include <cstdint>
#include <set>
#include <vector>

namespace ssh {

using Felt = std::array<uint64_t, 4>;      // Montgomery limbs, little-endian
using Digest = std::array<uint8_t, 32>;

Digest keccak256(const uint8_t *msg, size_t len);
Digest blake2s256(const uint8_t *msg, size_t len);
// PublicCoin::verify_proof_of_work for a coin kind: enough leading zero bits in H(H(magic | digest | bits) | nonce)
bool verify_proof_of_work(int coin_kind, const Digest &digest, uint32_t bits, uint64_t nonce);
std::array<uint8_t, 32> mont_be_bytes(const Felt &f);     // to_montgomery(e).to_be_bytes::<32>()
Felt felt_from_u64(uint64_t v);
Felt felt_from_canonical(const Felt &value);              // little-endian limbs of an integer < p -> Montgomery
Felt felt_mul(const Felt &a, const Felt &b);
Felt felt_pow(const Felt &a, uint64_t e);
Felt felt_add(const Felt &a, const Felt &b);
Felt felt_sub(const Felt &a, const Felt &b);
Felt felt_neg(const Felt &a);
Felt felt_inv(const Felt &a);                            // a^(p-2); 0 -> 0
Felt root_of_unity(uint32_t log_n);                      // 3^((p-1)/2^log_n)
std::array<uint8_t, 32> canonical_be_bytes(const Felt &f);

class PublicCoin {
public:
    PublicCoin(int kind, const Digest &seed) : kind_(kind), digest_(seed), counter_(0) {}
    void reseed_with_bytes(const uint8_t *bytes, size_t len);
    void reseed_with_digest(const Digest &d) { reseed_with_bytes(d.data(), 32); }
    void reseed_with_field_elements(const std::vector<Felt> &v);        // solidity.rs:66-71 / cairo.rs:76-80
    void reseed_with_field_element_vector(const std::vector<Felt> &v);
    void reseed_with_int(uint64_t v);
    Felt draw();
    std::vector<uint64_t> draw_queries(size_t max_n, uint64_t domain_size);   // sorted, de-duplicated (BTreeSet)
    const Digest &digest() const { return digest_; }
    uint64_t counter() const { return counter_; }
    int kind() const { return kind_; }
private:
    Digest hash(const uint8_t *m, size_t n) const;
    Digest draw_bytes();
    int kind_;
    Digest digest_;
    uint64_t counter_;
};

}  // namespace ssh
