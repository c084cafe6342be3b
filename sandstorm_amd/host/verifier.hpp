// verifier.hpp — host-side verifier of proofs in the reference's wire format, C++ mirror of
// sandstorm_amd/verifier.py / wire.py: transcript replay, out-of-domain identity, Merkle openings, DEEP values, FRI
// chain, remainder, proof of work — under the conventions the reference's shipped proofs pin (prover.hpp Conventions).
// Keccak trees as the reference's files pin them; FriendlyMerkleTree as its source has it (no sample file: source-pinned).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "prover.hpp"

namespace ssh {

struct WireOpening {
    int variant = 0;                        // 0: hashed leaves (digest sibling / leaf), 1: single column (felt sibling / leaf)
    std::vector<Digest> path;               // above the leaf pair, bottom-up (a Pedersen node: big-endian canonical bytes)
    std::vector<uint8_t> path_tags;         // FriendlyMerkleTree: MixedMerkleDigest tag per path entry (0 Pedersen, 1 Blake2s)
    Digest sibling_digest{}, leaf_digest{};
    Felt sibling_felt{}, leaf_felt{};       // Montgomery
};
struct WireFriLayer { std::vector<Felt> rows; std::vector<WireOpening> openings; Digest root{}; uint8_t root_tag = 0; };
struct WireProof {
    uint32_t options[5] = {0, 0, 0, 0, 0};
    uint64_t trace_len = 0, pow_nonce = 0;
    Digest base_root{}, extension_root{}, composition_root{};
    bool has_extension = false;
    int tree_kind = SS_TREE_KECCAK_M20;     // what the bytes were parsed as
    uint8_t root_tags[3] = {0, 0, 0};       // FriendlyMerkleTree: MixedMerkleDigest tags of the three trace roots
    std::vector<WireFriLayer> fri_layers;
    std::vector<Felt> remainder, base_rows, extension_rows, composition_rows, ood_trace, ood_composition;
    std::vector<WireOpening> base_openings, extension_openings, composition_openings;
};

// throws std::runtime_error("malformed proof: ...") on any structural problem, trailing bytes included.  tree_kind: the
// claim's tree - the bytes do not say which digest encoding they use (SS_TREE_FRIENDLY: MixedMerkleDigest nodes and
// FriendlyMerkleTreeProof openings, crypto/src/merkle/mixed.rs:46-101, mod.rs:168-236)
WireProof parse_wire(const uint8_t *data, size_t len, int tree_kind = SS_TREE_KECCAK_M20);

// `Proof::security_level_bits` (cli/src/main.rs:203; conjectured): num_queries * log2(blowup) + grinding bits, capped by the
// field (252 - log2 of the LDE domain) and by the collision resistance of the claim's hashes (crypto/src/hash/keccak.rs:17,64,
// blake2s.rs:14,67, pedersen.rs:48; merkle/mod.rs:100-102,283-285,434-436)
uint32_t conjectured_security_bits(const uint32_t options[5], uint64_t trace_len, int tree_kind);

// throws std::runtime_error naming the failed check; returns the query positions.  The proof's own options are untrusted
// (`claim.verify(proof, required_security_bits)`, cli/src/main.rs:176, default 80): a proof whose options conjecture less is
// rejected, and so is one whose options differ from `expected_options` when given.
std::vector<uint64_t> verify(const WireProof &proof, Air &air, int tree_kind, int coin_kind, const Digest &coin_seed,
                             const Conventions &conv = Conventions(), uint32_t required_security_bits = 80,
                             const ProofOptions *expected_options = nullptr, uint32_t n_friendly_layers = 22);

}  // namespace ssh
