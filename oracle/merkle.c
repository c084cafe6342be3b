/* ORACLE — TEST INFRASTRUCTURE ONLY (see fp252.h).
 *
 * Merkle commitment rows H2/H3/H4 of SURVEY.md §8(a).  The tree *builder*
 * (ministark MerkleTreeImpl::new) is un-vendored; the tree *configuration* is
 * the reference's:
 *   - LeafVariantMerkleTree: crypto/src/merkle/mod.rs:240-304, UnhashedLeaf
 *     config mod.rs:419-437 (hash_leaves = H::hash_elements([l0,l1]),
 *     hash_nodes = H::merge)
 *   - FriendlyMerkleTree: mod.rs:43-123 with MixedHashMerkleTreeConfigImpl
 *     (mixed.rs:106-125) and hash_boundary (mixed.rs:148-155): output nodes at
 *     depth < N_FRIENDLY_LAYERS are Pedersen hashes, Blake2s digests entering
 *     the boundary are read as big-endian integers.
 * PARITY UNPINNED: pairing of adjacent nodes (2i,2i+1)
 * and `depth` = level of the OUTPUT node with root = 0 are assumed; no
 * reference test carries an expected root.
 */
#include "oracle.h"

static void pedersen_merge_bytes(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) {
    fp_t x = fp_from_be_bytes_reduce(a), y = fp_from_be_bytes_reduce(b);
    fp_canonical_be_bytes(or_pedersen_hash(x, y), out);
}

void or_merkle_build(int tree, unsigned n_friendly_layers, int leaf_kind, const uint8_t *leaves,
                     size_t n, uint8_t *nodes, uint8_t *tags) {
    unsigned log_n = 0;
    while (((size_t)1 << log_n) < n) ++log_n;
    const int hk = tree == OR_TREE_KECCAK ? OR_HASH_KECCAK
                 : tree == OR_TREE_KECCAK_M20 ? OR_HASH_KECCAK_M20 : OR_HASH_BLAKE2S_M20;
    or_pedersen_init();   /* before the OpenMP regions below */
    memset(nodes, 0, 64);
    if (tags) memset(tags, 0, 2 * n);
    /* leaf slots */
    for (size_t i = 0; i < n; ++i) {
        if (leaf_kind == OR_LEAF_FELT) {
            fp_t v; memcpy(&v, leaves + 32 * i, 32);
            fp_mont_be_bytes(v, nodes + 32 * (n + i));
        } else {
            memcpy(nodes + 32 * (n + i), leaves + 32 * i, 32);
            if (tags) tags[n + i] = 1;
        }
    }
    /* parents of leaves: depth log_n - 1 */
    unsigned depth = log_n - 1;
#pragma omp parallel for schedule(static) if (n >= 2048)
    for (size_t k = n / 2; k < n; ++k) {
        const uint8_t *l0 = nodes + 32 * (2 * k), *l1 = l0 + 32;
        uint8_t *o = nodes + 32 * k;
        if (leaf_kind == OR_LEAF_FELT) {
            fp_t e[2]; memcpy(&e[0], leaves + 32 * (2 * k - n), 32); memcpy(&e[1], leaves + 32 * (2 * k - n + 1), 32);
            if (tree == OR_TREE_FRIENDLY) {
                /* SingleCol: MerkleTreeImpl<UnhashedLeafConfig<PedersenHashFn>> */
                fp_canonical_be_bytes(or_pedersen_hash_elements(e, 2), o);
            } else {
                or_hash_elements(hk, e, 2, o);
            }
        } else if (tree == OR_TREE_FRIENDLY) {
            if (depth < n_friendly_layers) { pedersen_merge_bytes(l0, l1, o); if (tags) tags[k] = 0; }
            else { or_hash_merge(OR_HASH_BLAKE2S_M20, l0, l1, o); if (tags) tags[k] = 1; }
        } else {
            or_hash_merge(hk, l0, l1, o);
        }
    }
    /* inner levels */
    for (unsigned d = depth; d-- > 0;) {
        size_t lo = (size_t)1 << d, hi = lo << 1;
#pragma omp parallel for schedule(static) if (lo >= 1024)
        for (size_t k = lo; k < hi; ++k) {
            const uint8_t *n0 = nodes + 32 * (2 * k), *n1 = n0 + 32;
            uint8_t *o = nodes + 32 * k;
            if (tree == OR_TREE_FRIENDLY) {
                if (leaf_kind == OR_LEAF_FELT || d < n_friendly_layers) {
                    pedersen_merge_bytes(n0, n1, o); if (tags) tags[k] = 0;
                } else {
                    or_hash_merge(OR_HASH_BLAKE2S_M20, n0, n1, o); if (tags) tags[k] = 1;
                }
            } else {
                or_hash_merge(hk, n0, n1, o);
            }
        }
    }
}
