"""Multi-GPU sharding of the hot path — one process per GPU.

Two modes:
  * independent proofs per rank (what bench.py measures: weak scaling, no data-path
    collective at all);
  * one proof across ranks: NTT/LDE stages are sharded by COLUMN (no traffic), the
    commitment by ROW BLOCK: every rank builds the Merkle sub-tree of its contiguous row
    block, the sub-tree roots (32 bytes + tag each) are all-gathered — the only exchange
    a commitment needs — and the log2(world) top levels are recomputed by every rank on
    the host (<= 7 hashes).  Nothing here is a sum-reduction, so no all-reduce.

The collectives go through torch.distributed (backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests).  Host hashing of the top levels uses the coin's host
Keccak / Blake2s / Pedersen (product code, sandstorm_amd/coin.py).
"""
import numpy as np

from . import backend as be
from .coin import blake2s256, canonical, keccak256


def column_partition(ncols, world):
    """columns of rank r: contiguous, sizes differ by at most one (10 columns over 8 GPUs: 2,2,1,1,1,1,1,1)"""
    base, extra = divmod(ncols, world)
    out, start = [], 0
    for r in range(world):
        cnt = base + (1 if r < extra else 0)
        out.append(list(range(start, start + cnt)))
        start += cnt
    return out


def row_block(nrows, rank, world):
    assert nrows % world == 0 and world & (world - 1) == 0, "row blocks need a power-of-two world size"
    blk = nrows // world
    return rank * blk, (rank + 1) * blk


def subtree_friendly_layers(n_friendly_layers, world):
    """A rank's sub-tree root sits at global depth log2(world): its local depth d is the
    global depth d + log2(world), so `depth < N_FRIENDLY` becomes `d < N_FRIENDLY - log2(world)`."""
    return max(0, n_friendly_layers - (world.bit_length() - 1))


def merge_nodes(tree_kind, n_friendly_layers, depth, left, right):
    """One inner node on the host: MerkleTreeConfig::hash_nodes (crypto/src/merkle/mixed.rs:106-125,
    mod.rs:430-432).  left/right: (32 bytes, tag).  Returns (32 bytes, tag)."""
    (a, ta), (b, tb) = left, right
    if tree_kind == be.TREE_KECCAK:
        return keccak256(a + b), 0
    if tree_kind == be.TREE_KECCAK_M20:
        return keccak256(a + b)[:20] + bytes(12), 0
    if depth < n_friendly_layers:            # Pedersen (hash_boundary reads Blake2s digests as big-endian integers)
        x = be.felt(int.from_bytes(a, "big") % be.P)
        y = be.felt(int.from_bytes(b, "big") % be.P)
        return canonical(be.pedersen_hash_host(x, y)).to_bytes(32, "big"), 0
    assert ta == 1 and tb == 1
    return bytes(12) + blake2s256(a + b)[12:], 1


def combine_subtree_roots(tree_kind, n_friendly_layers, roots):
    """roots: list of (32 bytes, tag) in rank order -> the root of the whole tree."""
    level = list(roots)
    depth = (len(level).bit_length() - 1) - 1          # depth of the parents of the gathered roots
    while len(level) > 1:
        level = [merge_nodes(tree_kind, n_friendly_layers, depth, level[2 * i], level[2 * i + 1])
                 for i in range(len(level) // 2)]
        depth -= 1
    return level[0]


def all_gather_roots(local_root, local_tag, device=None):
    """all-gather of the (root, tag) pairs over the default process group (33 bytes per rank)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    mine = torch.tensor(list(local_root) + [local_tag], dtype=torch.uint8, device=device)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return [(bytes(t[:32].tolist()), int(t[32])) for t in (o.cpu() for o in out)]


def sharded_commit(build_local_subtree, tree_kind, n_friendly_layers, device=None):
    """build_local_subtree(n_friendly_local) -> (root bytes, tag) for this rank's row block."""
    import torch.distributed as dist
    world = dist.get_world_size()
    root, tag = build_local_subtree(subtree_friendly_layers(n_friendly_layers, world))
    roots = all_gather_roots(root, tag, device)
    return combine_subtree_roots(tree_kind, n_friendly_layers, roots)
