"""Worker of tests/test_distributed.py: run under torch.distributed.run with the gloo backend."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle_py as oracle                      # noqa: E402  (checker; stands in for the GPU sub-tree build)
from sandstorm_amd import backend as be, sharding           # noqa: E402
from tests.util import random_column                        # noqa: E402


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n = 64
    cols = [random_column(n, c) for c in range(3)]
    ok = True
    for tree_kind, row_kind, nf in ((be.TREE_KECCAK_M20, be.HASH_KECCAK_M20, 0), (be.TREE_KECCAK, be.HASH_KECCAK, 0),
                                    (be.TREE_FRIENDLY, be.HASH_BLAKE2S_M20, 22), (be.TREE_FRIENDLY, be.HASH_BLAKE2S_M20, 3),
                                    (be.TREE_FRIENDLY, be.HASH_BLAKE2S_M20, 1)):
        leaves = oracle.hash_rows(row_kind, cols)
        full_nodes, full_tags = oracle.merkle_build(tree_kind, nf, 0, leaves)
        lo, hi = sharding.row_block(n, rank, world)

        def build_local(nf_local):
            nodes, tags = oracle.merkle_build(tree_kind, nf_local, 0, leaves[lo:hi])
            return bytes(nodes[1]), int(tags[1]) if tree_kind == be.TREE_FRIENDLY else 0
        root, tag = sharding.sharded_commit(build_local, tree_kind, nf)
        ok &= root == bytes(full_nodes[1])
        if tree_kind == be.TREE_FRIENDLY:
            ok &= tag == int(full_tags[1])
    parts = sharding.column_partition(10, world)
    ok &= sorted(sum(parts, [])) == list(range(10)) and max(map(len, parts)) - min(map(len, parts)) <= 1
    import torch
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("DIST_OK" if int(flag) == 1 else "DIST_FAIL")
    dist.destroy_process_group()
    sys.exit(0 if int(flag) == 1 else 1)


if __name__ == "__main__":
    main()
