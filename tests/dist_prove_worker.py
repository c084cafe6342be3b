"""Worker of tests/test_sharded.py: one rank of a sharded proof, under torch.distributed.run with the gloo backend.
The real driver (sandstorm_amd/sharded_prover.py) runs on every rank; the CPU oracle (oracle/cpu_context.py) stands in
for the HIP kernels only.  Rank 0 writes the proof (reference wire format) to argv[2]."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle_py as oracle                      # noqa: E402  (checker; stands in for the kernels)
from oracle.cpu_context import CpuContext                   # noqa: E402
from sandstorm_amd import backend as be, wire               # noqa: E402
from sandstorm_amd.coin import canonical, keccak256         # noqa: E402
from sandstorm_amd.prover import Claim, ProofOptions        # noqa: E402
from sandstorm_amd.sharded_prover import Comm, ShardedProver  # noqa: E402


def tensor(limbs):
    return torch.from_numpy(np.ascontiguousarray(limbs, dtype=np.uint64).view(np.int64).copy())


def blake2s_m20_leaf_hash(vals):
    import hashlib
    return bytes(12) + hashlib.blake2s(b"".join((v * wire._R % wire.P).to_bytes(32, "big") for v in vals)).digest()[12:]


def friendly_tree(n_friendly):
    """FriendlyMerkleTree<n_friendly, _>: a small N puts the Blake2s / Pedersen boundary inside a 2^10-leaf tree"""
    return type("FriendlyMerkleTree%d" % n_friendly, (be.FriendlyMerkleTree,), {"n_friendly": n_friendly})


def mini(log_n, max_remainder, flavour="eth", n_friendly=22):
    from tests import mini_air
    n = 1 << log_n
    c0, c1 = mini_air.base_trace(n)
    cols = {0: oracle.to_mont(c0), 1: oracle.to_mont(c1)}
    if flavour == "cairo":                          # CairoVerifierClaim's parts (src/claims.rs:31-32) around the mini AIR
        claim = Claim(mini_air.make_air(oracle.to_mont), friendly_tree(n_friendly), be.COIN_CAIRO)
    else:
        claim = Claim(mini_air.make_air(oracle.to_mont), be.LeafVariantMerkleTree, be.COIN_SOLIDITY)
    opt = ProofOptions(num_queries=12, grinding_factor=8, fri_max_remainder_coeffs=max_remainder)

    def ext(challenges, owner):
        return {2: tensor(oracle.to_mont(mini_air.extension_trace(c0, canonical(challenges[0]))))} if owner(2) else {}

    def leaf_hash(vals):
        return keccak256(b"".join((v * wire._R % wire.P).to_bytes(32, "big") for v in vals))[:20] + bytes(12)
    return n, cols, claim, opt, ext, bytes(range(32)), blake2s_m20_leaf_hash if flavour == "cairo" else leaf_hash


def example():
    """the reference's array-sum run, recursive layout, the real 93-constraint AIR (tests/golden/array_sum_recursive_eth.proof)"""
    from sandstorm_amd import extension, public_input
    from sandstorm_amd.layouts import recursive as rec
    from tests.test_layout_recursive import load_run
    states, memory, pi = load_run()
    host = [oracle.to_mont(c) for c in rec.base_trace(states, memory, pi)]
    n = len(host[0])
    ctx = CpuContext()
    claim = Claim(rec.make_air(ctx, pi, n), be.LeafVariantMerkleTreeUnmasked, be.COIN_SOLIDITY)
    opt = ProofOptions(num_queries=12, grinding_factor=8)

    def ext(challenges, owner):
        # the extension columns come out of one scan each over auxiliary base columns: built where they are owned, from
        # the host trace (the trace generator's output), like the base columns themselves
        full = [ctx.column(c) for c in host]
        m = extension.build_extension_columns("recursive", ctx, rec.trace_columns(ctx, full, n), challenges)
        return {7 + k: tensor(m.cols[k].download(np.uint64, (n, 4))) for k in range(3) if owner(7 + k)}

    def leaf_hash(vals):
        return keccak256(b"".join((v * wire._R % wire.P).to_bytes(32, "big") for v in vals))
    return n, dict(enumerate(host)), claim, opt, ext, public_input.public_coin_seed(pi, be.COIN_SOLIDITY), leaf_hash


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    case, out_path = sys.argv[1], sys.argv[2]
    if case == "example":
        n, cols, claim, opt, ext, seed, leaf_hash = example()
    else:
        f = case.split(":")                         # mini:<log n>:<max remainder>[:cairo:<N of FriendlyMerkleTree<N>>]
        n, cols, claim, opt, ext, seed, leaf_hash = mini(int(f[1]), int(f[2]), *((f[3], int(f[4])) if len(f) > 3 else ()))
    comm = Comm(device=torch.device("cpu"))
    prover = ShardedProver(CpuContext(), claim, comm, opt)
    mine = {c: tensor(v) for c, v in cols.items() if c % world == rank}
    proof = prover.prove(seed, mine, lambda ch: ext(ch, lambda c: c % world == rank), n)
    if rank == 0:
        with open(out_path, "wb") as f:
            f.write(wire.serialize(wire.from_proof(proof, leaf_hash)))
        print("SHARDED_PROOF_WRITTEN")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
