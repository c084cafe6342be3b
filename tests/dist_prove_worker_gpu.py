"""Worker of tests/test_gpu_sharded.py: one rank of a sharded proof with the REAL kernels.  The GPU box has one GPU, so
the ranks share it (each with its own context) and exchange through gloo, staged through host memory; on a multi-GPU
node the same driver runs one rank per GPU over RCCL (bench.py --gpus N)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sandstorm_amd import backend as be, wire               # noqa: E402
from sandstorm_amd.coin import canonical, keccak256         # noqa: E402
from sandstorm_amd.prover import Claim, ProofOptions        # noqa: E402
from sandstorm_amd.sharded_prover import Comm, ShardedProver  # noqa: E402


def main():
    backend = sys.argv[3] if len(sys.argv) > 3 else "gloo"
    case, out_path = sys.argv[1], sys.argv[2]
    if os.environ.get("SS_TEST_HIPEMU") == "1":
        # tests/test_device_code_on_host.py: the same ranks over the HOST BUILD OF THE DEVICE CODE (tests/hipemu, test infrastructure) -
        # "device" buffers are CPU tensors, gloo moves them as they are
        from sandstorm_amd import _lib
        _lib.LIB_PATH = os.environ.get("SS_TEST_HIPEMU_LIB", os.path.join(ROOT, "tests", "hipemu", "_build", "libsandstorm_hipemu.so"))
        device = torch.device("cpu")
        dist.init_process_group(backend="gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        ctx = be.Context(0)
    else:
        torch.cuda.set_device(0)
        device = torch.device("cuda", 0)
        dist.init_process_group(backend=backend, **({"device_id": device} if backend == "nccl" else {}))
        rank, world = dist.get_rank(), dist.get_world_size()
        # ONE stream for torch's tensor ops, the collectives and the C ABI's kernels: torch's default stream has handle 0,
        # which ss_ctx_set_stream reads as "the context's own stream" - an explicit stream makes the ordering real
        stream = torch.cuda.Stream(device)
        torch.cuda.set_stream(stream)
        ctx = be.Context(0, stream=stream.cuda_stream)

    def tensor(limbs):
        return torch.from_numpy(np.ascontiguousarray(limbs, dtype=np.uint64).view(np.int64).copy()).to(device)

    def mont(values):                                       # canonical ints -> Montgomery limbs (host, R = 2^256)
        return np.stack([be.felt(v) for v in values])
    if case == "example":
        from sandstorm_amd import extension, public_input
        from sandstorm_amd.layouts import recursive as rec
        from tests.test_layout_recursive import load_run
        states, memory, pi = load_run()
        host = [mont(c) for c in rec.base_trace(states, memory, pi)]
        n = len(host[0])
        claim = Claim(rec.make_air(ctx, pi, n), be.LeafVariantMerkleTreeUnmasked, be.COIN_SOLIDITY)
        opt = ProofOptions(num_queries=12, grinding_factor=8)
        seed = public_input.public_coin_seed(pi, be.COIN_SOLIDITY)
        cols = dict(enumerate(host))

        def ext(challenges):
            full = [ctx.column(c) for c in host]
            m = extension.build_extension_columns("recursive", ctx, rec.trace_columns(ctx, full, n), challenges)
            return {7 + k: tensor(m.cols[k].download(np.uint64, (n, 4))) for k in range(3) if (7 + k) % world == rank}

        def leaf_hash(vals):
            return keccak256(b"".join((v * wire._R % wire.P).to_bytes(32, "big") for v in vals))
    elif case.startswith("recursive:"):
        # the north-star's claim (cli/src/main.rs:95-99: recursive layout -> CairoVerifierClaim = FriendlyMerkleTree<22> + the
        # Cairo coin) on the example run padded to 2^k steps: C++ trace generator, the C++ host's lowering of the real
        # 93-constraint AIR behind the Python driver, CLI-default options
        import hashlib
        from sandstorm_amd import binary, extension, hostlib, public_input
        from sandstorm_amd.layouts import recursive as rec
        from tests.test_layout_recursive import recursive_example
        log_steps = int(case.split(":")[1])
        states, memory, rpi = recursive_example(log_steps)
        host = hostlib.recursive_base_trace(binary.write_register_states(states), binary.write_memory(memory), rpi)
        del states, memory
        n = len(host[0])
        host_air = hostlib.RecursiveHostAir(ctx, rpi, log_steps + 4, 1)
        claim = Claim(hostlib.prover_air(host_air), be.FriendlyMerkleTree, be.COIN_CAIRO)
        opt = ProofOptions()
        seed = public_input.public_coin_seed(rpi, be.COIN_CAIRO)
        cols = {c: v for c, v in enumerate(host) if c % world == rank}
        my_ext = [c for c in (7, 8, 9) if c % world == rank]
        aux_host = [host[c] for c in (rec.COL_NPC, rec.COL_MEMORY, rec.COL_RANGE_CHECK, rec.COL_DILUTED_UNORDERED, rec.COL_DILUTED_ORDERED)] if my_ext else None
        del host

        def ext(challenges):
            if not my_ext:
                return {}
            aux = [tensor(c) for c in aux_host]
            out = be.Matrix(ctx, [torch.zeros((n, 4), dtype=torch.int64, device=device) for _ in range(3)], n)
            extension.build_extension_columns("recursive", ctx, extension.TraceColumns(aux[0], aux[1], aux[2], n, aux[3], aux[4]), challenges, check=True, out=out)
            return {c: out.cols[c - 7] for c in my_ext}

        def leaf_hash(vals):
            return bytes(12) + hashlib.blake2s(b"".join((v * wire._R % wire.P).to_bytes(32, "big") for v in vals)).digest()[12:]
    elif case.startswith("starknet:"):
        # the reference's array-sum run re-declared for the starknet layout at 2^k steps: C++ trace generator, the C++
        # host's lowering of the real 195-constraint AIR behind the Python driver, EthVerifierClaim, CLI-default options
        from sandstorm_amd import binary, extension, hostlib, public_input
        from sandstorm_amd.layouts import starknet as sk
        from tests.test_layout_starknet import starknet_example
        log_steps = int(case.split(":")[1])
        states, memory, spi = starknet_example(log_steps)
        host = hostlib.starknet_base_trace(binary.write_register_states(states), binary.write_memory(memory), spi)
        del states, memory
        n = len(host[0])
        host_air = hostlib.StarknetHostAir(ctx, spi, log_steps + 4, 1)
        claim = Claim(hostlib.prover_air(host_air), be.LeafVariantMerkleTree, be.COIN_SOLIDITY)
        opt = ProofOptions()
        seed = public_input.public_coin_seed(spi, be.COIN_SOLIDITY)
        cols = {c: v for c, v in enumerate(host) if c % world == rank}
        aux_host = [host[c] for c in (sk.COL_NPC, sk.COL_MEMORY, sk.COL_RANGE_CHECK)] if 9 % world == rank else None
        del host

        def ext(challenges):
            if aux_host is None:
                return {}
            aux = [tensor(c) for c in aux_host]
            out = be.Matrix(ctx, [torch.zeros((n, 4), dtype=torch.int64, device=device)], n)
            extension.build_extension_columns("starknet", ctx, extension.TraceColumns(aux[0], aux[1], aux[2], n), challenges, check=True, out=out)
            return {9: out.cols[0]}

        def leaf_hash(vals):
            return keccak256(b"".join((v * wire._R % wire.P).to_bytes(32, "big") for v in vals))[:20] + bytes(12)
    else:
        from tests import mini_air
        log_n, max_remainder = (int(v) for v in case.split(":")[1:])
        n = 1 << log_n
        c0, c1 = mini_air.base_trace(n)
        cols = {0: mont(c0), 1: mont(c1)}
        claim = Claim(mini_air.make_air(mont), be.LeafVariantMerkleTree, be.COIN_SOLIDITY)
        opt = ProofOptions(num_queries=12, grinding_factor=8, fri_max_remainder_coeffs=max_remainder)
        seed = bytes(range(32))

        def ext(challenges):
            return {2: tensor(mont(mini_air.extension_trace(c0, canonical(challenges[0]))))} if 2 % world == rank else {}

        def leaf_hash(vals):
            return keccak256(b"".join((v * wire._R % wire.P).to_bytes(32, "big") for v in vals))[:20] + bytes(12)
    comm = Comm(device=device)
    prover = ShardedProver(ctx, claim, comm, opt)
    mine = {c: tensor(v) for c, v in cols.items() if c % world == rank}
    proof = prover.prove(seed, mine, ext, n)
    if rank == 0:
        with open(out_path, "wb") as f:
            f.write(wire.serialize(wire.from_proof(proof, leaf_hash)))
        print("SHARDED_PROOF_WRITTEN")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
