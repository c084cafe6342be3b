"""One proof over R ranks by the C++ host (sandstorm_amd/host/sharded.cpp) with the ranks as THREADS of this process, every thread
with its own context (LocalTransport) - shared by tests/test_gpu_sharded_host.py (the MI355X: the ranks share the GPU) and
tests/hipemu/extra_sharded_host.py (the device code on the CPU).  The proofs must be the single-device proofs, byte for byte."""
import os
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def run_ranks(world, make_rank, group=None, repeat=1):
    """make_rank(rank, ctx) -> (air, tree_kind, n_friendly, coin_kind, seed, my_base, log_n, build_extension, options); one
    thread per rank -> rank 0's proof bytes.  group: a hostlib.LocalGroup (default), or "rccl": every rank makes its RcclGroup from one
    unique id (world must be 1: one process per GPU) and proves `repeat` times over it"""
    from sandstorm_amd import backend as be, hostlib
    own = group is None
    rccl_id = hostlib.rccl_unique_id() if group == "rccl" else None
    if own:
        group = hostlib.LocalGroup(world)
    out, errs = [None] * world, [None] * world

    def body(rank):
        ctx = None
        try:
            ctx = be.Context(0)
            air, tree_kind, nf, coin_kind, seed, mine, log_n, ext, opt = make_rank(rank, ctx)
            grp = hostlib.RcclGroup(ctx, rccl_id, rank, world) if rccl_id is not None else group
            ext.group = grp                      # (the block form of the extension trace exchanges its blocks' totals over it)
            try:
                for _ in range(repeat):          # a communicator outlives a proof (an RCCL unique id serves ONE ncclCommInitRank per rank)
                    out[rank] = hostlib.prove_sharded(ctx, air, tree_kind, nf, coin_kind, seed, rank, world, grp, mine, log_n, ext, opt,
                                                      extension_blocks=getattr(ext, "blocks", None))
            finally:
                if rccl_id is not None:
                    grp.close()                             # everything that lives in the context's pool goes before the context does
                for m in getattr(ext, "matrices", []):
                    m.close()
                air.close()
                del mine, ext
        except BaseException as e:               # noqa: BLE001 - reported below
            errs[rank] = e
        finally:
            if ctx is not None:
                ctx.close()
    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if own:
        group.close()
    for e in sorted((e for e in errs if e is not None), key=lambda e: "another rank failed" in str(e)):
        raise e                                  # the rank that failed first, not the ones it released from their barriers
    assert all(o is None for o in out[1:])
    return out[0]


def mini_case(log_n, max_remainder, flavour="eth", n_friendly=22):
    """tests/mini_air.py through the C++ host's mini AIR"""
    from sandstorm_amd import backend as be, hostlib
    from sandstorm_amd.coin import canonical
    from sandstorm_amd.prover import ProofOptions
    from tests import mini_air
    n = 1 << log_n
    c0, c1 = mini_air.base_trace(n)
    host = {0: np.stack([be.felt(v) for v in c0]), 1: np.stack([be.felt(v) for v in c1])}
    opt = ProofOptions(num_queries=12, grinding_factor=8, fri_max_remainder_coeffs=max_remainder)
    tree, nf, coin = (be.TREE_FRIENDLY, n_friendly, be.COIN_CAIRO) if flavour == "cairo" else (be.TREE_KECCAK_M20, 0, be.COIN_SOLIDITY)

    def make(world):
        def make_rank(rank, ctx):
            air = hostlib.HostAir(ctx, hostlib.AIR_MINI, log_n)
            mine = {c: ctx.column(v) for c, v in host.items() if c % world == rank}
            keep = []

            def ext(challenges):
                if 2 % world != rank:
                    return {}
                keep.append(ctx.column(np.stack([be.felt(v) for v in mini_air.extension_trace(c0, canonical(challenges[0]))])))
                return {2: keep[-1]}
            return air, tree, nf, coin, bytes(range(32)), mine, log_n, ext, opt
        return make_rank
    return make, (tree, nf, coin, opt, host, log_n)


def single_device_mini(ctx, case):
    """the same statement through the single-device C++ prover (hostlib.prove, wire format)"""
    from sandstorm_amd import backend as be, hostlib
    from sandstorm_amd.coin import canonical
    from tests import mini_air
    tree, nf, coin, opt, host, log_n = case
    air = hostlib.HostAir(ctx, hostlib.AIR_MINI, log_n)
    base = [ctx.column(host[0]), ctx.column(host[1])]
    c0 = [canonical(v) for v in host[0]]
    keep = []

    def ext(challenges):
        keep.append(ctx.column(np.stack([be.felt(v) for v in mini_air.extension_trace(c0, canonical(challenges[0]))])))
        return [keep[-1]]
    raw = hostlib.prove(ctx, air, tree, nf, coin, bytes(range(32)), base, log_n, ext, opt, wire=True)
    air.close()
    return raw


def recursive_case(log_steps, claim="cairo"):
    """the reference's example run (padded to 2^log_steps steps) with the real 93-constraint AIR, C++ trace generator; CLI-default
    options; CairoVerifierClaim (FriendlyMerkleTree<22> + Cairo coin) or the Eth claim's parts"""
    from sandstorm_amd import backend as be, binary, hostlib, public_input
    from sandstorm_amd.layouts import recursive as rec
    from sandstorm_amd.prover import ProofOptions
    from tests.test_layout_recursive import recursive_example
    states, memory, pi = recursive_example(log_steps)
    host = hostlib.recursive_base_trace(binary.write_register_states(states), binary.write_memory(memory), pi)
    log_n = log_steps + 4
    n = 1 << log_n
    tree, nf, coin = (be.TREE_FRIENDLY, 22, be.COIN_CAIRO) if claim == "cairo" else (be.TREE_KECCAK, 0, be.COIN_SOLIDITY)
    seed = public_input.public_coin_seed(pi, coin)
    opt = ProofOptions()
    aux_cols = (rec.COL_NPC, rec.COL_MEMORY, rec.COL_RANGE_CHECK, rec.COL_DILUTED_UNORDERED, rec.COL_DILUTED_ORDERED)

    def make(world, blocks=False):
        """blocks: the extension trace as row blocks on every rank (hostlib.build_extension_blocks: the scans divided over the ranks)"""
        def make_rank(rank, ctx):
            air = hostlib.RecursiveHostAir(ctx, pi, log_n, 1)
            mine = {c: ctx.column(v) for c, v in enumerate(host) if c % world == rank}
            my_ext = [c for c in (7, 8, 9) if c % world == rank]
            keep = []
            nb = n // world

            def ext_blocks(challenges):
                aux = [ctx.column(host[c][rank * nb:(rank + 1) * nb]) for c in aux_cols]
                m = hostlib.build_extension_blocks(ctx, "recursive", aux, n, rank, world, ext.group, challenges)
                keep.append(aux)
                ext.matrices.append(m)
                return m.cols

            def ext(challenges):
                if not my_ext:
                    return {}
                aux = [ctx.column(host[c]) for c in aux_cols]
                m = hostlib.build_extension_columns(ctx, "recursive", aux, n, challenges)
                keep.append(aux)
                ext.matrices.append(m)
                return {c: m.cols[c - 7] for c in my_ext}
            ext.matrices = []
            if blocks:
                ext.blocks = ext_blocks
            return air, tree, nf, coin, seed, mine, log_n, ext, opt
        return make_rank
    return make, (tree, nf, coin, opt, host, log_n, pi, seed)


def starknet_case(log_steps):
    """the reference's array-sum run re-declared for the starknet layout and padded to 2^log_steps steps (tests/test_layout_starknet.py
    starknet_example), the real 195-constraint AIR, C++ trace generator; CLI-default options; the Eth claim's parts (masked Keccak trees,
    Solidity coin) - the statement tests/test_gpu_full_size.py proves on one device"""
    from sandstorm_amd import backend as be, binary, hostlib, public_input
    from sandstorm_amd.layouts import starknet as sk
    from sandstorm_amd.prover import ProofOptions
    from tests.test_layout_starknet import starknet_example
    states, memory, spi = starknet_example(log_steps)
    host = hostlib.starknet_base_trace(binary.write_register_states(states), binary.write_memory(memory), spi)
    del states, memory
    log_n = log_steps + 4
    n = 1 << log_n
    tree, nf, coin = be.TREE_KECCAK_M20, 0, be.COIN_SOLIDITY
    seed = public_input.public_coin_seed(spi, coin)
    opt = ProofOptions()
    nb = len(host)
    aux_cols = (sk.COL_NPC, sk.COL_MEMORY, sk.COL_RANGE_CHECK)

    def make(world, blocks=False):
        def make_rank(rank, ctx):
            air = hostlib.StarknetHostAir(ctx, spi, log_n)
            mine = {c: ctx.column(v) for c, v in enumerate(host) if c % world == rank}
            keep = []
            nb_rows = n // world

            def ext_blocks(challenges):
                aux = [ctx.column(host[c][rank * nb_rows:(rank + 1) * nb_rows]) for c in aux_cols]
                m = hostlib.build_extension_blocks(ctx, "starknet", aux, n, rank, world, ext.group, challenges)
                keep.append(aux)
                ext.matrices.append(m)
                return m.cols

            def ext(challenges):
                if nb % world != rank:               # the one extension column is column nb
                    return {}
                aux = [mine[c] if c in mine else ctx.column(host[c]) for c in aux_cols]
                m = hostlib.build_extension_columns(ctx, "starknet", aux, n, challenges)
                keep.append(aux)
                ext.matrices.append(m)
                return {nb: m.cols[0]}
            ext.matrices = []
            if blocks:
                ext.blocks = ext_blocks
            return air, tree, nf, coin, seed, mine, log_n, ext, opt
        return make_rank
    return make, (tree, nf, coin, opt, host, log_n, spi, seed)
