"""One rank = one PROCESS of a proof sharded by the C++ host (sandstorm_amd/host/sharded.cpp, ssh_prove_sharded): the path
`bench.py --gpus N` takes by default, with the ranks meeting in the driver's CallbackTransport over torch.distributed
(gloo) instead of RCCL.  Launched under torch.distributed.run by tests/hipemu/extra_sharded_host_procs.py (the device code on
the CPU: every process loads the emulated library) and by tests/test_gpu_sharded_host.py (the MI355X: the processes share the GPU,
the exchanges are staged through the host).  Each process runs its own coin in lock step, deals its columns, and enters every
exchange - what eight processes on eight GPUs do.

argv: case out_path [repeat]
  mini:<log_n>:<max_remainder>            tests/mini_air.py, masked-Keccak trees + Solidity coin
  mini-cairo:<log_n>:<max_remainder>:<N>  the same under FriendlyMerkleTree<N> + the Cairo coin
  recursive:<log_steps>[:blocks]          the reference's example padded to 2^log_steps steps, the real recursive AIR, CairoVerifierClaim
  starknet:<log_steps>[:blocks]           the same run re-declared for the starknet layout, the real starknet AIR, the Eth claim's parts
                                          (blocks: the extension trace as row blocks on every rank, hostlib.build_extension_blocks)
  selfcheck[:corrupt]                     only the group's self check (hostlib.group_self_check); corrupt: rank 1's all-to-all flips one
                                          received byte - the check must fail on every rank, naming rank 1 (out_path: one verdict line per rank)
rank 0 writes the proof (reference wire format) to out_path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    case, out_path = sys.argv[1], sys.argv[2]
    repeat = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    world = int(os.environ["WORLD_SIZE"])
    emulated = os.environ.get("SS_TEST_HIPEMU") == "1"
    if emulated:
        # the emulator spreads a launch's workgroups over OS threads: the ranks are the parallelism here
        os.environ.setdefault("HIPEMU_THREADS", str(max(1, (os.cpu_count() or 1) // world)))
        os.environ.setdefault("SS_PED_WINDOW", "16")
        os.environ.setdefault("SS_PED_SMALL_MAX", "128")
    import torch                                             # before the library: one HIP runtime per process
    import torch.distributed as dist
    from sandstorm_amd import _lib
    if emulated:
        _lib.LIB_PATH = os.environ.get("SS_TEST_HIPEMU_LIB", os.path.join(ROOT, "tests", "hipemu", "_build", "libsandstorm_hipemu.so"))
    from sandstorm_amd import backend as be, hostlib
    from tests import mini_air_host, sharded_host_cases as cases
    mini_air_host.register()
    dist.init_process_group(backend="gloo")
    rank = dist.get_rank()
    assert dist.get_world_size() == world
    kind = case.split(":")
    if kind[0] == "selfcheck":
        ctx = be.Context(0)
        group = hostlib.torch_dist_group()
        if kind[1:] == ["corrupt"] and rank == 1:
            honest = group._all_to_all

            def flip(send, sc, rc):
                got = honest(send, sc, rc).copy()
                got[len(got) // 2] ^= 0x10
                return got
            group._all_to_all = flip
        try:
            rate = hostlib.group_self_check(ctx, rank, world, group, 1 << 16)
            assert rate > 0 or world == 1
            verdict = "PASSED"
        except _lib.SandstormHipError as e:
            verdict = "REFUSED rank %d: %s" % (rank, e)
        got = [None] * world
        dist.all_gather_object(got, verdict)
        if rank == 0:
            with open(out_path, "w") as f:
                f.write("\n".join(got))
            print("SHARDED_PROOF_WRITTEN")
        group.close()
        ctx.close()
        dist.destroy_process_group()
        return
    if kind[0] == "mini":
        make, _ = cases.mini_case(int(kind[1]), int(kind[2]))
    elif kind[0] == "mini-cairo":
        make, _ = cases.mini_case(int(kind[1]), int(kind[2]), "cairo", int(kind[3]))
    elif kind[0] in ("recursive", "starknet"):
        make_case, _ = (cases.recursive_case if kind[0] == "recursive" else cases.starknet_case)(int(kind[1]))
        make = (lambda w: make_case(w, blocks=True)) if kind[2:] == ["blocks"] else make_case
    else:
        raise SystemExit("unknown case %r" % case)
    ctx = be.Context(0)
    air, tree_kind, nf, coin_kind, seed, mine, log_n, ext, opt = make(world)(rank, ctx)
    group = hostlib.torch_dist_group()
    ext.group = group
    hostlib.group_self_check(ctx, rank, world, group)       # as bench.py does before its warm-up
    proof = None
    try:
        for _ in range(repeat):                              # a group outlives a proof
            proof = hostlib.prove_sharded(ctx, air, tree_kind, nf, coin_kind, seed, rank, world, group, mine, log_n, ext, opt,
                                          extension_blocks=getattr(ext, "blocks", None))
            assert (proof is not None) == (rank == 0)
    finally:
        group.close()
        for m in getattr(ext, "matrices", []):
            m.close()
        air.close()
    if rank == 0:
        with open(out_path, "wb") as f:
            f.write(proof)
        print("SHARDED_PROOF_WRITTEN")
    dist.barrier()
    del mine, ext
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
