"""`SS_BENCH_SELFTEST=1 python bench.py --gpus N`: bench.py's own launcher, rendezvous, timing protocol and one-line contract,
exercised WITHOUT a GPU (tests/test_bench_launcher.py).  The ranks are gloo processes running what the real run's ranks run by
default - the C++ host's sharded prover (sandstorm_amd/host/sharded.cpp, `--sharded-host cpp`), group self check first - with the
device code on the CPU (tests/hipemu: test infrastructure) and the driver's CallbackTransport over gloo where the MI355X run has
RCCL.  The line it prints is labelled as what it is: not a measurement of anything."""
import os

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, rank, world, timed_steps, emit):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    comm_check = None
    os.environ.setdefault("HIPEMU_THREADS", str(max(1, (os.cpu_count() or 1) // world)))
    os.environ.setdefault("SSH_FRI_SPREAD_MIN_LOG", "6")       # the ranks fold this small proof's first FRI layers together, as they do 2^25-value ones
    from sandstorm_amd import _lib
    _lib.LIB_PATH = os.environ.get("SS_TEST_HIPEMU_LIB", os.path.join(ROOT, "tests", "hipemu", "_build", "libsandstorm_hipemu.so"))
    from sandstorm_amd import backend as be, hostlib
    from tests import mini_air_host, sharded_host_cases as cases
    mini_air_host.register()
    make, _ = cases.mini_case(9, 4)
    ctx = be.Context(0)
    air, tree_kind, nf, coin_kind, seed, mine, log_n, ext, opt = make(world)(rank, ctx)
    group = hostlib.torch_dist_group()
    comm_check = {"ok": True, "exchange_gbps": hostlib.group_self_check(ctx, rank, world, group, 1 << 16)}
    proofs = []

    def step():
        proofs.append(hostlib.prove_sharded(ctx, air, tree_kind, nf, coin_kind, seed, rank, world, group, mine, log_n, ext, opt))
    last = lambda: proofs[-1]
    what = "the C++ host's sharded prover over the emulated device code, CallbackTransport over gloo"
    def all_max(dt):
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    sec = timed_steps(step, args.steps, args.warmup, dist.barrier, all_max)
    if rank == 0:
        assert len(proofs) == args.steps + args.warmup
        with open(os.path.join(ROOT, "tests", "golden", "mini_proof_eth_log9.bin"), "rb") as f:
            same = f.read() == last()
        emit({"metric": "prove_wall_time_s", "value": sec, "unit": "s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": sec * 1e3, "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "selftest": True,
              "data": "LAUNCHER SELF-TEST: gloo ranks, %s, a 512-row mini AIR - NOT A MEASUREMENT" % what,
              "proof_is_the_single_device_proof": same,
              "config": {"workload": "selftest mini:9:4", "sharded_host": args.sharded_host, "comm_self_check": comm_check}})
    dist.barrier()
    group.close()
    air.close()
    del mine, ext
    ctx.close()
    dist.destroy_process_group()
