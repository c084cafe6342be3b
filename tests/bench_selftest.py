"""`SS_BENCH_SELFTEST=1 python bench.py --gpus N`: bench.py's own launcher, rendezvous, timing protocol and one-line contract,
exercised WITHOUT a GPU (tests/test_bench_launcher.py).  The ranks are gloo processes; each runs the real sharded driver
(sandstorm_amd/sharded_prover.py) with the CPU oracle standing in for the HIP kernels - test infrastructure, as in
tests/dist_prove_worker.py.  The line it prints is labelled as what it is: not a measurement of anything."""
import os

import torch
import torch.distributed as dist

from tests import dist_prove_worker as w


def run(args, rank, world, timed_steps, emit):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    n, cols, claim, opt, ext, seed, leaf_hash = w.mini(9, 4)
    comm = w.Comm(device=torch.device("cpu"))
    prover = w.ShardedProver(w.CpuContext(), claim, comm, opt)
    mine = {c: w.tensor(v) for c, v in cols.items() if c % world == rank}
    proofs = []

    def step():
        proofs.append(prover.prove(seed, mine, lambda ch: ext(ch, lambda c: c % world == rank), n))

    def all_max(dt):
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    sec = timed_steps(step, args.steps, args.warmup, dist.barrier, all_max)
    if rank == 0:
        assert len(proofs) == args.steps + args.warmup
        raw = w.wire.serialize(w.wire.from_proof(proofs[-1], leaf_hash))
        with open(os.path.join(w.ROOT, "tests", "golden", "mini_proof_eth_log9.bin"), "rb") as f:
            same = f.read() == raw
        emit({"metric": "prove_wall_time_s", "value": sec, "unit": "s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": sec * 1e3, "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "selftest": True,
              "data": "LAUNCHER SELF-TEST: gloo ranks, the CPU oracle behind the sharded driver, a 512-row mini AIR - NOT A MEASUREMENT",
              "proof_is_the_single_device_proof": same, "config": {"workload": "selftest mini:9:4"}})
    dist.barrier()
    dist.destroy_process_group()
