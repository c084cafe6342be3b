#!/usr/bin/env python3
"""bench.py — the hot path on synthetic traces, one rank per GPU.

    python bench.py --gpus N --steps K --warmup W [--workload starknet_2p20]

A step = one pass of the GPU hot path over one synthetic trace batch resident
in HBM (see `config.stages` in the output for exactly which stages are inside
the timed region).  Prints ONE JSON line on rank 0 (contract in the task
statement): whole-job Fp field-ops/s of the LDE NTTs, plus `roofline` (dominant
kernel, HIP-event timed) and `cpu_baseline` (the CPU oracle on a bounded sample,
timed on this box's host cores — a *port*, not the reference binary).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # first: its bundled HIP runtime must be the one the C ABI library binds to
import torch.distributed as dist

WORKLOADS = {
    # name: (log2 trace rows n, trace columns, description)        rows n = 16 * steps
    "starknet_2p20": (24, 10, "starknet layout shape, 2^20 steps: 10 columns x 2^24 rows, LDE blowup 2"),
    "recursive_2p16": (20, 10, "recursive layout shape, 2^16 steps: 10 columns x 2^20 rows, LDE blowup 2"),
    "tiny": (12, 10, "plumbing"),
}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def ntt_field_ops(log_size):
    """1.5 * N * log2 N field operations per size-N transform"""
    return 1.5 * (1 << log_size) * log_size


def synth_columns(device, ncols, log_n, seed):
    """uint64[ncols, n, 4] random Montgomery images < p, generated on the GPU"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n = 1 << log_n
    t = torch.randint(0, 2**63 - 1, (ncols, n, 4), dtype=torch.int64, device=device, generator=g)
    t[:, :, 3] &= (1 << 59) - 1          # < 2^251 < p
    return t


def cpu_baseline(sample_log_n, sample_cols):
    """The CPU oracle (a port of the reference's algorithm, OpenMP) on a bounded sample."""
    import numpy as np
    from oracle import oracle_py as oracle
    from tests.util import random_column
    g = oracle.to_mont([3])[0]
    n = 1 << sample_log_n
    cols = [random_column(n, c) for c in range(sample_cols)]
    oracle.lde(cols[0][:1024], 1, g)    # warm: build tables, spin up OpenMP
    t0 = time.perf_counter()
    for c in cols:
        oracle.lde(c, 1, g)
    dt = time.perf_counter() - t0
    ops = sample_cols * (ntt_field_ops(sample_log_n) + ntt_field_ops(sample_log_n + 1))
    return {"value": ops / dt / 1e9, "unit": "Gfield-ops/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "oracle LDE (iNTT n + coset NTT 2n, OpenMP) of %d columns x 2^%d rows, %.1f s"
                      % (sample_cols, sample_log_n, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="starknet_2p20", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node N for --gpus N"
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: there is no CPU fallback for the hot path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    from sandstorm_amd import backend as be
    log_n, ncols, desc = WORKLOADS[args.workload]
    log_blowup = 1
    n, N = 1 << log_n, 1 << (log_n + log_blowup)

    ctx = be.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
    offset = be.felt(3)

    # inputs resident in HBM before the timed region (every rank: its own trace -> weak scaling)
    trace = synth_columns(device, ncols, log_n, seed=0x53414E44 + rank)
    evals = torch.empty((ncols, N, 4), dtype=torch.int64, device=device)
    coeffs = torch.empty((ncols, n, 4), dtype=torch.int64, device=device)
    t_cols = [trace[c] for c in range(ncols)]
    e_cols = [evals[c] for c in range(ncols)]
    c_cols = [coeffs[c] for c in range(ncols)]

    def step():
        ctx.lde(t_cols, log_n, log_blowup, offset, e_cols, c_cols)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):          # also builds the twiddle plans
        step()
    barrier()
    ctx.profile(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    ntt_ms, ntt_launches = ctx.profile_read(be.PROF_NTT_PASS)
    ctx.profile(False)

    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    ops_per_step = ncols * (ntt_field_ops(log_n) + ntt_field_ops(log_n + log_blowup))
    value = world * ops_per_step * args.steps / dt / 1e9

    if rank == 0:
        # roofline of the dominant kernel family (ntt_pass_kernel): algorithmic bytes per
        # launch = SURVEY §8d's 2*N*32 B per transform, shared equally by the passes of that
        # transform, x columns per launch.  Summed over the timed region:
        algo_bytes = args.steps * ncols * 32.0 * (2 * n + 2 * N)
        achieved = algo_bytes / (ntt_ms * 1e-3) / 1e9 if ntt_ms > 0 else 0.0
        # bytes every pass really streams (each pass reads and writes every column once; the
        # expanding pass reads n and writes N), for reference
        def passes(lg):
            r0 = min(11, lg)
            return 1 + (0 if lg <= r0 else math.ceil((lg - r0) / 7))
        streamed = args.steps * ncols * 32.0 * (2 * n * passes(log_n) + (n + N) + 2 * N * (passes(log_n + log_blowup) - 1))
        out = {
            "metric": "fp252_lde_ntt_gfield_ops_per_s", "value": value, "unit": "Gfield-ops/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u256 (Fp252 Montgomery, 8x u32 limbs)", "data": "synthetic",
            "config": {"workload": args.workload, "description": desc,
                       "stages": ["iNTT n (x%d cols)" % ncols, "coset NTT 2n, offset 3 (x%d cols)" % ncols],
                       "trace_rows_log2": log_n, "columns": ncols, "lde_blowup": 2,
                       "per_gpu": "one full trace per rank"},
            "roofline": {"bound": "hbm", "kernel": "ss::ntt_pass_kernel", "achieved": achieved,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": None, "launches": ntt_launches,
                         "avg_launch_ms": ntt_ms / max(1, ntt_launches),
                         "streamed_GBps": streamed / (ntt_ms * 1e-3) / 1e9 if ntt_ms > 0 else 0.0,
                         "mulmod_per_s": args.steps * ncols * ((n // 2) * log_n + (N // 2) * (log_n + log_blowup) - n // 1 * 0) / (ntt_ms * 1e-3) if ntt_ms > 0 else 0.0,
                         "note": "Fp252 butterflies are integer-ALU bound before HBM bound (DESIGN.md)"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(20 if log_n >= 20 else log_n, 4)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
