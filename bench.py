#!/usr/bin/env python3
"""bench.py — the proving hot path on a synthetic trace, one rank per GPU.

    python bench.py --gpus N --steps K --warmup W [--workload starknet_2p20]

A step = ONE PROOF: every GPU stage of SURVEY.md §3.1 steps 2-9 (LDE of base and
extension columns, row hashing + Merkle trees, composition-constraint evaluation,
composition LDE, OOD evaluation, DEEP composition, FRI layers, proof-of-work,
query openings) on a trace that is already resident in HBM.  The host trace
generation (A1) is outside: base and auxiliary columns are synthetic
random columns; the AIR is the layout's REAL composition constraint (195 constraints
/ 269 mask cells for starknet, 93 / 133 for recursive: sandstorm_amd/host/air_*.cpp,
the program the reference's own proof verifies under) - a constraint program does
not depend on the trace's contents, and every stage's cost is data-independent.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches its own N ranks
(torch.distributed.run, one per GPU) and prints rank 0's line; under an external launcher it runs as one rank.
The default run (starknet_2p20, one GPU) also times the north-star's own configuration - recursive layout, 2^20 steps,
CairoVerifierClaim - for a few proofs and reports it as `north_star` in the same line.

Prints ONE JSON line on rank 0: prove wall-time (s), plus the Fp NTT rate,
`roofline` (NTT pass kernel, HIP-event timed inside the same K proofs) and
`cpu_baseline` (the CPU oracle — a port, not the reference binary — on a bounded
sample of the same stages, on this box's host cores).
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


def effective_cpus():
    """the CPUs this process may actually use: os.cpu_count() capped by the scheduler affinity and by the cgroup's CPU quota
    (cpu.max / cpu.cfs_quota_us).  The GPU boxes SHOW 256 hardware threads under a quota of 16 CPUs: 256 OpenMP threads there are
    16 CPUs' worth of time sliced 256 ways (trace generation 0.55 s against 0.13 s with 16-32 threads: profiles/r04_end_to_end.txt)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                              # cgroup v2: "<quota|max> <period>"
            q, period = f.read().split()
            if q != "max":
                quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, period = float(f.read()), float(g.read())
                if q > 0 and period > 0:
                    quota = q / period
        except (OSError, ValueError):
            pass
    if quota and quota >= 1:
        n = min(n, int(math.ceil(quota)))
    return max(1, n)


HOST_CPUS = effective_cpus()
os.environ.setdefault("OMP_NUM_THREADS", str(HOST_CPUS))       # before any OpenMP runtime starts: the oracle port, the trace generators, torch

import torch  # first: its bundled HIP runtime must be the one the C ABI library binds to
import torch.distributed as dist

WORKLOADS = {
    # name: (layout, log2 steps).  trace rows n = 16 * steps (CYCLE_HEIGHT, recursive/mod.rs:16)
    "starknet_2p20": ("starknet", 20),      # BASELINE.json: the size the metric is quoted on
    "starknet_2p22": ("starknet", 22),      # BASELINE.json configs[3] (8-GPU config) on ONE GPU: ~210 GB of the 288 GB
    "recursive_2p20": ("recursive", 20),    # BASELINE.json north_star target size (2^20-step recursive trace)
    "recursive_2p16": ("recursive", 16),    # BASELINE.json configs[1]
    "recursive_2p10": ("recursive", 10),    # plumbing
    # BASELINE.json configs[0]: the reference's own example run (tests/golden/example/: cairo-run output of array-sum, 2^14
    # steps) with the REAL recursive AIR; C++ host from the raw files (host/trace_recursive.cpp, host/air_recursive.cpp)
    "array_sum_example": ("recursive-real", 14),
    "recursive_2p7": ("recursive", 7),      # 128 steps: the size of the only figure the reference publishes (186 ms)
    # BASELINE.json configs[4]: the 64-bit field variant at 2^20 steps.  What exists for that field is the low-degree extension,
    # the out-of-domain evaluation / DEEP composition and the FRI fold over Fq3 (ss_lde_gl64, ss_ood_eval_gl64x3,
    # ss_deep_compose_gl64x3, ss_fri_fold_gl64x3; the experimental `plain` layout's AIR is not built): a step = the LDE of 10
    # columns of 2^24 rows, blowup 2 - the NTT work of that configuration's trace commitment; the other kernels are timed beside it
    "goldilocks_lde_2p20": ("goldilocks", 20),
    # BASELINE.json configs[4] as a whole proof: the `plain` layout's 47-constraint AIR (5 base columns + one Fq3 extension column =
    # 8 committed coordinate columns) at 2^20 steps through sandstorm_amd/goldilocks.py - this library's own instantiation of the
    # claim the reference builds from un-vendored parts (parity unpinned); synthetic columns, as for the other workloads
    "goldilocks_plain_2p20": ("goldilocks-plain", 20),
    "goldilocks_plain_2p16": ("goldilocks-plain", 16),
}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec

# The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version banner from C stdio
# when a communicator comes up): file descriptor 1 is pointed at stderr for the whole run and the JSON line goes to a
# duplicate of the original stdout.
_REAL_STDOUT = None


def _claim_stdout():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(obj):
    line = json.dumps(obj)
    if _REAL_STDOUT is None:
        print(line, flush=True)
    else:
        _REAL_STDOUT.write(line + "\n")
        _REAL_STDOUT.flush()


def ntt_field_ops(log_size):
    """1.5 * N * log2 N field operations per size-N transform"""
    return 1.5 * (1 << log_size) * log_size


def synth_columns(device, ncols, log_n, seed):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    t = torch.randint(0, 2**63 - 1, (ncols, 1 << log_n, 4), dtype=torch.int64, device=device, generator=g)
    t[:, :, 3] &= (1 << 59) - 1          # < 2^251 < p: any such 256-bit image is a valid Montgomery felt
    return t


def _sample_statement(layout, log_steps):
    """the public input of the bench statements: the reference's example run re-declared for this layout / step count
    (it only feeds constants of the constraint program)"""
    from sandstorm_amd import public_input
    pi = public_input.AirPublicInput.from_json(os.path.join(ROOT, "tests", "golden", "air_public_input_array_sum.json"))
    pi.n_steps = 1 << log_steps
    if layout == "starknet":
        from sandstorm_amd.layouts import starknet as sk
        return sk, sk.example_public_input(pi)
    from sandstorm_amd.layouts import recursive as rec
    return rec, pi


def python_host_proof(ctx, layout, log_steps, proofs=1):
    """One whole proof of the layout's real AIR on synthetic columns through the Python host (sandstorm_amd/prover.py)
    on `ctx` - a backend.Context (the HIP kernels) or the oracle's CpuContext (the CPU port): the same host code either
    way.  -> seconds per proof (the AIR's tables and the input columns are set up outside the timed region, as for the
    GPU runs)."""
    import numpy as np
    from sandstorm_amd import backend as be, extension
    from sandstorm_amd.prover import Claim, ProofOptions, Prover
    from sandstorm_amd.examples import random_column
    L, pi = _sample_statement(layout, log_steps)
    n = 16 << log_steps
    air = L.make_air(ctx, pi, n)
    base = be.Matrix.from_host(ctx, [random_column(n, c) for c in range(air.num_base_columns)])
    aux = [ctx.column(random_column(n, 40 + c)) for c in range(5 if layout == "recursive" else 3)]
    tc = extension.TraceColumns(aux[0], aux[1], aux[2], n, *(aux[3:5] if layout == "recursive" else ()))
    if layout == "recursive":
        tree, coin = be.FriendlyMerkleTree, be.COIN_CAIRO
    else:
        tree, coin = be.LeafVariantMerkleTree, be.COIN_SOLIDITY
    prover = Prover(ctx, Claim(air, tree, coin), ProofOptions())
    build = lambda ch: extension.build_extension_columns(layout, ctx, tc, ch, check=False)
    seed = bytes(range(32))
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(proofs):
        prover.prove(seed, base, build)
    ctx.sync()
    return (time.perf_counter() - t0) / proofs


# The port at the bench's FULL size, measured once outside bench.py on the GPU box's host cores (tools/cpu_full_size.py,
# profiles/r06_cpu_port_full_size.txt): what `value` of the CPU leg is held against.  {layout: (seconds, threads)}
CPU_PORT_FULL_SIZE = {}
try:
    with open(os.path.join(ROOT, "profiles", "cpu_port_full_size.json")) as _f:
        CPU_PORT_FULL_SIZE = json.load(_f)
except (OSError, ValueError):
    pass


def cpu_baseline(layout, log_steps_full, gpu_ctx):
    """The CPU port timed beside the GPU, MEASURED on a whole proof: the oracle (C + OpenMP restatement of every stage,
    oracle/cpu_context.py) driven by the same host code, on the layout's real AIR at a reduced step count - LDE, extension
    scans, row hashing, trees, the constraint program, out-of-domain evaluation, DEEP, FRI, proof of work, openings - and
    the GPU on exactly that sample.  `value` scales the measured CPU time to the bench workload's size (n log n); the
    measured pair is reported as it is, and so is the port's full-size run (once, outside the bench: profiles/cpu_port_full_size.json)
    and what one of its Montgomery products costs on one core - the figure to hold against ark-ff's when bounding the reference's time."""
    threads = int(os.environ.get("OMP_NUM_THREADS", HOST_CPUS))      # set at import: the CPUs the cgroup grants, not the ones it shows
    from oracle import oracle_py
    from oracle.cpu_context import CpuContext
    # starknet needs 2^17 steps before its diluted check fits (its smallest statement); recursive: 2^16 (round 6: the tuned port does it
    # in the time the untuned one took for the example's 2^14)
    sample = min(log_steps_full, 17 if layout == "starknet" else 16)
    mulmod_ns = min(oracle_py.mulmod_ns(1_000_000) for _ in range(3))
    t_cpu = python_host_proof(CpuContext(), layout, sample)
    python_host_proof(gpu_ctx, layout, sample)                      # warm: plans, tables, pool
    t_gpu = python_host_proof(gpu_ctx, layout, sample, proofs=3)
    ls, lf = sample + 4, log_steps_full + 4
    scale, scale_is = float(1 << (lf - ls)) * (lf + 1) / (ls + 1), "n log n"
    full = CPU_PORT_FULL_SIZE.get("%s_2p%d" % (layout, log_steps_full))
    if full and full.get("sample_log_steps") == sample and full.get("sample_seconds"):
        # the port's full-size / sample ratio MEASURED once on a box of this pool (n log n is 11 % off for starknet, 2.4 x too high for
        # the recursive layout, whose Pedersen layers stop at depth 22: profiles/r06_cpu_port_full_size.txt)
        scale, scale_is = full["seconds"] / full["sample_seconds"], "the port's own full-size / sample ratio, measured once (profiles/cpu_port_full_size.json)"
    out = {"value": t_cpu * scale, "unit": "s", "cores": threads, "host_cpus": os.cpu_count(), "cpu_quota": HOST_CPUS, "kind": "port",
           "measured_sample_s": t_cpu, "gpu_same_sample_s": t_gpu, "sample_speedup": t_cpu / t_gpu,
           "port_mulmod_ns_per_core": mulmod_ns, "port_mulmod_per_s_per_core": 1e9 / mulmod_ns,
           "sample": "MEASURED whole proof (every stage incl. constraint program and DEEP) of the %s layout's real AIR at 2^%d steps "
                     "(2^%d trace rows), CLI-default options, by the oracle (C + OpenMP port, not the reference binary) "
                     "through the same Python host as the GPU: %.2f s on %d threads; the GPU on the same sample %.4f s; "
                     "`value` = that CPU time x %.2f (%s, to 2^%d steps)" % (layout, sample, ls, t_cpu, threads, t_gpu, scale, scale_is, log_steps_full)}
    if full:
        out["full_size_measured_once"] = dict(full, note="the port's whole proof at THIS size, run once outside bench.py (tools/cpu_full_size.py) on a box of "
                                                         "this pool: what `value`'s n log n extrapolation is held against")
    # the reference's own CPU prover is Rust (nightly, an un-vendored git crate) and cannot be built in this image, so north_star's
    # ">= 10x the reference CPU prover" has no measured denominator; what IS measured: this port (Montgomery products by the prime's
    # shape, Pedersen from 4-bit window tables as starknet-crypto looks its points up, twiddles precomputed by all threads) and the
    # cost of one of its field products on one core of this box
    out["note"] = ("the reference CPU prover cannot be built here (Rust nightly + un-vendored ministark): this is the oracle's port on the cores the box "
                   "grants; `value` is its measured sample scaled to the full size (see `sample`).  port_mulmod_ns_per_core is one Montgomery product of the port on one core "
                   "(a dependent chain); arkworks' own benches put ark-ff's 4-limb Montgomery product at ~20-30 ns on 3+ GHz x86 - the reference's "
                   "field-bound stages cannot be faster than this port's by more than that ratio, its hash-bound ones (Pedersen: starknet-crypto's "
                   "4-bit windows, as here) are of the same shape")
    return out


def bench_goldilocks(args, log_steps, rank, local_rank, world, device):
    """--workload goldilocks_lde_2p20: the low-degree extension of 10 columns x 2^24 rows over p = 2^64 - 2^32 + 1 (blowup 2):
    per column an inverse transform of size n and a coset transform of size 2n.  8-byte elements: this is the field where
    the HBM roofline is the binding one.  One independent batch per rank (weak scaling)."""
    import numpy as np
    from sandstorm_amd import backend as be
    log_n, lb, ncols = log_steps + 4, 1, 10
    n = 1 << log_n
    ctx = be.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device=device)
    g.manual_seed(0x474C + rank)
    cols = torch.randint(0, 2**62, (ncols, n), dtype=torch.int64, device=device, generator=g)       # < 2^62 < p
    evals = torch.zeros((ncols, n << lb), dtype=torch.int64, device=device)
    coeffs = torch.zeros((ncols, n), dtype=torch.int64, device=device)

    def step():
        ctx.lde_gl64([cols[c] for c in range(ncols)], log_n, lb, 7, [evals[c] for c in range(ncols)], [coeffs[c] for c in range(ncols)])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(max(1, args.warmup)):
        step()
    barrier()
    ctx.profile(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    ntt_ms, launches = ctx.profile_read(be.PROF_NTT_PASS)
    ctx.profile(False)
    # the field's other kernels on the same columns, timed beside the headline (not part of `value`): out-of-domain
    # evaluation and DEEP composition over Fq3 for a plain-layout-sized mask, then the FRI layers down to <= 16 x blowup points
    other = {}
    if rank == 0:
        def timed(name, fn):
            fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            other[name] = (time.perf_counter() - t) * 1e3
        rng = np.random.default_rng(3)
        mask = [(c, o) for c in range(ncols) for o in (0, 1, 2, 3, 16)]
        mc, mo = [c for c, _ in mask], [o for _, o in mask]
        felt3 = lambda k: rng.integers(0, 2**62, size=(k, 3), dtype=np.uint64)
        z = felt3(1)[0]
        ood = [None]
        timed("ood_eval_fq3", lambda: ood.__setitem__(0, ctx.ood_eval_gl64x3([coeffs[c] for c in range(ncols)], log_n, mc, mo, z)))
        deep = torch.zeros(((n << lb), 3), dtype=torch.int64, device=device)
        coef, none = felt3(len(mask)), np.zeros((0, 3), dtype=np.uint64)
        timed("deep_compose_fq3", lambda: ctx.deep_compose_gl64x3([evals[c] for c in range(ncols)], [], log_n, lb, 7, mc, mo, ood[0], coef, none, none, z, z, deep))
        layers = [deep]
        ll = log_n + lb
        while (1 << ll) > 16 << lb:
            layers.append(torch.zeros(((1 << ll) // 8, 3), dtype=torch.int64, device=device))
            ll -= 3

        def fold_all():
            ll, off = log_n + lb, 7
            for k in range(len(layers) - 1):
                ctx.fri_fold_gl64x3(layers[k], ll, 8, felt3(1)[0], off, layers[k + 1], be.FRI_UNNORMALISED)
                ll, off = ll - 3, pow(off, 8, 2**64 - 2**32 + 1)
        timed("fri_fold_fq3_all_layers", fold_all)
        other["mask_cells"], other["fri_layers"] = len(mask), len(layers) - 1
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    sec = dt / args.steps
    if rank == 0:
        ops = ncols * (ntt_field_ops(log_n) + ntt_field_ops(log_n + lb))
        algo = 8.0 * ncols * (2 * n + 2 * (n << lb))                 # 2 N e per transform, e = 8 bytes
        kern_s = ntt_ms * 1e-3 / args.steps
        passes = launches / args.steps
        achieved = algo / kern_s / 1e9
        out = {"metric": "lde_wall_time_s", "value": sec, "unit": "s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": sec * 1e3, "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
               "dtype": "u64 (p = 2^64 - 2^32 + 1)", "data": "synthetic", "ntt_gfield_ops_per_s": world * ops / kern_s / 1e9,
               "config": {"workload": args.workload, "field": "p = 2^64 - 2^32 + 1 (Goldilocks), 8-byte elements", "columns": ncols,
                          "trace_rows_log2": log_n, "blowup": 2, "per_gpu": "one independent batch per rank",
                          "other_stages_ms": other,
                          "note": "the LDE of BASELINE.json configs[4]'s trace; out-of-domain evaluation, DEEP composition and the FRI folds "
                                  "over Fq3 are timed beside it (other_stages_ms); the rest of that configuration (the `plain` layout's "
                                  "constraints over Fq3, SHA-256 trees, ministark's generic coin) is not built"},
               "roofline": {"bound": "hbm", "kernel": "ss::gl_ntt_pass_kernel", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "launches": launches, "avg_launch_ms": ntt_ms / max(1, launches),
                            "algorithmic_bytes_per_launch": algo / max(1.0, passes),
                            "streamed_bytes_per_s": 8.0 * ncols * sum(2 * (1 << ln) * len(_gl_passes(ln)) for ln in (log_n, log_n + lb)) / kern_s,
                            "note": "algorithmic bytes = 2 N 8 B per transform (SURVEY 8d); a transform of 2^24 (2^25) points is "
                                    "%d (%d) passes of 16 B per element, so the kernel's own stream rate is `streamed_bytes_per_s`"
                                    % (len(_gl_passes(log_n)), len(_gl_passes(log_n + lb)))}}
        if not args.no_cpu_baseline and world == 1:
            from oracle import oracle_py as oracle
            sl = 20
            col = np.random.default_rng(1).integers(0, 2**62, size=1 << sl, dtype=np.uint64)
            t0 = time.perf_counter()
            for _ in range(4):
                oracle.gl_lde(col, lb, 7)
            t_cpu = (time.perf_counter() - t0) / 4
            scale = ncols * float(1 << (log_n - sl)) * (log_n + 0.5) / (sl + 0.5)
            out["cpu_baseline"] = {"value": t_cpu * scale, "unit": "s", "cores": int(os.environ.get("OMP_NUM_THREADS", HOST_CPUS)), "kind": "port",
                                   "sample": "oracle (C + OpenMP) LDE of one 2^%d-row column: %.3f s, scaled n log n to %d columns of 2^%d rows"
                                             % (sl, t_cpu, ncols, log_n)}
        emit(out)
    if world > 1:
        dist.destroy_process_group()


def _gl_passes(log_n, log_tile=13):
    """the pass split of csrc/capi.hip gl_plan_passes: 13 stages in the contiguous pass, then <= 8 per strided pass"""
    r0 = min(log_n, log_tile)
    rem, out = log_n - r0, [r0]
    if rem:
        k = -(-rem // (log_tile - 5))
        for i in range(k):
            r = -(-rem // (k - i))
            out.append(r)
            rem -= r
    return out


def bench_goldilocks_plain(args, log_steps, rank, local_rank, world, device):
    """--workload goldilocks_plain_2p20: ONE proof of the plain layout's real AIR over p = 2^64 - 2^32 + 1 with Fq3 challenges at
    2^20 steps (2^24 rows, 8 committed coordinate columns, blowup 2): LDE, row hashing + trees, the constraint program, composition
    split, out-of-domain evaluation, DEEP over Fq3, FRI (fold 8), proof of work, openings - and the extension column's running
    products between the two trace commitments.  Base columns are synthetic (a constraint program's cost does not depend on the
    trace).  One independent proof per rank."""
    import numpy as np
    from sandstorm_amd import backend as be, goldilocks as gs
    from sandstorm_amd.layouts import plain as pl
    n = 16 << log_steps
    stream = torch.cuda.Stream(device)            # one stream for torch's tensor ops and the C ABI's kernels
    torch.cuda.set_stream(stream)
    ctx = be.Context(local_rank, stream=stream.cuda_stream)
    prog = pl.example_program(10)
    states, memory = pl.run(prog, 64)
    pi = pl.public_input_of(prog, states, memory)
    pi.n_steps = 1 << log_steps                   # the statement only feeds constants of the program
    air, opt = gs.plain_air(), gs.Options()
    g = torch.Generator(device=device)
    g.manual_seed(0x504C + rank)
    cols = torch.randint(0, 2**62, (5, n), dtype=torch.int64, device=device, generator=g)
    prover = gs.Prover(ctx, air, opt)
    seed = bytes((11 * i) & 0xff for i in range(32))
    base = [cols[c] for c in range(5)]
    # the extension column's running products are built on the device inside the timed region (check=False: synthetic columns are
    # not permutations of each other)
    build_ext = lambda ch: gs.plain_extension_on_device(ctx, base, ch, check=False)[0]
    tables = air.make_tables(n, opt.log_blowup)     # periodic columns of (n, blowup): constants like the twiddles, kept across proofs
    if args.gl_host == "cpp":
        from sandstorm_amd import hostlib
        step = lambda: hostlib.gl_prove(ctx, air, opt, seed, base, build_ext, tables=tables, statement=pi)
    else:
        step = lambda: prover.prove(seed, base, build_ext, statement=pi, tables=tables)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    proof = step()
    for _ in range(args.warmup):
        step()
    barrier()
    ctx.profile(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    kinds = [("ntt_pass", be.PROF_NTT_PASS), ("hash_rows", be.PROF_HASH_ROWS), ("merkle", be.PROF_MERKLE), ("fri_fold", be.PROF_FRI),
             ("quotient", be.PROF_QUOTIENT), ("deep", be.PROF_DEEP), ("extension_scans", be.PROF_EXT)]
    prof = {name: ctx.profile_read(k) for name, k in kinds}
    ctx.profile(False)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    sec = dt / args.steps
    if rank == 0:
        ntt_ms, launches = prof["ntt_pass"]
        # HBM bytes of the transform's launches from the PMC passes of this workload (tools/gl64_pmc.sh; FETCH_SIZE / WRITE_SIZE cannot
        # be read live): the committed summary is cited
        traffic, tpath = {}, os.path.join(ROOT, "profiles", "hbm_traffic_%s.json" % args.workload)
        if os.path.exists(tpath):
            with open(tpath) as fh:
                traffic = json.load(fh)
        emit({"metric": "prove_wall_time_s", "value": sec, "unit": "s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": sec * 1e3, "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
              "dtype": "u64 (p = 2^64 - 2^32 + 1), challenges in Fq3", "data": "synthetic", "proofs_per_s": world / sec,
              "config": {"workload": args.workload, "layout": "plain (47 constraints, %d mask cells over 8 coordinate columns)" % len(air.mask),
                         "steps_log2": log_steps, "trace_rows_log2": log_steps + 4, "blowup": 2,
                         "proof_options": "%d queries, blowup 2, %d PoW bits, FRI fold %d, <= %d remainder coefficients"
                                          % (opt.num_queries, opt.grinding, opt.fold, opt.max_remainder),
                         "claim": "this library's own instantiation (Blake2s trees over the rows' bytes, Keccak coin, Fq3 columns as three "
                                  "coordinate columns): the reference's parts for this claim are un-vendored - PARITY UNPINNED",
                         "host": ("C++ host (host/goldilocks_prover.cpp, ssh_gl_prove) over the C ABI; the layout's composition is lowered in Python per proof"
                                  if args.gl_host == "cpp" else "Python host (sandstorm_amd/goldilocks.py) over the C ABI"), "fri_layers": len(proof.fri_layers),
                         "outside": "host trace generation: base columns resident in HBM",
                         "per_gpu": "one independent proof per rank"},
              "stage_ms_per_proof": {k: round(v[0] / args.steps, 3) for k, v in prof.items()},
              "roofline": (lambda algo: {"bound": "hbm", "kernel": "ss::gl_ntt_pass_kernel", "achieved": algo / (ntt_ms * 1e-3 / args.steps) / 1e9,
                                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": algo / (ntt_ms * 1e-3 / args.steps) / 1e9 / HBM_PEAK_GBPS,
                                         "traffic": traffic.get("bytes_per_launch"), "traffic_bytes_per_proof": traffic.get("bytes_per_proof"),
                                         "traffic_source": traffic.get("source"), "launches": launches, "avg_launch_ms": ntt_ms / max(1, launches),
                                         "algorithmic_bytes_per_proof": algo,
                                         "note": "algorithmic bytes = 2 N 8 B per transform (SURVEY 8d) over the proof's transforms: LDE of 8 columns "
                                                 "(n and 2n), composition (3 of 2n in, 6 of 2n out), out-of-domain (3 per shifted column, n), DEEP "
                                                 "extension (3 of n, 3 of 2n); measured traffic is 3.6x that: three passes per transform plus twiddles. "
                                                 "The kernel is ALU-bound (a butterfly is ~38 vector instructions, five of them quarter-rate; "
                                                 "profiles/r02_end2_sq_counters_*): see --workload goldilocks_lde_2p20"})(16.0 * (n * (8 + 24 + 3) + 2 * n * (8 + 9 + 3)))})
    if world > 1:
        dist.destroy_process_group()


def bench_sharded(args, layout, log_steps, rank, local_rank, world, device):
    """--gpus N > 1: ONE proof over the N GPUs of the node by the C++ host's sharded prover (sandstorm_amd/host/sharded.cpp): trace
    columns extended on their owner (column c on rank c % N), point-to-point re-shards into row blocks over RCCL, row hashing /
    constraint evaluation / DEEP on row blocks, leaf-block sub-trees + root all-gather, single-vector transforms and the large FRI
    layers spread over the ranks.  A step = one whole proof; the time is the max over ranks between two
    barriers; strong scaling."""
    from sandstorm_amd import backend as be, hostlib
    from sandstorm_amd.prover import ProofOptions
    L, pi = _sample_statement(layout, log_steps)
    n = 16 << log_steps
    # ONE stream for torch's tensor ops, the collectives and the C ABI's kernels (torch's default stream has handle 0, which
    # ss_ctx_set_stream reads as "the context's own stream")
    stream = torch.cuda.Stream(device)
    torch.cuda.set_stream(stream)
    ctx = be.Context(local_rank, stream=stream.cuda_stream)
    # the C++ host's AIR (tables on the device, program lowered in C++ per proof)
    host_air = (hostlib.StarknetHostAir if layout == "starknet" else hostlib.RecursiveHostAir)(ctx, pi, log_steps + 4, 1)
    air = hostlib.prover_air(host_air)
    nb, ne = air.num_base_columns, air.num_extension_columns
    coin = be.COIN_CAIRO if layout == "recursive" else be.COIN_SOLIDITY
    mine = {c: synth_columns(device, 1, log_steps + 4, seed=0x53414E44 + c)[0] for c in range(nb) if c % world == rank}
    # the extension trace as ROW BLOCKS: every rank holds its n / N rows of the auxiliary columns and runs the layout's scans on them;
    # one all-gather of the blocks' totals (7 field elements per rank) and the blocks before a block are folded in
    # (host/extension.cpp build_extension_blocks) - no rank makes or scatters a whole extension column
    log_block = log_steps + 4 - (world.bit_length() - 1)
    aux = synth_columns(device, 5 if layout == "recursive" else 3, log_block, seed=0x7E57 + rank)
    ext_keep = []

    def extension_blocks(challenges):
        while ext_keep:
            ext_keep.pop().close()
        m = hostlib.build_extension_blocks(ctx, layout, aux, n, rank, world, group, challenges, check=False)      # synthetic cells: no permutation closes
        ext_keep.append(m)
        return m.cols
    seed = bytes((7 * i) & 0xff for i in range(32))

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()
    comm_check = None
    transport = "a group of one" if world == 1 else "RCCL through ss_comm_*"
    if True:
        # the C++ host's sharded prover (sandstorm_amd/host/sharded.cpp) over RCCL through the C ABI (ss_comm_*): rank 0 makes the
        # communicator's id, torch.distributed only hands it out
        if world == 1:                              # a group of one needs no communicator
            group = hostlib.LocalGroup(1)
        else:
            # ONE communicator for the whole run, made before the warm-up (an RCCL unique id serves one ncclCommInitRank per
            # rank; its set-up is not part of a proof).  A watchdog turns a bootstrap that never completes into an error.
            import threading
            uid, uid_err = None, None
            if rank == 0:
                try:
                    uid = hostlib.rccl_unique_id()
                except Exception as e:              # noqa: BLE001 - librccl not loadable / ncclGetUniqueId refused: every rank must learn it, none may wait for an id
                    uid_err = "%s: %s" % (type(e).__name__, e)
            box = [(uid, uid_err)]
            dist.broadcast_object_list(box, src=0)
            uid, uid_err = box[0]
            done = threading.Event()

            def watchdog():
                if not done.wait(float(os.environ.get("SS_BENCH_RCCL_TIMEOUT_S", "300"))):
                    sys.stderr.write("bench.py: rank %d: the RCCL communicator of the C++ host did not come up or hung in its self check (ss_comm_*)\n" % rank)
                    sys.stderr.flush()
                    os._exit(3)
            threading.Thread(target=watchdog, daemon=True).start()
            # RCCL between more than one GPU has never run where this was developed (one GPU per box): if the communicator cannot be
            # made on ANY rank (library not found, ncclCommInitRank refused) or fails its self check, every rank runs the SAME driver
            # over the caller's own collectives instead (ssh_callback_group_create: torch.distributed's gloo group, device memory
            # staged through the host - how the CPU suite runs it as 2 / 4 / 8 processes): same proof bytes, same kernels, slower
            # exchanges - and the JSON line says so
            group, group_err = None, uid_err
            if uid is not None:
                try:
                    group = hostlib.RcclGroup(ctx, uid, rank, world)
                except Exception as e:              # noqa: BLE001 - reported below, by every rank the same way
                    group_err = "%s: %s" % (type(e).__name__, e)
            bad = torch.tensor([0 if group is not None else 1], dtype=torch.int32, device=device)
            dist.all_reduce(bad)
            if not int(bad.item()):
                # the group's first exchanges are a SELF CHECK, not a proof's (host/sharded.cpp transport_self_check): two messages of
                # different sizes between every ordered pair of ranks whose bytes name (source, destination, message), an all-gather, a
                # variable-length all-gather - every rank learns every rank's verdict -, then one timed all-to-all of 64 MiB per pair
                try:
                    comm_check = {"ok": True, "exchange_gbps": hostlib.group_self_check(ctx, rank, world, group, 64 << 20)}
                except Exception as e:              # noqa: BLE001
                    group_err = "%s: %s" % (type(e).__name__, e)
                    comm_check = {"ok": False, "error": group_err}
                bad = torch.tensor([0 if comm_check["ok"] else 1], dtype=torch.int32, device=device)
                dist.all_reduce(bad)
                rates = [None] * world
                dist.all_gather_object(rates, comm_check.get("exchange_gbps"))
                comm_check["exchange_gbps_by_rank"] = rates
                comm_check["note"] = ("ss_comm_exchange + ss_comm_all_gather over the run's RCCL group before the warm-up: contents checked on every "
                                      "rank; exchange_gbps = a rank's send + receive rate in one all-to-all of 64 MiB per ordered pair")
            done.set()
            if int(bad.item()):
                if rank == 0 or group_err:
                    sys.stderr.write("bench.py: rank %d: the C++ host's RCCL group did not come up or failed its self check on %d rank(s) (%s): "
                                     "the same driver over torch.distributed's gloo group (host-staged exchanges)\n" % (rank, int(bad.item()), group_err))
                args.transport_fallback = group_err or "another rank failed"
                if group is not None:
                    group.close()
                group = hostlib.torch_dist_group(dist.new_group(backend="gloo"))
                transport = "the caller's collectives (gloo, host-staged): the RCCL group failed"
                comm_check = dict(comm_check or {}, fallback_ok=True, fallback_exchange_gbps=hostlib.group_self_check(ctx, rank, world, group, 1 << 20))
    tree_kind, n_friendly = (be.TREE_FRIENDLY, 22) if layout == "recursive" else (be.TREE_KECCAK_M20, 0)
    wire_proof = [None]

    def prove_once():
        wire_proof[0] = hostlib.prove_sharded(ctx, host_air, tree_kind, n_friendly, coin, seed, rank, world, group, mine, log_steps + 4,
                                              None, ProofOptions(), extension_blocks=extension_blocks)
        return wire_proof[0]
    proof = prove_once()
    if rank == 0:
        from sandstorm_amd import wire as wire_format
        proof = wire_format.parse(proof, tree_kind)

    def all_max(dt):
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item())
    for _ in range(args.warmup):
        prove_once()
    barrier()
    ctx.profile(True)
    ctx.profile_reset()
    host_profile = None
    if os.environ.get("SS_BENCH_PROFILE") == "1" and rank == 0:        # diagnosis: where the driver's HOST time goes (-> stderr)
        import cProfile
        host_profile = cProfile.Profile()
        host_profile.enable()
    sec = timed_steps(prove_once, args.steps, 0, barrier, all_max)
    if host_profile is not None:
        import pstats
        host_profile.disable()
        pstats.Stats(host_profile, stream=sys.stderr).sort_stats("tottime").print_stats(45)
    kinds = [("ntt_pass", be.PROF_NTT_PASS), ("hash_rows", be.PROF_HASH_ROWS), ("merkle", be.PROF_MERKLE), ("fri_fold", be.PROF_FRI),
             ("quotient", be.PROF_QUOTIENT), ("deep", be.PROF_DEEP), ("extension_scans", be.PROF_EXT)]
    prof = {name: ctx.profile_read(k) for name, k in kinds}
    ctx.profile(False)
    # every rank's kernel time per stage (HIP events on its own stream), gathered for the report: where the proof's time goes per GPU
    mine_ms = {k: round(v[0] / args.steps, 3) for k, v in prof.items()}
    by_rank = [None] * world
    dist.all_gather_object(by_rank, (mine_ms, prof["ntt_pass"][1]))
    if rank == 0:
        algo = stage_algorithmic_bytes(nb, ne, log_steps + 4, 1, len(proof.fri_layers))["ntt_pass"]
        slowest = max(range(world), key=lambda r: by_rank[r][0]["ntt_pass"])
        ntt_ms, ntt_launches = by_rank[slowest][0]["ntt_pass"], by_rank[slowest][1]
        ach = algo / (ntt_ms * 1e-3) / 1e9 if ntt_ms > 0 else 0.0
        emit({
            "metric": "prove_wall_time_s", "value": sec, "unit": "s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": sec * 1e3, "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
            "dtype": "u256 (Fp252, Montgomery R=2^256, 8 x u32 limbs)", "data": "synthetic", "proofs_per_s": 1.0 / sec,
            "stage_ms_per_proof_by_rank": [r[0] for r in by_rank],
            "roofline": {"bound": "hbm", "kernel": "ss::ntt_pass_kernel", "stage": "ntt_pass", "achieved": ach, "peak": HBM_PEAK_GBPS * world,
                         "unit": "GB/s", "frac": ach / (HBM_PEAK_GBPS * world), "traffic": None,
                         "launches": ntt_launches, "avg_launch_ms": ntt_ms * args.steps / max(1, ntt_launches),
                         "algorithmic_bytes_per_proof": algo,
                         "note": "the WHOLE proof's transform bytes (SURVEY 8d: 2 N 32 B per transform) over the transform time of the rank "
                                 "that spends most in them (rank %d: columns are dealt round-robin, 9-10 columns do not divide by %d), "
                                 "against %d x the HBM peak; per-launch traffic is the single-device run's (profiles/hbm_traffic_*.json)"
                                 % (slowest, world, world)},
            "config": {"workload": args.workload, "layout_shape": layout, "steps_log2": log_steps, "trace_rows_log2": log_steps + 4,
                       "columns": "%d base + %d extension" % (nb, ne),
                       "parallelism": ("ONE proof sharded over %d GPUs: base-column LDE by column (column c on rank c %% %d), row hashing / "
                                       "constraint evaluation / DEEP by row block (point-to-point re-shard over RCCL, wrap-around halo of %d "
                                       "rows), leaf-block sub-trees + root all-gather, " % (world, world, max(o for _, o in air.mask) << 1))
                                      + ("the extension trace's scans by row block (one all-gather of the blocks' totals), "
                                         "extension columns / composition interpolation + extension / DEEP extension each ONE transform over "
                                         "the ranks (two equal-split all-to-alls per transform), FRI layers above 2^21 values folded and "
                                         "committed by all ranks, the rest on rank 0"),
                       "air": "the REAL %s AIR (%d mask cells) on synthetic columns" % (layout, len(air.mask)),
                       "claim": "CairoVerifierClaim (Blake2s+Pedersen-22 tree, Cairo coin)" if layout == "recursive"
                                else "EthVerifierClaim (Keccak-masked-20 tree, Solidity coin)",
                       "proof_options": "65 queries, blowup 2, 16 PoW bits, FRI fold 8, <=16 remainder coeffs",
                       "host": "C++ host (sandstorm_amd/host/sharded.cpp) over the C ABI, " + transport,
                       "fri_layers": len(proof.fri_layers) if proof is not None else None,
                       "transport_fallback": getattr(args, "transport_fallback", None),
                       "comm_self_check": comm_check,
                       "note": "python bench.py --gpus N --mode replicas runs N independent proofs instead (weak scaling)"},
        })
    dist.destroy_process_group()


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` as typed (no launcher around it): start the N ranks - one process per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1 - wait for them and pass rank 0's ONE JSON line through (every rank's
    stdout is this process's; only rank 0 writes to it).  -> the exit code."""
    import subprocess
    selftest = os.environ.get("SS_BENCH_SELFTEST") == "1"
    if not selftest:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            sys.stderr.write("bench.py --gpus %d: this node has %d GPU(s); one rank per GPU is the only mode\n" % (args.gpus, have))
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # the host driver only supports dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    return subprocess.call(cmd, env=env)


def timed_steps(step, steps, warmup, barrier, all_max):
    """the timing protocol of the contract: `warmup` untimed steps, a barrier + device sync, EXACTLY `steps` steps, a barrier
    + device sync, the MAX over ranks.  -> seconds per step"""
    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    return all_max(time.perf_counter() - t0) / steps


def ood_by_transform(log_n):
    """csrc/capi.hip ss_ood_eval: one coset transform per column below 2^20 coefficients (or SS_OOD_TRANSFORM=1), point by point above"""
    return os.environ.get("SS_OOD_TRANSFORM") == "1" or log_n < int(os.environ.get("SS_OOD_SPARSE_MIN_LOG", "20"))


def stage_algorithmic_bytes(nb, ne, log_n, lb, fri_layers, fold=8):
    """SURVEY.md 8d's algorithmic HBM bytes per proof and stage (every datum read once, every result written once)"""
    n, N, C = 1 << log_n, 1 << (log_n + lb), nb + ne
    by_transform = ood_by_transform(log_n)
    out = {"quotient": 32.0 * (C + 1) * N,                                   # every trace cell once + the composition evaluations
           # composed on the n-point sub-coset; the point-by-point out-of-domain kernels read every coefficient once
           "deep": 32.0 * ((C + 2) * n + n + (0 if by_transform else (C + 2) * n)),
           # 2 N e per transform: trace LDE, composition (one inverse, two forward), the DEEP polynomial's extension, OOD if by transform
           "ntt_pass": 32.0 * (C * (2 * n + 2 * N) + 3 * 2 * N + (2 * n + 2 * N) + (C * 2 * n if by_transform else 0))}
    hashed = [(nb, N)] + ([(ne, N)] if ne > 1 else []) + [(2, N)]
    trees = [N, N, N] if ne else [N, N]
    fri = 0.0
    L = N
    for _ in range(fri_layers):
        hashed.append((fold, L // fold))
        trees.append(L // fold)
        fri += 32.0 * (L + L // fold)
        L //= fold
    out["hash_rows"] = float(sum(rows * (c * 32 + 32) for c, rows in hashed))
    out["merkle"] = float(sum(3 * 32 * (leaves - 1) for leaves in trees))
    out["fri_fold"] = fri
    return out


def _side_leg(name, fn, *args):
    """a leg of the report that is not the timed region (the CPU baseline, files -> proof): its result, or what went wrong with it"""
    try:
        return fn(*args)
    except Exception as e:              # noqa: BLE001 - reported in the line, the traceback on stderr
        import traceback
        traceback.print_exc()
        sys.stderr.write("bench.py: the %s leg failed: the line goes on without it\n" % name)
        return {"error": "%s: %s" % (type(e).__name__, e)}


def end_to_end(ctx, layout, log_steps, device, repeats=5):
    """`files -> proof`: what the reference's "Proof generated in" timer wraps - claim.prove(options, witness) (cli/src/main.rs:200-202)
    INCLUDES generate_trace (src/lib.rs:94-100).  A REAL statement of the layout at this step count (the reference's example run
    re-declared and padded with its final state: sandstorm_amd/examples.py; the same statements tests/test_gpu_full_size.py and
    tests/test_gpu_recursive_claim.py prove and verify); the raw `cairo-run` bytes are the input.
      total_s      ONE call from the files to the proof with the base trace made ON the device (ssh_prove_files_device: trace.bin /
                   memory.bin uploaded as they are, csrc/trace.hip makes the 7 / 9 columns in HBM, then the proof) - round 6
      trace_gen_s  inside that call: until the columns were final (upload of the files, plans, kernels, the status read)
      prove_s      the proof alone on resident columns: every stage of bench.py's timed region plus the REAL extension columns (check on)
      host_generated  the round-5 path beside it: the C++ host's ExecutionTrace::new (OpenMP) into PINNED host columns, every column
                   uploaded the moment it is final, the prover extending them as they land (ssh_prove_files), and the same three
                   steps one after the other (`serial`)
    -> the means over `repeats` runs after one untimed run (plans and tables built)."""
    from sandstorm_amd import backend as be, binary, examples, hostlib, public_input
    from sandstorm_amd.prover import ProofOptions
    log_n = log_steps + 4
    n = 1 << log_n
    if layout == "starknet":
        from sandstorm_amd.layouts import starknet as sk
        states, memory, xpi = examples.starknet_example(log_steps)
        gen, nb = hostlib.starknet_base_trace, 9
        aux_idx = (sk.COL_NPC, sk.COL_MEMORY, sk.COL_RANGE_CHECK)
        tree_kind, n_friendly, coin_kind = be.TREE_KECCAK_M20, 0, be.COIN_SOLIDITY
        air = hostlib.StarknetHostAir(ctx, xpi, log_n, 1)
    else:
        from sandstorm_amd.layouts import recursive as rec
        states, memory, xpi = examples.recursive_example(log_steps)
        gen, nb = hostlib.recursive_base_trace, 7
        aux_idx = (rec.COL_NPC, rec.COL_MEMORY, rec.COL_RANGE_CHECK, rec.COL_DILUTED_UNORDERED, rec.COL_DILUTED_ORDERED)
        tree_kind, n_friendly, coin_kind = be.TREE_FRIENDLY, 22, be.COIN_CAIRO
        air = hostlib.RecursiveHostAir(ctx, xpi, log_n, 1)
    trace_bin, memory_bin = binary.write_register_states(states), binary.write_memory(memory)
    del states, memory
    seed = public_input.public_coin_seed(xpi, coin_kind)
    dev = [torch.empty((n, 4), dtype=torch.int64, device=device) for _ in range(nb)]
    keep = []

    def build_extension(challenges):
        del keep[:]
        keep.append(hostlib.build_extension_columns(ctx, layout, [dev[c] for c in aux_idx], n, challenges))
        return keep[0].cols
    options = ProofOptions()
    # ---- the base trace made on the device: ONE call from the files to the proof; then the proof alone on the columns it left
    total, inside, prove = 0.0, 0.0, 0.0
    for it in range(repeats + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, tm = hostlib.prove_files_device(ctx, layout, trace_bin, memory_bin, xpi, None, dev, air, tree_kind, n_friendly, coin_kind, seed, build_extension, options,
                                           want_proof=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        hostlib.prove(ctx, air, tree_kind, n_friendly, coin_kind, seed, dev, log_n, build_extension, options, want_proof=False)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if it:
            total += t1 - t0
            inside += tm["trace_gen_s"]
            prove += t2 - t1
    total, inside, prove = total / repeats, inside / repeats, prove / repeats
    out = {"total_s": total, "generator": "device", "trace_gen_s": inside, "prove_s": prove, "total_over_prove": total / prove if prove > 0 else None,
           "input_bytes": len(trace_bin) + len(memory_bin), "column_bytes": 32 * n * nb}
    # ---- the host-generated path (round 5), fewer repeats: pinned columns, the generator on the host's cores, uploads overlapped
    host_repeats = 2
    pinned = [torch.empty((n, 4), dtype=torch.int64).pin_memory() for _ in range(nb)]
    views = [t.numpy().view("uint64") for t in pinned]
    acc = {"trace_gen_s": 0.0, "h2d_s": 0.0, "prove_s": 0.0}
    for it in range(host_repeats + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gen(trace_bin, memory_bin, xpi, out=views)
        t1 = time.perf_counter()
        for c in range(nb):
            dev[c].copy_(pinned[c], non_blocking=True)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        hostlib.prove(ctx, air, tree_kind, n_friendly, coin_kind, seed, dev, log_n, build_extension, options, want_proof=False)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        if it:
            acc["trace_gen_s"] += t1 - t0
            acc["h2d_s"] += t2 - t1
            acc["prove_s"] += t3 - t2
    serial = {k: v / host_repeats for k, v in acc.items()}
    over = {"total_s": 0.0, "trace_gen_s": 0.0}
    for it in range(host_repeats + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, tm = hostlib.prove_files(ctx, layout, trace_bin, memory_bin, xpi, None, views, dev, air, tree_kind, n_friendly, coin_kind, seed, build_extension,
                                    options, want_proof=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if it:
            over["total_s"] += t1 - t0
            over["trace_gen_s"] += tm["trace_gen_s"]
    over = {k: v / host_repeats for k, v in over.items()}
    for m in keep:
        m.close()
    del keep[:], dev, pinned, views
    air.close()
    out["host_generated"] = {"total_s": over["total_s"], "upload_overlapped": True, "trace_gen_s": over["trace_gen_s"], "h2d_s": serial["h2d_s"],
                             "serial": dict(serial, total_s=sum(serial.values())), "host_threads": int(os.environ.get("OMP_NUM_THREADS", HOST_CPUS)),
                             "host_cpus_visible": os.cpu_count(), "host_cpu_quota": HOST_CPUS}
    out["statement"] = ("the reference's array-sum run re-declared for the %s layout, padded to 2^%d steps (a real, verifiable statement: "
                        "%d base columns x 2^%d rows from %.1f MB of trace.bin / memory.bin).  total_s: ONE call from the files to the proof "
                        "(ssh_prove_files_device): the files' bytes uploaded as they are, the base columns made in HBM by csrc/trace.hip, then "
                        "the proof; trace_gen_s: until the columns were final inside that call; prove_s: the proof alone on resident columns.  "
                        "host_generated: the round-5 path (host generator into pinned columns, uploads overlapped: ssh_prove_files)"
                        % (layout, log_steps, nb, log_n, (len(trace_bin) + len(memory_bin)) / 1e6))
    return out


def bench_proof(args, workload, rank, local_rank, world, device, steps, warmup, cpu_leg):
    """K whole proofs of one workload through the C++ host on this rank's GPU -> the report (rank 0) or None"""
    from sandstorm_amd import backend as be, extension, hostlib, public_input
    from sandstorm_amd.prover import ProofOptions
    layout, log_steps = WORKLOADS[workload]
    log_n, lb = log_steps + 4, 1
    n = 1 << log_n
    ctx = be.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
    real = layout == "recursive-real"
    pi = public_input.AirPublicInput.from_json(os.path.join(ROOT, "tests", "golden", "air_public_input_array_sum.json"))
    if real:
        # the reference's example: raw files -> base trace on the host (C++) -> HBM; the real 93-constraint AIR; the seed
        # from its air-public-input.json; CairoVerifierClaim as the CLI picks for this layout (cli/src/main.rs:95-99)
        layout = "recursive"
        ex = os.path.join(ROOT, "tests", "golden", "example")
        with open(os.path.join(ex, "trace.bin"), "rb") as fh:
            trace_bin = fh.read()
        with open(os.path.join(ex, "memory.bin"), "rb") as fh:
            memory_bin = fh.read()
        t0 = time.perf_counter()
        host_cols = hostlib.recursive_base_trace(trace_bin, memory_bin, pi)
        trace_gen_s = time.perf_counter() - t0
        assert host_cols[0].shape[0] == n
        base_cols = [ctx.column(c) for c in host_cols]
        air = hostlib.RecursiveHostAir(ctx, pi, log_n, lb)
        tree_kind, n_friendly, coin_kind = be.TREE_FRIENDLY, 22, be.COIN_CAIRO
        seed = public_input.public_coin_seed(pi, coin_kind)
        aux_cols = [base_cols[3], base_cols[4], base_cols[5], base_cols[1], base_cols[2]]     # npc, memory, range check, diluted x2
        keep = []

        def build_extension(challenges):
            del keep[:]
            keep.append(hostlib.build_extension_columns(ctx, "recursive", aux_cols, n, challenges))      # check=True: real permutations
            return keep[0].cols
    else:
        # the C++ host side (libsandstorm_host.so): coin, Expr lowering, prover, the layout's REAL AIR.  The statement's public
        # input only feeds constants of the program (hints, public memory product): the reference's example run re-declared
        # for this layout and step count
        pi.n_steps = 1 << log_steps
        if layout == "starknet":
            from sandstorm_amd.layouts import starknet as sk
            air = hostlib.StarknetHostAir(ctx, sk.example_public_input(pi), log_n, lb)
            tree_kind, n_friendly, coin_kind = be.TREE_KECCAK_M20, 0, be.COIN_SOLIDITY        # cli/src/main.rs:90-94 -> EthVerifierClaim
        else:
            air = hostlib.RecursiveHostAir(ctx, pi, log_n, lb)
            tree_kind, n_friendly, coin_kind = be.TREE_FRIENDLY, 22, be.COIN_CAIRO           # cli/src/main.rs:95-99 -> CairoVerifierClaim
        # inputs resident in HBM before the timed region; one independent trace per rank (weak scaling)
        base_t = synth_columns(device, air.num_base_columns, log_n, seed=0x53414E44 + rank)
        base_cols = [base_t[c] for c in range(air.num_base_columns)]
        # the trace's auxiliary columns (npc, memory, range check [, diluted unordered / ordered]): the extension columns
        # are built from them ON THE DEVICE inside the timed region, after the challenges are drawn
        # (Trace::build_extension_columns, sandstorm_amd/extension.py; random data, so the is_one asserts are off)
        aux_t = synth_columns(device, 5 if layout == "recursive" else 3, log_n, seed=0x7E57 + rank)
        trace_cols = extension.TraceColumns(aux_t[0], aux_t[1], aux_t[2], n, *(aux_t[3:5] if layout == "recursive" else ()))
        seed = bytes((7 * i + rank) & 0xff for i in range(32))

        def build_extension(challenges):
            m = extension.build_extension_columns(layout, ctx, trace_cols, challenges, check=False)
            assert m.num_cols == air.num_extension_columns
            return m.cols
    options = ProofOptions()        # CLI defaults: 65 queries, blowup 2, 16 PoW bits, fold 8, <=16 remainder

    def step(want_proof=False):
        return hostlib.prove(ctx, air, tree_kind, n_friendly, coin_kind, seed, base_cols, log_n,
                             build_extension, options, want_proof=want_proof)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def all_max(dt):
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    proof = step(want_proof=True)         # untimed: metadata for the report (also builds every plan and table)
    for _ in range(warmup):
        step()
    barrier()
    ctx.profile(True)
    ctx.profile_reset()
    sec_per_proof = timed_steps(step, steps, 0, barrier, all_max)
    kinds = [("ntt_pass", be.PROF_NTT_PASS), ("hash_rows", be.PROF_HASH_ROWS), ("merkle", be.PROF_MERKLE),
             ("fri_fold", be.PROF_FRI), ("quotient", be.PROF_QUOTIENT), ("deep", be.PROF_DEEP),
             ("extension_scans", be.PROF_EXT)]
    prof = {name: ctx.profile_read(k) for name, k in kinds}
    # one more proof, untimed, with shader-clock stamps around every profiled launch (ss_profile_enable level 2: s_memtime /
    # s_memrealtime read by one wave per XCD before and after each scope): the clock each stage's kernels are granted IN THIS RUN
    stage_clock = {}
    if not args.no_stage_clocks:
        try:
            ctx.profile(2)
        except Exception as e:          # noqa: BLE001 - no stream for the clock monitor that leaves the measured one alone: the line goes without clocks
            sys.stderr.write("bench.py: stage clocks not measured (%s)\n" % e)
        else:
            ctx.profile_reset()
            step()
            for name, k in kinds:
                cyc, ref = ctx.profile_read_clock(k)
                ms2 = ctx.profile_read(k)[0]
                if ref > 0 and ms2 > 0:
                    stage_clock[name] = {"ghz": cyc / ref * 0.1, "ref_ticks_per_s": ref / (ms2 * 1e-3), "stamped_ms": ms2}
    ctx.profile(False)
    out = None
    if rank == 0:
        nb, ne = air.num_base_columns, air.num_extension_columns
        ncols = nb + ne
        N = n << lb
        # NTT work inside one proof: trace LDE (iNTT n + NTT N per column), composition (iNTT N, 2 x NTT N), the DEEP polynomial's
        # extension (iNTT n + NTT N), OOD where it is one transform per column (NTT n), FRI remainder (tiny)
        ntt_ops = ncols * (ntt_field_ops(log_n) + ntt_field_ops(log_n + lb)) + 3 * ntt_field_ops(log_n + lb) \
            + ntt_field_ops(log_n) + ntt_field_ops(log_n + lb) + (ncols * ntt_field_ops(log_n) if ood_by_transform(log_n) else 0)
        algo = stage_algorithmic_bytes(nb, ne, log_n, lb, len(proof.fri_layers))
        # ... and the pruned transforms of DEEP's rational functions (csrc/capi.hip deep_compose_impl, from 2^20 points on): the columns
        # with >= 12 mask cells, the constants' and the denominator polynomial, 2^poly_log coefficients each onto the n sub-coset
        # points - only poly_log of the log n stages are real, and what they move is the n x 32 B each writes
        pruned = {"polynomials": 0, "real_stages": 0, "butterflies": 0.0}
        if log_n >= int(os.environ.get("SS_DEEP_RATIONAL_MIN_LOG", "20")) and os.environ.get("SS_DEEP_TAPS") != "1":
            try:
                mask = hostlib.prover_air(air).mask
                per_col = {}
                for c, _ in mask:
                    per_col[c] = per_col.get(c, 0) + 1
                min_cells = int(os.environ.get("SS_DEEP_RATIONAL_MIN_CELLS", "12"))
                big = [c for c, k in per_col.items() if k >= min_cells]
                offs = len({o for _, o in mask})
                poly_log = max(1, (offs + 1 - 1).bit_length())
                moved = sum(per_col[c] for c in big) + offs
                if big and poly_log < log_n and moved >= min_cells * (len(big) + 3):
                    pruned = {"polynomials": len(big) + 2, "real_stages": poly_log, "butterflies": (len(big) + 2) * poly_log * (n / 2.0)}
                    ntt_ops += 3.0 * pruned["butterflies"]
                    algo["ntt_pass"] += 32.0 * n * (len(big) + 2)
            except Exception:
                pass
        stage_ms = {k: v[0] / steps for k, v in prof.items()}
        ntt_ms, ntt_launches = prof["ntt_pass"]
        ntt_s = ntt_ms * 1e-3 / steps

        def stage_roofline(name, kernel, note):
            sec = stage_ms[name] * 1e-3
            ach = algo[name] / sec / 1e9 if sec > 0 else 0.0
            # HBM bytes per launch from the PMC passes of this workload (FETCH_SIZE / WRITE_SIZE cannot be read live;
            # tools/profile_round.sh collects them per kernel, the committed summary is cited here)
            traffic, traffic_src, tj = None, None, {}
            tpath = os.path.join(ROOT, "profiles", "hbm_traffic_%s.json" % workload)
            if os.path.exists(tpath):
                with open(tpath) as fh:
                    tj = json.load(fh)
                ent = tj.get("kernels", {}).get(name) or (tj if name == "ntt_pass" else None)
                if ent:
                    traffic, traffic_src = ent.get("bytes_per_launch"), ent.get("source", tj.get("source"))
            launches = prof[name][1]
            # the ALU roofline beside the HBM one (these kernels are bound by vector-instruction issue): the
            # stage's SQ_INSTS_VALU per proof (a deterministic count: the counters' own --pmc pass, profiles/alu_counters_<workload>.json)
            # over THIS run's HIP-event stage time, against 1024 SIMDs issuing one wave-instruction per `weighted_cycles_per_inst`
            # cycles of the clock THIS run's stamps measured for the stage (ss_profile_read_clock) - the kernels' static instruction
            # mix priced with the issue cost of each mnemonic in cycles of the probe's own clock (tools/ubench.hip stamps itself)
            alu = None
            apath = os.path.join(ROOT, "profiles", "alu_counters_%s.json" % workload)
            clk = stage_clock.get(name)
            if os.path.exists(apath) and sec > 0 and clk:
                with open(apath) as fh:
                    aj = json.load(fh)
                st = aj.get("stages", {}).get(name)
                if st and st.get("weighted_cycles_per_inst"):
                    insts, cyc = st["valu_wave_insts_per_proof"], st["weighted_cycles_per_inst"]
                    peak_i = 1024 * clk["ghz"] * 1e9 / cyc
                    units = {"ntt_pass": ntt_ops / 3.0, "quotient": float(N), "deep": float(n), "hash_rows": float(3 * N), "merkle": float(3 * N)}.get(name)
                    alu = {"valu_wave_insts_per_proof": insts, "valu_insts_per_unit": insts * 64.0 / units if units else None,
                           "unit": {"ntt_pass": "butterfly", "quotient": "LDE point", "deep": "sub-coset point", "hash_rows": "row", "merkle": "leaf"}.get(name),
                           "weighted_cycles_per_inst": cyc, "cycles_are": aj.get("cycles_are", "cycles at a nominal 2.4 GHz (round-4 price list)"),
                           "clock_ghz": clk["ghz"], "clock_source": "this run: s_memtime / s_memrealtime stamps around the stage's launches (ss_profile_read_clock)",
                           "ref_counter_hz_measured": clk["ref_ticks_per_s"],
                           "peak_wave_insts_per_s": peak_i, "achieved_wave_insts_per_s": insts / sec,
                           "frac": insts / sec / peak_i, "counters_source": aj.get("source"), "counters_commit": aj.get("commit")}
                    if alu["frac"] > 1.05:
                        # c-bar prices the STATIC mix of the stage's kernels; where most of the executed instructions sit in a few
                        # loop bodies (the Pedersen trees: window loops of curve additions around one-off glue) the executed mix is
                        # cheaper per instruction than the static one and the fraction overshoots 1 - read it as "issue-bound"
                        alu["caveat"] = ("frac > 1: the static instruction mix over-prices this stage's executed mix (loop-heavy kernels); "
                                         "the stage is issue-bound, the fraction is not a precise utilisation")
            return {"bound": "hbm", "kernel": kernel, "stage": name, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBPS, "alu": alu, "traffic": traffic, "traffic_source": traffic_src,
                    "traffic_commit": tj.get("commit") if traffic is not None else None,
                    "algorithmic_bytes_per_proof": algo[name], "algorithmic_bytes_per_launch": algo[name] / max(1.0, launches / steps),
                    "launches": launches, "avg_launch_ms": prof[name][0] / max(1, launches), "stage_ms_per_proof": stage_ms[name], "note": note}
        dominant = max(("quotient", "ntt_pass", "deep", "merkle", "hash_rows"), key=lambda k: stage_ms[k])
        kernel_of = {"quotient": "ss::quotient_%s_kernel" % layout, "ntt_pass": "ss::ntt_pass_kernel", "deep": "ss::deep_kernel",
                     "merkle": "ss::pedersen_*_kernel + blake2s pairs" if layout == "recursive" else "ss::keccak_pairs_kernel",
                     "hash_rows": "ss::blake2s_rows_kernel" if layout == "recursive" else "ss::keccak_rows_kernel"}
        out = {
            "metric": "prove_wall_time_s", "value": sec_per_proof, "unit": "s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": sec_per_proof * 1e3,
            "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
            "dtype": "u256 (Fp252, Montgomery R=2^256, 8 x u32 limbs)", "data": "synthetic",
            "proofs_per_s": world / sec_per_proof,
            "ntt_gfield_ops_per_s": ntt_ops / ntt_s / 1e9 if ntt_s > 0 else 0.0,
            "ntt_gbutterflies_per_s": ntt_ops / 3 / ntt_s / 1e9 if ntt_s > 0 else 0.0,
            "config": {"workload": workload, "layout_shape": layout, "steps_log2": log_steps,
                       "trace_rows_log2": log_n, "columns": "%d base + %d extension" % (nb, ne),
                       "claim": "CairoVerifierClaim (Blake2s+Pedersen-22 tree, Cairo coin)" if layout == "recursive"
                                else "EthVerifierClaim (Keccak-masked-20 tree, Solidity coin)",
                       "air": ("the REAL recursive AIR (93 constraints, %d mask cells: sandstorm_amd/host/air_recursive.cpp) on the reference's "
                               "example run; base trace generated by the C++ host in %.3f s (outside the timed region)" % (air.mask_size, trace_gen_s))
                              if real else
                              ("the REAL %s AIR (%s, %d mask cells: sandstorm_amd/host/air_%s.cpp) on synthetic columns" %
                               (layout, "195 constraints" if layout == "starknet" else "93 constraints", air.mask_size, layout)),
                       "proof_options": "65 queries, blowup 2, 16 PoW bits, FRI fold 8, <=16 remainder coeffs",
                       "in_timed_region": "LDE x2, extension-column scans (A2), commits x3, quotient, composition LDE, OOD, DEEP, FRI, PoW, openings",
                       "outside": "host trace generation (A1): base and auxiliary columns are resident in HBM",
                       "per_gpu": "one independent proof per rank",
                       "host": "C++ prover (sandstorm_amd/host, libsandstorm_host.so) through ctypes",
                       "fri_layers": len(proof.fri_layers), "pow_nonce": proof.pow_nonce},
            "stage_ms_per_proof": {k: round(v, 3) for k, v in stage_ms.items()},
            # every stage's shader clock in THIS run (one stamped proof after the timed ones) and its issue fraction at that clock
            "stage_clock_ghz": {k: round(v["ghz"], 4) for k, v in stage_clock.items()},
            "stage_alu_frac": {k: round(r["alu"]["frac"], 4) for k in stage_ms if k in algo
                               for r in [stage_roofline(k, "", "")] if r.get("alu")},
            "roofline": dict(stage_roofline("ntt_pass", "ss::ntt_pass_kernel",
                                            "the kernel north_star names.  algorithmic bytes = 2*N*32 B per transform (SURVEY 8d), shared by "
                                            "its passes; Fp252 butterflies are integer-ALU bound before HBM bound (DESIGN.md section 3): "
                                            "`gbutterflies_per_s` against the 135 G/s of a bare butterfly loop (tools/mulbench.hip); the pruned "
                                            "transforms of DEEP's rational polynomials are counted with their real stages and the bytes they "
                                            "write (`pruned_deep_transforms`)"),
                             gbutterflies_per_s=ntt_ops / 3 / ntt_s / 1e9 if ntt_s > 0 else 0.0, butterfly_ceiling_g_per_s=135.0,
                             pruned_deep_transforms=pruned),
            "roofline_dominant": stage_roofline(dominant, kernel_of[dominant],
                                                "the stage with the largest share of the proof, same computation as `roofline` (SURVEY 8d bytes / "
                                                "HIP-event time of the stage's launches)"),
        }
        if workload == "recursive_2p7":
            out["config"]["reference_published"] = ("186 ms for the 128-step array-sum proof on the author's machine "
                                                    "(BASELINE.md section 1: other hardware, older CLI, the real AIR)")
        # the legs beside the measurement must not cost the line that carries it: one that fails is reported in its place
        if cpu_leg:
            out["cpu_baseline"] = _side_leg("cpu_baseline", cpu_baseline, layout, log_steps, ctx)
        if world == 1 and not real and not args.no_end_to_end and log_steps >= 17:
            out["end_to_end"] = _side_leg("end_to_end", end_to_end, ctx, layout, log_steps, device)
    # everything that lives in the context's pool goes before the context does
    if real:
        for m in keep:
            m.close()
        del keep[:], aux_cols
    else:
        del base_t, aux_t, trace_cols
    del base_cols, build_extension, step
    air.close()
    ctx.close()
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="starknet_2p20", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-north-star", action="store_true", help="skip the recursive_2p20 leg of the default run")
    ap.add_argument("--no-stage-clocks", action="store_true",
                    help="skip the one extra, untimed proof whose launches are stamped for the stages' shader clocks (the counter passes of "
                         "tools/*.sh count kernels per run: two proofs, not three)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the files -> proof leg (trace generation + upload + proof of a real statement)")
    ap.add_argument("--gl-host", default="cpp", choices=["python", "cpp"],
                    help="--workload goldilocks_plain_*: the host above the C ABI - host/goldilocks_prover.cpp (ssh_gl_prove) or "
                         "sandstorm_amd/goldilocks.py; both write the same proof")
    ap.add_argument("--sharded-host", default="cpp", choices=["python", "cpp"],
                    help="--mode shard: the driver above the C ABI is the C++ host's sharded.cpp (RCCL through ss_comm_*; every single-vector "
                         "transform and the large FRI layers spread over the ranks).  `python` named the round-3 Python driver, retired in "
                         "round 6 (one distribution in the tree): it is refused")
    ap.add_argument("--mode", default="auto", choices=["auto", "shard", "replicas"],
                    help="N > 1 GPUs: shard = ONE proof over the N GPUs (column-sharded LDE, row-block hashing / constraints / DEEP, "
                         "RCCL point-to-point re-shards; strong scaling) - the default; replicas = one independent proof per GPU (weak scaling)")
    args = ap.parse_args()
    if args.sharded_host != "cpp":
        sys.exit("--sharded-host python: the Python sharded driver was retired in round 6; the one distribution is host/sharded.cpp (the default)")
    if args.gpus < 1:
        sys.exit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    _claim_stdout()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py --gpus %d inside a group of %d ranks: launch with --nproc-per-node %d (or run it bare and let it "
                 "launch its own ranks)" % (args.gpus, world, args.gpus))
    if os.environ.get("SS_BENCH_SELFTEST") == "1":
        # tests/test_bench_launcher.py: this file's launcher, rendezvous, timing protocol and one-line contract on gloo ranks,
        # the sharded driver over the CPU oracle (test infrastructure, imported from tests/ only) - never a measurement
        from tests import bench_selftest
        return bench_selftest.run(args, rank, world, timed_steps, emit)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: there is no CPU fallback for the hot path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # one node, one process per GPU: RCCL's bootstrap over the loopback interface, for torch's group as for the C ABI's own
        # communicator (csrc/capi.hip sets the same default before its first ncclCommInitRank: on a box with no other interface the
        # probing took minutes); a deployment that wants another interface sets the variable itself
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        # a rank that fails must not leave the others in a collective for the default ten minutes
        import datetime
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device, timeout=datetime.timedelta(seconds=300))

    layout, log_steps = WORKLOADS[args.workload]
    if layout == "goldilocks":
        return bench_goldilocks(args, log_steps, rank, local_rank, world, device)
    if layout == "goldilocks-plain":
        return bench_goldilocks_plain(args, log_steps, rank, local_rank, world, device)
    if (world > 1 and args.mode == "auto" or args.mode == "shard") and layout in ("starknet", "recursive"):
        if world == 1:                      # --mode shard on one GPU: the sharded driver with a group of one (smoke / profiling)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=device)
        return bench_sharded(args, layout, log_steps, rank, local_rank, world, device)
    out = bench_proof(args, args.workload, rank, local_rank, world, device, args.steps, args.warmup,
                      cpu_leg=not args.no_cpu_baseline and world == 1)
    if rank == 0 and world == 1 and args.workload == "starknet_2p20" and not args.no_north_star:
        # north_star's own target beside BASELINE's metric configuration: recursive layout, 2^20 steps, the CLI's claim for it
        # (cli/src/main.rs:95-99: FriendlyMerkleTree<22> + Cairo coin), the same protocol on fewer proofs, in the same run
        # (a leg beside BASELINE's metric: if it fails, the line that carries the metric still goes out and says so)
        ns = _side_leg("north_star", bench_proof, args, "recursive_2p20", rank, local_rank, world, device, max(10, args.steps), 2,
                       not args.no_cpu_baseline)          # >= 10 timed proofs in a driver-style run (3 were a thin sample)
        if "error" in ns and "value" not in ns:
            out["north_star"] = {"workload": "recursive_2p20", "error": ns["error"]}
        else:
            out["north_star"] = {"workload": "recursive_2p20", "value": ns["value"], "unit": "s", "steps": ns["steps"], "warmup": ns["warmup"],
                                 "claim": ns["config"]["claim"], "air": ns["config"]["air"], "stage_ms_per_proof": ns["stage_ms_per_proof"],
                                 "stage_clock_ghz": ns.get("stage_clock_ghz"), "stage_alu_frac": ns.get("stage_alu_frac"),
                                 "ntt_gfield_ops_per_s": ns["ntt_gfield_ops_per_s"], "roofline": ns["roofline"],
                                 "roofline_dominant": ns["roofline_dominant"], "cpu_baseline": ns.get("cpu_baseline"),
                                 "end_to_end": ns.get("end_to_end"),
                                 "target": ">= 10x the CPU prover's end-to-end time at 1 GPU (BASELINE.json north_star); cpu_baseline here "
                                           "is the oracle port, not the reference binary"}
    if rank == 0:
        emit(out)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
