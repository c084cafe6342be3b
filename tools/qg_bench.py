#!/usr/bin/env python3
"""GPU box: the compiled constraint kernels' variants side by side (library built with `make QG_AB=1`).

    python tools/qg_bench.py <layout> [log2 steps = 20] [variants ...]  ->  one JSON line

The layout's REAL composition program (C++ lowering) over random columns of 2^(steps+5) LDE points; every variant is held
bit for bit to variant 1 (round 2's single kernel) and, at the first size, to the interpreter; times are HIP-event times of
the launches (ss_ctx_profile).  The variant with the smallest time is named so that a script can pick it up."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sandstorm_amd import backend as be, hostlib, public_input      # noqa: E402
from bench import synth_columns                                     # noqa: E402


def main():
    layout = sys.argv[1]
    log_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    variants = [int(v) for v in sys.argv[3:]] or list(range(3 if layout == "starknet" else 2))
    log_n, lb = log_steps + 4, 1
    n, N = 1 << log_n, 2 << log_n
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx = be.Context(0, stream=stream.cuda_stream)
    pi = public_input.AirPublicInput.from_json(os.path.join(ROOT, "tests", "golden", "air_public_input_array_sum.json"))
    pi.n_steps = 1 << log_steps
    if layout == "starknet":
        from sandstorm_amd.layouts import starknet as sk
        host_air = hostlib.StarknetHostAir(ctx, sk.example_public_input(pi), log_n, lb)
    else:
        host_air = hostlib.RecursiveHostAir(ctx, pi, log_n, lb)
    air = hostlib.prover_air(host_air)
    cols_t = synth_columns(dev, air.num_base_columns + air.num_extension_columns, log_n + lb, seed=11)
    cols = [cols_t[c] for c in range(cols_t.shape[0])]
    ch = [be.felt(pow(7, 11 + 3 * i, be.P)) for i in range(air.num_challenges)]
    program, tables, desc = air.build_program(n, ch, be.felt(pow(5, 77, be.P)))
    g = be.felt(3)
    out = torch.zeros((N, 4), dtype=torch.int64, device=dev)
    ref = None

    def run(env):
        for k in ("SS_QG_VARIANT", "SS_QUOTIENT_INTERPRET"):
            os.environ.pop(k, None)
        os.environ.update(env)
        out.zero_()
        ctx.eval_quotient(program, tables, desc, cols, log_n, lb, g, out)           # warm: code object load
        ctx.sync()
        ctx.profile(True)
        ctx.profile_reset()
        reps = 3
        for _ in range(reps):
            ctx.eval_quotient(program, tables, desc, cols, log_n, lb, g, out)
        ctx.sync()
        ms, launches = ctx.profile_read(be.PROF_QUOTIENT)
        ctx.profile(False)
        return ms / reps, launches // reps, out.clone()
    res = {}
    t_int, _, ref = run({"SS_QUOTIENT_INTERPRET": "1"})
    res["interpreter"] = round(t_int, 2)
    for v in variants:
        try:
            t, launches, got = run({"SS_QG_VARIANT": str(v)})
        except Exception as e:               # a variant that is not in this build
            res["v%d" % v] = "unavailable: %s" % str(e)[:80]
            continue
        same = bool(torch.equal(got, ref))
        res["v%d" % v] = {"ms": round(t, 2), "launches": launches, "equals_interpreter": same}
        if not same:
            print("MISMATCH variant %d" % v, file=sys.stderr)
    ok = {k: v["ms"] for k, v in res.items() if isinstance(v, dict) and v["equals_interpreter"]}
    best = min(ok, key=ok.get) if ok else None
    print(json.dumps({"layout": layout, "log_steps": log_steps, "points": N, "ms": res, "best": best, "best_variant": int(best[1:]) if best else None}))


if __name__ == "__main__":
    main()
