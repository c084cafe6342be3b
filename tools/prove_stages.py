#!/usr/bin/env python3
"""Wall-clock per pipeline stage of one proof (synchronising after every stage): where the
time goes that the kernel profile does not show (host lowering, plan builds, copies)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from sandstorm_amd import backend as be, synthetic_air  # noqa: E402
from sandstorm_amd.prover import Claim, ProofOptions, Prover  # noqa: E402
from bench import WORKLOADS, synth_columns  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "starknet_2p20"
    layout, log_steps = WORKLOADS[workload]
    log_n = log_steps + 4
    n = 1 << log_n
    device = torch.device("cuda", 0)
    ctx = be.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    air = synthetic_air.make_air(layout, ctx, log_n, 1)
    claim = Claim(air, be.FriendlyMerkleTree, be.COIN_CAIRO) if layout == "recursive" else Claim(air, be.LeafVariantMerkleTree, be.COIN_SOLIDITY)
    prover = Prover(ctx, claim, ProofOptions())
    base_t = synth_columns(device, air.num_base_columns, log_n, 1)
    ext_t = synth_columns(device, air.num_extension_columns, log_n, 2)
    base = be.Matrix(ctx, [base_t[c] for c in range(air.num_base_columns)], n)
    ext = be.Matrix(ctx, [ext_t[c] for c in range(air.num_extension_columns)], n)
    seed = bytes(range(32))
    prover.prove(seed, base, lambda ch: ext)          # warm-up
    prover.timings = {"enabled": True}
    reps = 2
    for _ in range(reps):
        prover.prove(seed, base, lambda ch: ext)
    t = {k: round(v / reps * 1e3, 2) for k, v in prover.timings.items() if k != "enabled"}
    t["total"] = round(sum(t.values()), 2)
    print(json.dumps({"workload": workload, "stage_wall_ms": t}))


if __name__ == "__main__":
    main()
