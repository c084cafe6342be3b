"""Times the end-to-end proof of the reference's own example (BASELINE.json configs[0]: example/array-sum.cairo,
recursive layout, 2^14 steps = 2^18 trace rows) with the REAL 93-constraint AIR (sandstorm_amd/layouts/recursive.py):
base trace resident in HBM -> LDE, commits, extension columns on the device, quotient, DEEP, FRI, PoW, openings.
Python host (the layout is not mirrored in C++ yet), CLI-default proof options.  One JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sandstorm_amd import backend as be, binary, extension, public_input  # noqa: E402
from sandstorm_amd.layouts import recursive as rec                          # noqa: E402
from sandstorm_amd.prover import Claim, ProofOptions, Prover                # noqa: E402

EX = os.path.join(ROOT, "tests", "golden", "example")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
states = binary.read_register_states(open(os.path.join(EX, "trace.bin"), "rb").read())
memory = binary.read_memory(open(os.path.join(EX, "memory.bin"), "rb").read())
pi = public_input.AirPublicInput.from_json(os.path.join(ROOT, "tests", "golden", "air_public_input_array_sum.json"))
t0 = time.perf_counter()
cols = rec.base_trace(states, memory, pi)
t_trace = time.perf_counter() - t0
n = len(cols[0])
ctx = be.Context(0, stream=torch.cuda.current_stream().cuda_stream)
base = be.Matrix.from_host(ctx, [np.stack([be.felt(v) for v in c]) for c in cols])
air = rec.make_air(ctx, pi, n)
trace_cols = rec.trace_columns(ctx, base.cols, n)
out = {}
for name, tree, coin in (("EthVerifierClaim (Keccak tree, Solidity coin)", be.LeafVariantMerkleTreeUnmasked, be.COIN_SOLIDITY),
                         ("CairoVerifierClaim (Blake2s + Pedersen tree, Cairo coin)", be.FriendlyMerkleTree, be.COIN_CAIRO)):
    claim = Claim(air, tree, coin)
    seed = public_input.public_coin_seed(pi, coin)
    prover = Prover(ctx, claim, ProofOptions())
    prover.prove(seed, base, lambda ch: extension.build_extension_columns("recursive", ctx, trace_cols, ch))      # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        proof = prover.prove(seed, base, lambda ch: extension.build_extension_columns("recursive", ctx, trace_cols, ch))
    torch.cuda.synchronize()
    out[name] = round((time.perf_counter() - t0) / steps, 4)
print(json.dumps({"workload": "array-sum example, recursive layout, 2^14 steps, real AIR (93 constraints, 133 mask cells)",
                  "trace_rows_log2": n.bit_length() - 1, "prove_wall_time_s": out, "steps": steps,
                  "host": "python mirror (prover.py); includes lowering the composition per proof",
                  "base_trace_generation_s_python": round(t_trace, 2), "fri_layers": len(proof.fri_layers)}))
