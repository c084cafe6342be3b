"""Round-2 first GPU call: proves the reference's own starknet-layout run (example/bootloader: 2^17 steps = 2^21 trace
rows, committed under tests/golden/bootloader) end to end with the REAL 195-constraint AIR
(sandstorm_amd/layouts/starknet.py), verifies the proof on the host, and times it.  The layout was written and
validated on the CPU only (tests/test_layout_starknet.py); nothing here has run on a device yet.
Usage (GPU box): python tools/starknet_real_air.py [timed proofs, default 3] -> one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from sandstorm_amd import backend as be, extension, public_input, verifier, wire   # noqa: E402
from sandstorm_amd.coin import keccak256                                            # noqa: E402
from sandstorm_amd.layouts import starknet as sk                                    # noqa: E402
from sandstorm_amd.prover import Claim, ProofOptions, Prover                        # noqa: E402
from test_layout_starknet import bootloader_run                                     # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
states, memory, pi, private = bootloader_run()
t0 = time.perf_counter()
cols = sk.base_trace(states, memory, pi, private)
t_trace = time.perf_counter() - t0
n = len(cols[0])


def limbs(col):                                   # canonical ints -> Montgomery limbs on the host (R = 2^256)
    out = np.empty((len(col), 4), dtype=np.uint64)
    for i, v in enumerate(col):
        m = (v << 256) % sk.P
        out[i] = (m & 0xFFFFFFFFFFFFFFFF, (m >> 64) & 0xFFFFFFFFFFFFFFFF, (m >> 128) & 0xFFFFFFFFFFFFFFFF, m >> 192)
    return out


ctx = be.Context(0, stream=torch.cuda.current_stream().cuda_stream)
base = be.Matrix.from_host(ctx, [limbs(c) for c in cols])
t0 = time.perf_counter()
air = sk.make_air(ctx, pi, n)
t_tables = time.perf_counter() - t0
assert len(air.mask) == 269
trace_cols = sk.trace_columns(ctx, base.cols, n)
claim = Claim(air, be.LeafVariantMerkleTree, be.COIN_SOLIDITY)        # starknet::EthVerifierClaim: masked Keccak, Solidity coin
seed = public_input.public_coin_seed(pi, be.COIN_SOLIDITY)
prover = Prover(ctx, claim, ProofOptions())
build = lambda ch: extension.build_extension_columns("starknet", ctx, trace_cols, ch)      # check=True: the products must close
proof = prover.prove(seed, base, build)
torch.cuda.synchronize()


def masked_keccak_leaf_hash(vals):                # the row digest the wire format carries: Keccak of the Montgomery big-endian bytes, 20 bytes kept
    return keccak256(b"".join((v * wire._R % sk.P).to_bytes(32, "big") for v in vals))[:20] + bytes(12)


raw = wire.serialize(wire.from_proof(proof, masked_keccak_leaf_hash))
positions = verifier.verify(raw, sk.verifier_air(pi), be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed)
assert positions == proof.query_positions
t0 = time.perf_counter()
for _ in range(steps):
    prover.prove(seed, base, build)
torch.cuda.synchronize()
print(json.dumps({"workload": "example/bootloader, starknet layout, 2^17 steps, real AIR (195 constraints, 269 mask cells)",
                  "trace_rows_log2": n.bit_length() - 1, "prove_wall_time_s": round((time.perf_counter() - t0) / steps, 4), "steps": steps,
                  "verified_on_host": True, "proof_bytes": len(raw), "fri_layers": len(proof.fri_layers),
                  "base_trace_generation_s_python": round(t_trace, 2), "tables_s": round(t_tables, 2)}))
