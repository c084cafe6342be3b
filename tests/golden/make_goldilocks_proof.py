"""GPU box: the fixture tests/golden/goldilocks_plain_proof.npz - a proof, made on the MI355X, of the example run of
sandstorm_amd/layouts/plain.py (64 steps of example_program(10)) under sandstorm_amd/goldilocks.py, with 20 queries and 8
grinding bits.  tests/test_goldilocks_stark.py verifies it on the CPU.  Run:  python tests/golden/make_goldilocks_proof.py <out.npz>"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sandstorm_amd import backend as be, goldilocks as gs          # noqa: E402
from sandstorm_amd.layouts import plain as pl                      # noqa: E402

prog = pl.example_program(10)
states, memory = pl.run(prog, 64)
pi = pl.public_input_of(prog, states, memory)
cols = pl.base_trace(states, memory, pi)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
ctx = be.Context(0, stream=stream.cuda_stream)
base = [torch.from_numpy(np.array(c, dtype=np.uint64).view(np.int64)).to(dev) for c in cols]
air, opt = gs.plain_air(), gs.Options(num_queries=20, grinding=8)
proof = gs.Prover(ctx, air, opt).prove(bytes(range(32)), base, lambda ch: gs.plain_extension_on_device(ctx, base, ch)[0], statement=pi)
gs.verify(proof, air, bytes(range(32)), statement=pi, expected_options=opt, required_security_bits=28)
np.savez_compressed(sys.argv[1], **gs.proof_to_arrays(proof))
print("written", sys.argv[1])
