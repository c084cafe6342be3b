"""Run ONLY against the host build of the device code (tests/test_device_code_on_host.py passes this file to pytest explicitly): the
prover of sandstorm_amd/goldilocks.py over the C ABI with torch CPU tensors as the "device" buffers - every kernel of the 64-bit
field's proof executed lane by lane on the CPU - writes the proof the MI355X wrote (tests/golden/goldilocks_plain_proof.npz),
array for array, and the verifier accepts it."""
import os

import numpy as np
import pytest

from sandstorm_amd.layouts import plain as pl

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.mark.gpu
def test_device_code_reproduces_the_gpu_made_proof(oracle):
    import torch
    from sandstorm_amd import goldilocks as gs
    from sandstorm_amd.backend import Context
    prog = pl.example_program(10)
    states, memory = pl.run(prog, 64)
    pi = pl.public_input_of(prog, states, memory)
    cols = pl.base_trace(states, memory, pi)
    ctx = Context(0)
    try:
        base = [torch.from_numpy(np.array(c, dtype=np.uint64).view(np.int64)) for c in cols]
        air, opt = gs.plain_air(), gs.Options(num_queries=20, grinding=8)
        seed = bytes(range(32))
        proof = gs.Prover(ctx, air, opt).prove(seed, base, lambda ch: gs.plain_extension_on_device(ctx, base, ch)[0], statement=pi)
        got = gs.proof_to_arrays(proof)
        with np.load(os.path.join(ROOT, "tests", "golden", "goldilocks_plain_proof.npz")) as f:
            want = {k: f[k] for k in f.files}
        assert set(got) == set(want)
        for k in sorted(want):
            assert np.array_equal(np.asarray(got[k]), want[k]), k
        gs.verify(proof, air, seed, statement=pi, expected_options=opt, required_security_bits=28)
    finally:
        ctx.close()
