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


@pytest.mark.gpu
def test_sha256_claim_on_the_device_code():
    """Options(hash="sha256") - the parts cli/src/main.rs:119-120 names for the 64-bit field's claim: SHA-256 row digests and trees
    (FIPS 180-4: the verifier recomputes them with hashlib), a coin with SHA-256 inside - proves and verifies; not the default claim's
    proof, and neither verifies as the other"""
    import dataclasses
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
        air, seed = gs.plain_air(), bytes(range(32))
        proofs = {}
        for h in ("sha256", "blake2s"):
            opt = gs.Options(num_queries=20, grinding=8, hash=h)
            proofs[h] = gs.Prover(ctx, air, opt).prove(seed, base, lambda ch: gs.plain_extension_on_device(ctx, base, ch)[0], statement=pi)
            gs.verify(proofs[h], air, seed, statement=pi, expected_options=opt, required_security_bits=28)
        assert proofs["sha256"].base_root != proofs["blake2s"].base_root
        again = gs.proof_from_arrays(gs.proof_to_arrays(proofs["sha256"]))
        assert again.options.hash == "sha256"
        gs.verify(again, air, seed, statement=pi, required_security_bits=28)
        for h, other in (("sha256", "blake2s"), ("blake2s", "sha256")):
            forged = dataclasses.replace(proofs[h], options=dataclasses.replace(proofs[h].options, hash=other))
            with pytest.raises(gs.VerificationError):
                gs.verify(forged, air, seed, statement=pi, required_security_bits=28)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("hash_name", ["blake2s", "sha256"])
def test_cpp_host_writes_the_python_hosts_proof(hash_name):
    """hostlib.gl_prove (host/goldilocks_prover.cpp: the 64-bit field's claim in the C++ host - coin, transcript, every stage's call) against
    goldilocks.Prover on the same statement: the same proof, array for array - and with it, for "blake2s", the MI355X-made fixture"""
    import torch
    from sandstorm_amd import goldilocks as gs, hostlib
    from sandstorm_amd.backend import Context
    prog = pl.example_program(10)
    states, memory = pl.run(prog, 64)
    pi = pl.public_input_of(prog, states, memory)
    cols = pl.base_trace(states, memory, pi)
    ctx = Context(0)
    try:
        base = [torch.from_numpy(np.array(c, dtype=np.uint64).view(np.int64)) for c in cols]
        air, seed = gs.plain_air(), bytes(range(32))
        opt = gs.Options(num_queries=20, grinding=8, hash=hash_name)
        ext = lambda ch: gs.plain_extension_on_device(ctx, base, ch)[0]
        want = gs.proof_to_arrays(gs.Prover(ctx, air, opt).prove(seed, base, ext, statement=pi))
        proof = hostlib.gl_prove(ctx, air, opt, seed, base, ext, statement=pi)
        got = gs.proof_to_arrays(proof)
        assert set(got) == set(want)
        for k in sorted(want):
            assert np.array_equal(np.asarray(got[k]), np.asarray(want[k])), k
        gs.verify(proof, air, seed, statement=pi, expected_options=opt, required_security_bits=28)
    finally:
        ctx.close()
