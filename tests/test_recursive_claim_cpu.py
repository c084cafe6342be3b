"""The CairoVerifierClaim proof of the reference's example that the MI355X wrote (tests/golden/array_sum_recursive_cairo.proof,
made by tests/test_gpu_recursive_claim.py::test_recursive_2p14_steps_the_shipped_example: C++ host, real 93-constraint AIR,
FriendlyMerkleTree<22> + Cairo coin, CLI-default options), held on the CPU: both hosts' verifiers accept it from the public
input alone - seed, Pedersen-chained out-of-domain reseed (crypto/src/public_coin/cairo.rs:76-80), 65 queries, every
`MixedMerkleDigest` path - and reject it after a flipped bit, under another statement, and as a tree with another number of
Pedersen layers."""
import copy
import os

import pytest

from tests.test_layout_recursive import load_run

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def committed():
    with open(os.path.join(ROOT, "tests", "golden", "array_sum_recursive_cairo.proof"), "rb") as f:
        return f.read()


def test_python_verifier_accepts_the_gpu_made_cairo_claim_proof(committed):
    from sandstorm_amd import backend as be, public_input, verifier, wire
    from sandstorm_amd.layouts import recursive as rec
    from sandstorm_amd.prover import ProofOptions
    _, _, pi = load_run()
    seed = public_input.public_coin_seed(pi, be.COIN_CAIRO)
    w = wire.parse(committed, be.TREE_FRIENDLY)
    assert wire.serialize(w) == committed and w.trace_len == 1 << 18 and len(w.ood_trace) == 133
    assert {t for o in w.base_openings for t in o.tags} == {0}        # 2^19 leaves: every node level is Pedersen (19 < 22)
    positions = verifier.verify(committed, rec.verifier_air(pi), be.TREE_FRIENDLY, be.COIN_CAIRO, seed, expected_options=ProofOptions())
    assert len(positions) == len(w.base_openings) >= 60
    for at in (40, len(committed) // 3, len(committed) - 100):
        bad = bytearray(committed)
        bad[at] ^= 4
        with pytest.raises(verifier.VerificationError):
            verifier.verify(bytes(bad), rec.verifier_air(pi), be.TREE_FRIENDLY, be.COIN_CAIRO, seed)
    other = copy.deepcopy(pi)
    other.rc_max += 1
    with pytest.raises(verifier.VerificationError):
        verifier.verify(committed, rec.verifier_air(other), be.TREE_FRIENDLY, be.COIN_CAIRO, public_input.public_coin_seed(other, be.COIN_CAIRO))
    # the same bytes under the Eth claim's parts are another proof system's
    with pytest.raises(verifier.VerificationError):
        verifier.verify(committed, rec.verifier_air(pi), be.TREE_KECCAK, be.COIN_SOLIDITY, public_input.public_coin_seed(pi, be.COIN_SOLIDITY))
    # a tree of 2^19 leaves lies wholly above depth 22: fewer Pedersen layers than its height changes the bottom levels' hash
    with pytest.raises(verifier.VerificationError):
        verifier.verify(committed, rec.verifier_air(pi), be.TREE_FRIENDLY, be.COIN_CAIRO, seed, n_friendly_layers=18)


def test_cpp_verifier_accepts_the_gpu_made_cairo_claim_proof(committed):
    from sandstorm_amd import _lib, backend as be, hostlib, public_input, verifier
    from sandstorm_amd.layouts import recursive as rec
    from sandstorm_amd.prover import ProofOptions
    _, _, pi = load_run()
    seed = public_input.public_coin_seed(pi, be.COIN_CAIRO)
    air = hostlib.RecursiveHostAir(None, pi, 18)                       # host-only handle: no device involved
    try:
        positions = hostlib.verify(air, be.TREE_FRIENDLY, be.COIN_CAIRO, seed, committed, expected_options=ProofOptions(), n_friendly_layers=22)
        assert positions == verifier.verify(committed, rec.verifier_air(pi), be.TREE_FRIENDLY, be.COIN_CAIRO, seed)
        bad = bytearray(committed)
        bad[len(bad) // 2] ^= 1
        with pytest.raises(_lib.SandstormHipError):
            hostlib.verify(air, be.TREE_FRIENDLY, be.COIN_CAIRO, seed, bytes(bad), n_friendly_layers=22)
        with pytest.raises(_lib.SandstormHipError):
            hostlib.verify(air, be.TREE_FRIENDLY, be.COIN_CAIRO, seed, committed, n_friendly_layers=18)
    finally:
        air.close()
