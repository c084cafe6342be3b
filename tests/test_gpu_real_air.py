"""End to end on the reference's own example (BASELINE.json configs[0]: example/array-sum.cairo, recursive layout):
`cairo-run` artefacts -> base trace (layouts/recursive.py) -> HBM -> extension columns on the device -> the real
93-constraint AIR through the constraint VM -> proof in the reference's wire format -> host verifier with the same AIR.
EthVerifierClaim flavour for the recursive layout (src/claims.rs:29-30: unmasked Keccak tree, Solidity coin), seeded
from the public input like `sandstorm-cli prove`.  The proof is also written to gpurun_out/ (fixture for the CPU suite)."""
import os

import numpy as np
import pytest

from tests.test_layout_recursive import load_run

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def keccak_leaf_hash(vals):
    from sandstorm_amd import wire
    from sandstorm_amd.coin import keccak256
    from sandstorm_amd.layouts.recursive import P
    return keccak256(b"".join((v * wire._R % P).to_bytes(32, "big") for v in vals))


def test_prove_and_verify_the_reference_example(oracle):
    from sandstorm_amd import backend as be, extension, public_input, verifier, wire
    from sandstorm_amd.layouts import recursive as rec
    from sandstorm_amd.prover import Claim, ProofOptions, Prover
    states, memory, pi = load_run()
    cols = rec.base_trace(states, memory, pi)
    n = len(cols[0])
    ctx = be.Context(0)
    base = be.Matrix.from_host(ctx, [oracle.to_mont(c) for c in cols])        # test helper for the limb conversion only
    air = rec.make_air(ctx, pi, n)
    assert len(air.mask) == 133
    claim = Claim(air, be.LeafVariantMerkleTreeUnmasked, be.COIN_SOLIDITY)
    opt = ProofOptions(num_queries=12, grinding_factor=8)
    seed = public_input.public_coin_seed(pi, be.COIN_SOLIDITY)
    trace_cols = rec.trace_columns(ctx, base.cols, n)
    products = []

    def build_extension(challenges):
        m = extension.build_extension_columns("recursive", ctx, trace_cols, challenges)      # check=True: products close to one
        products.append(m)
        return m
    proof = Prover(ctx, claim, opt).prove(seed, base, build_extension)
    assert len(proof.fri_layers) == 5 and len(proof.ood_trace) == 133
    raw = wire.serialize(wire.from_proof(proof, keccak_leaf_hash))
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "array_sum_recursive_eth.proof"), "wb") as f:
        f.write(raw)
    positions = verifier.verify(raw, rec.verifier_air(pi), be.TREE_KECCAK, be.COIN_SOLIDITY, seed, required_security_bits=20)
    assert positions == proof.query_positions
    # a different public input (one more step claimed) is a different statement: seed and hints change
    import copy
    pi2 = copy.deepcopy(pi)
    pi2.memory_segments["execution"] = (pi.memory_segments["execution"][0], pi.memory_segments["execution"][1] + 1)
    with pytest.raises(verifier.VerificationError):
        verifier.verify(raw, rec.verifier_air(pi2), be.TREE_KECCAK, be.COIN_SOLIDITY, public_input.public_coin_seed(pi2, be.COIN_SOLIDITY), required_security_bits=20)
    ctx.close()


def test_cpp_host_proves_the_reference_example_from_its_files(oracle):
    """everything above the C ABI in C++: trace.bin / memory.bin -> base trace (host/trace_recursive.cpp) -> HBM -> the real
    AIR (host/air_recursive.cpp) -> Prover (host/prover.cpp) with the extension columns from host/extension.cpp -> the
    reference's wire format.  Byte for byte the proof of the Python mirror, and the verifier accepts it."""
    from sandstorm_amd import backend as be, extension, hostlib, public_input, verifier, wire
    from sandstorm_amd.layouts import recursive as rec
    from sandstorm_amd.prover import Claim, ProofOptions, Prover
    from tests.test_layout_recursive import EX
    _, _, pi = load_run()
    with open(os.path.join(EX, "trace.bin"), "rb") as f:
        trace_bin = f.read()
    with open(os.path.join(EX, "memory.bin"), "rb") as f:
        memory_bin = f.read()
    cols = hostlib.recursive_base_trace(trace_bin, memory_bin, pi)
    n = cols[0].shape[0]
    log_n = n.bit_length() - 1
    ctx = be.Context(0)
    base = be.Matrix.from_host(ctx, cols)
    opt = ProofOptions(num_queries=12, grinding_factor=8)
    seed = public_input.public_coin_seed(pi, be.COIN_SOLIDITY)
    air = hostlib.RecursiveHostAir(ctx, pi, log_n)
    keep = []

    def build_extension(challenges):
        m = hostlib.build_extension_columns(ctx, "recursive", [base.cols[3], base.cols[4], base.cols[5], base.cols[1], base.cols[2]], n, challenges)
        keep.append(m)
        return m.cols
    raw = hostlib.prove(ctx, air, be.TREE_KECCAK, 0, be.COIN_SOLIDITY, seed, base.cols, log_n, build_extension, opt, wire=True)
    air.close()
    verifier.verify(raw, rec.verifier_air(pi), be.TREE_KECCAK, be.COIN_SOLIDITY, seed, required_security_bits=20)
    # the Python mirror on the same statement
    pair = rec.make_air(ctx, pi, n)
    tc = rec.trace_columns(ctx, base.cols, n)
    ref = Prover(ctx, Claim(pair, be.LeafVariantMerkleTreeUnmasked, be.COIN_SOLIDITY), opt).prove(
        seed, base, lambda ch: extension.build_extension_columns("recursive", ctx, tc, ch))
    assert raw == wire.serialize(wire.from_proof(ref, keccak_leaf_hash))
    for m in keep:
        m.close()
    ctx.close()
