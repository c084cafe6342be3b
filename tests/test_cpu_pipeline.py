"""The whole proving pipeline on the CPU oracle (oracle/cpu_context.py: the oracle behind the `backend.Context`
interface), driven by the product's own host code (sandstorm_amd/prover.py).

This is the parity check of the PIPELINE, not of one kernel: the proofs committed under tests/golden/ were made on an
MI355X by the HIP kernels; the same host code over the CPU oracle must emit the same bytes.  It is also what bench.py's
`cpu_baseline` leg times and what stands in for the kernels in the multi-rank gloo tests of the sharded prover."""
import json
import os

import numpy as np
import pytest

from tests import mini_air
from tests.test_verifier import GOLD, mini_verifier_air, pv


def keccak_m20_leaf_hash(vals):
    from sandstorm_amd import wire
    from sandstorm_amd.coin import keccak256
    return keccak256(b"".join((v * wire._R % wire.P).to_bytes(32, "big") for v in vals))[:20] + bytes(12)


def cpu_mini_proof(oracle, log_n, options, seed):
    from oracle.cpu_context import CpuContext
    from sandstorm_amd import backend as be, wire
    from sandstorm_amd.coin import canonical
    from sandstorm_amd.prover import Claim, Prover
    ctx = CpuContext()
    n = 1 << log_n
    c0, c1 = mini_air.base_trace(n)
    base = be.Matrix.from_host(ctx, [oracle.to_mont(c0), oracle.to_mont(c1)])

    def build_extension(challenges):
        return be.Matrix.from_host(ctx, [oracle.to_mont(mini_air.extension_trace(c0, canonical(challenges[0])))])
    claim = Claim(mini_air.make_air(oracle.to_mont), be.LeafVariantMerkleTree, be.COIN_SOLIDITY)
    proof = Prover(ctx, claim, options).prove(seed, base, build_extension)
    return wire.serialize(wire.from_proof(proof, keccak_m20_leaf_hash))


@pytest.mark.parametrize("name,log_n,max_remainder", [("mini_proof_eth_log5.bin", 5, 4), ("mini_proof_eth_log9.bin", 9, 4),
                                                      ("mini_proof_eth_log5_nolayers.bin", 5, 32)])
def test_cpu_oracle_pipeline_emits_the_gpu_made_proof(oracle, name, log_n, max_remainder):
    """same statement, same options, same seed as tests/golden/make_mini_proof.py ran on the MI355X (C++ host + HIP
    kernels): the Python host over the CPU oracle writes the same file, byte for byte"""
    from sandstorm_amd import backend as be
    from sandstorm_amd.prover import ProofOptions
    with open(os.path.join(GOLD, "mini_proof_meta.json")) as f:
        meta = json.load(f)
    seed = bytes.fromhex(meta["seed_hex"])
    opt = ProofOptions(num_queries=12, grinding_factor=8, fri_max_remainder_coeffs=max_remainder)
    raw = cpu_mini_proof(oracle, log_n, opt, seed)
    with open(os.path.join(GOLD, name), "rb") as f:
        want = f.read()
    assert raw == want
    assert len(pv(raw, mini_verifier_air(), be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed)) >= 1


def test_cpu_oracle_pipeline_emits_the_gpu_made_proof_of_the_reference_example(oracle):
    """tests/golden/array_sum_recursive_eth.proof was made on the MI355X (tests/test_gpu_real_air.py: the reference's
    example run, the real 93-constraint recursive AIR, extension columns by the device scans).  The same host code over
    the CPU oracle - LDE, row hashing, trees, the constraint VM, out-of-domain evaluation, DEEP, FRI, proof of work -
    writes the same 80 KB, byte for byte."""
    import time
    from oracle.cpu_context import CpuContext
    from sandstorm_amd import backend as be, extension, public_input, wire
    from sandstorm_amd.coin import keccak256
    from sandstorm_amd.layouts import recursive as rec
    from sandstorm_amd.prover import Claim, ProofOptions, Prover
    from tests.test_layout_recursive import load_run
    states, memory, pi = load_run()
    cols = rec.base_trace(states, memory, pi)
    n = len(cols[0])
    ctx = CpuContext()
    t0 = time.perf_counter()
    base = be.Matrix.from_host(ctx, [oracle.to_mont(c) for c in cols])
    air = rec.make_air(ctx, pi, n)
    claim = Claim(air, be.LeafVariantMerkleTreeUnmasked, be.COIN_SOLIDITY)
    opt = ProofOptions(num_queries=12, grinding_factor=8)
    seed = public_input.public_coin_seed(pi, be.COIN_SOLIDITY)
    tc = rec.trace_columns(ctx, base.cols, n)
    proof = Prover(ctx, claim, opt).prove(seed, base, lambda ch: extension.build_extension_columns("recursive", ctx, tc, ch))

    def leaf_hash(vals):
        return keccak256(b"".join((v * wire._R % wire.P).to_bytes(32, "big") for v in vals))
    raw = wire.serialize(wire.from_proof(proof, leaf_hash))
    print("cpu proof of the example: %.1f s" % (time.perf_counter() - t0))
    with open(os.path.join(GOLD, "array_sum_recursive_eth.proof"), "rb") as f:
        assert raw == f.read()


@pytest.mark.parametrize("n_friendly", [22, 3, 1])
def test_friendly_tree_proof_on_the_wire(oracle, n_friendly):
    """A CairoVerifierClaim-flavoured proof (FriendlyMerkleTree: masked Blake2s rows, Pedersen above depth N; Cairo coin)
    of the mini AIR, made by the product's host code over the CPU oracle: serialised with the MixedMerkleDigest /
    FriendlyMerkleTreeProof encodings of crypto/src/merkle/mixed.rs:46-101 and mod.rs:168-236 (source-pinned: the
    reference ships no such proof), parsed back to the same object, verified - transcript, out-of-domain identity, every
    opening through the Blake2s / Pedersen boundary (N = 3 and 1 put it inside this small tree), DEEP, FRI - and rejected
    after tampering with a digest variant tag, a Pedersen node, a leaf."""
    from oracle.cpu_context import CpuContext
    from sandstorm_amd import backend as be, verifier, wire
    from sandstorm_amd.coin import blake2s256, canonical
    from sandstorm_amd.prover import Claim, ProofOptions, Prover

    class Tree(be.FriendlyMerkleTree):
        pass
    Tree.n_friendly = n_friendly
    ctx = CpuContext()
    log_n = 6
    n = 1 << log_n
    c0, c1 = mini_air.base_trace(n)
    base = be.Matrix.from_host(ctx, [oracle.to_mont(c0), oracle.to_mont(c1)])
    claim = Claim(mini_air.make_air(oracle.to_mont), Tree, be.COIN_CAIRO)
    opt = ProofOptions(num_queries=10, grinding_factor=6, fri_max_remainder_coeffs=4)
    seed = bytes(range(100, 132))
    proof = Prover(ctx, claim, opt).prove(seed, base, lambda ch: be.Matrix.from_host(
        ctx, [oracle.to_mont(mini_air.extension_trace(c0, canonical(ch[0])))]))

    def leaf_hash(vals):                                   # MaskedBlake2sHashFn<20> of the row's Montgomery big-endian bytes
        return bytes(12) + blake2s256(b"".join((v * wire._R % wire.P).to_bytes(32, "big") for v in vals))[12:]
    w = wire.from_proof(proof, leaf_hash)
    raw = wire.serialize(w)
    back = wire.parse(raw, be.TREE_FRIENDLY)
    assert wire.serialize(back) == raw and back.root_tags == w.root_tags and back.base_openings[0].tags == w.base_openings[0].tags
    with pytest.raises(ValueError):
        wire.parse(raw)                                    # as a Keccak proof the bytes do not parse
    args = (mini_verifier_air(), be.TREE_FRIENDLY, be.COIN_CAIRO, seed)
    kw = dict(required_security_bits=16, n_friendly_layers=n_friendly)
    positions = verifier.verify(raw, *args, **kw)
    assert positions == proof.query_positions
    depth = log_n + 1
    tags = w.base_openings[0].tags                         # siblings at depths depth-1 .. 1: Pedersen above the boundary
    assert tags == [0 if depth - 1 - k < n_friendly else 1 for k in range(len(tags))]

    def rejected(mutate, match=None):
        t = wire.parse(raw, be.TREE_FRIENDLY)
        mutate(t)
        with pytest.raises(verifier.VerificationError, match=match):
            verifier.verify(t, *args, **kw)

    def flip_tag(t):
        t.composition_openings[0].tags[-1] ^= 1
    rejected(flip_tag, match="digest variant")

    def bump_node(t):
        o = t.base_openings[1]
        o.path[-1] = ((int.from_bytes(o.path[-1], "big") + 1) % verifier.P).to_bytes(32, "big")
    rejected(bump_node, match="authentication path")
    rejected(lambda t: setattr(t.extension_openings[0], "sibling", (t.extension_openings[0].sibling + 1) % verifier.P), match="authentication path")
    rejected(lambda t: t.root_tags.__setitem__(0, 1), match="root digest variant")
    if n_friendly == 22:                                   # the C++ host's verifier (host/verifier.cpp: its own parser and tree)
        from sandstorm_amd import hostlib
        from sandstorm_amd._lib import SandstormHipError
        cpp = hostlib.HostAir(None, hostlib.AIR_MINI, log_n)
        assert hostlib.verify(cpp, be.TREE_FRIENDLY, be.COIN_CAIRO, seed, raw, required_security_bits=16) == positions
        for mutate, match in ((flip_tag, "digest variant"), (bump_node, "authentication path")):
            t = wire.parse(raw, be.TREE_FRIENDLY)
            mutate(t)
            with pytest.raises(SandstormHipError, match=match):
                hostlib.verify(cpp, be.TREE_FRIENDLY, be.COIN_CAIRO, seed, wire.serialize(t), required_security_bits=16)
        with pytest.raises(SandstormHipError):
            hostlib.verify(cpp, be.TREE_KECCAK_M20, be.COIN_CAIRO, seed, raw, required_security_bits=16)
        cpp.close()
    if n_friendly < depth:                                 # a boundary elsewhere is another tree: other hashes
        with pytest.raises(verifier.VerificationError):
            verifier.verify(raw, *args, required_security_bits=16, n_friendly_layers=n_friendly + 1)
