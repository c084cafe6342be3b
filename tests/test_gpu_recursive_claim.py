"""The north-star's own configuration, as `sandstorm-cli prove` defines it for the recursive layout:

    cli/src/main.rs:95-99      Layout::Recursive -> recursive::CairoVerifierClaim
    src/claims.rs:10,31-32     = FriendlyMerkleTree<22, PedersenHashFn> trees (crypto/src/merkle/mixed.rs:106-155: Blake2s
                               masked-20 below depth 22, Pedersen above, the digests read as field elements at the boundary)
                               + CairoVerifierPublicCoin (crypto/src/public_coin/cairo.rs:60-174: Pedersen chain over the
                               out-of-domain values, Blake2s reseeds)

with the REAL 93-constraint AIR, through the C++ host (everything above the C ABI in C++: trace.bin / memory.bin -> base
trace -> HBM -> Prover with the extension columns' scans on the device -> the reference's wire format with
`MixedMerkleDigest` / `FriendlyMerkleTreeProof` encodings), CLI-default options, at

    2^14 steps  the reference's shipped example run (BASELINE configs[0]'s fixture)
    2^16 steps  BASELINE configs[1]'s size: 2^21 leaves - every tree node is Pedersen
    2^20 steps  the north-star's size: 2^25 leaves - the three lowest node levels are Blake2s, the boundary is crossed
                inside every authentication path and the wire format carries both digest variants

Statements above 2^14 steps are the example run padded with its final state (the program ends in `jmp rel 0`:
tests/test_layout_recursive.py::recursive_example).  The oracle cannot run these sizes in seconds; what holds the kernels
is that a proof of a true statement verifies - by the C++ verifier AND by the Python one, whose AIR evaluation at the
out-of-domain point is the independent restatement (layouts/recursive.py) - that a flipped bit does not, and that the
sharded driver writes the same bytes."""
import os
import time

import pytest

from tests.test_layout_recursive import recursive_example

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_FRIENDLY = 22                                    # src/claims.rs:10: FriendlyMerkleTree<22, _>


def blake2s_m20_leaf_hash(vals):
    """hash_rows of the CairoVerifierClaim: Blake2s-256 of the row's 32-byte big-endian Montgomery images, last 20 bytes kept
    (crypto/src/hash/blake2s.rs:94-100, hash/mod.rs:15-23)"""
    import hashlib
    from sandstorm_amd import wire
    d = hashlib.blake2s(b"".join((v * wire._R % wire.P).to_bytes(32, "big") for v in vals)).digest()
    return bytes(12) + d[12:]


def statement(log_steps):
    from sandstorm_amd import backend as be, binary, hostlib, public_input
    t0 = time.time()
    states, memory, pi = recursive_example(log_steps)
    cols = hostlib.recursive_base_trace(binary.write_register_states(states), binary.write_memory(memory), pi)
    del states, memory
    log_n = log_steps + 4
    assert len(cols) == 7 and cols[0].shape[0] == 1 << log_n
    t1 = time.time()
    ctx = be.Context(0)
    base = be.Matrix.from_host(ctx, cols)
    del cols
    air = hostlib.RecursiveHostAir(ctx, pi, log_n)
    assert air.mask_size == 133
    seed = public_input.public_coin_seed(pi, be.COIN_CAIRO)
    print("2^%d steps: base trace on the host %.1f s, to the device %.1f s" % (log_steps, t1 - t0, time.time() - t1))
    return ctx, base, air, seed, pi, log_n


def prove_and_verify(log_steps, python_verifier=True):
    from sandstorm_amd import backend as be, hostlib, verifier, wire
    from sandstorm_amd.layouts import recursive as rec
    from sandstorm_amd.prover import ProofOptions
    ctx, base, air, seed, pi, log_n = statement(log_steps)
    keep = []

    def build_extension(challenges):               # check=True: the memory / range-check / diluted products must close
        m = hostlib.build_extension_columns(ctx, "recursive", [base.cols[rec.COL_NPC], base.cols[rec.COL_MEMORY], base.cols[rec.COL_RANGE_CHECK],
                                                                base.cols[rec.COL_DILUTED_UNORDERED], base.cols[rec.COL_DILUTED_ORDERED]],
                                            1 << log_n, challenges)
        keep.append(m)
        return m.cols
    try:
        opt = ProofOptions()                       # cli/src/main.rs:51-60 defaults: 65 queries, blowup 2, 16 grinding bits
        t0 = time.time()
        raw = hostlib.prove(ctx, air, be.TREE_FRIENDLY, N_FRIENDLY, be.COIN_CAIRO, seed, base.cols, log_n, build_extension, opt, wire=True)
        t_prove = time.time() - t0
        parsed = wire.parse(raw, be.TREE_FRIENDLY)
        assert parsed.trace_len == 1 << log_n and len(parsed.ood_trace) == 133 and len(parsed.ood_composition) == 2
        assert len(parsed.fri_layers) == verifier.fri_layer_count(1 << log_n, 8, 16)[0]
        assert wire.serialize(parsed) == raw
        # which digest variants the authentication paths carry: a tree of 2^k leaves has node levels at depths 0..k-1, the
        # Blake2s ones are those at depth >= 22 (mixed.rs:106-125)
        log_leaves = log_n + 1
        tags = {t for o in parsed.base_openings for t in o.tags}
        assert tags == ({0, 1} if log_leaves > N_FRIENDLY else {0}), tags
        if log_leaves > N_FRIENDLY:
            for o in parsed.base_openings:         # bottom-up: Blake2s siblings first, Pedersen from the boundary on
                k = log_leaves - N_FRIENDLY        # the node levels at depths 22 .. log_leaves - 1 hold Blake2s digests
                assert len(o.tags) == log_leaves - 1 and list(o.tags) == [1] * k + [0] * (len(o.tags) - k), o.tags
        assert parsed.root_tags == [0, 0, 0]
        positions = hostlib.verify(air, be.TREE_FRIENDLY, be.COIN_CAIRO, seed, raw, expected_options=opt, n_friendly_layers=N_FRIENDLY)
        assert len(positions) == len(parsed.base_openings) and len(set(positions)) == len(positions)
        if python_verifier:
            assert verifier.verify(raw, rec.verifier_air(pi), be.TREE_FRIENDLY, be.COIN_CAIRO, seed, expected_options=opt,
                                   n_friendly_layers=N_FRIENDLY) == positions
        # a flipped bit is caught: in the middle of the proof, in the first root, in the last authentication path
        for at in (len(raw) // 2, 20, len(raw) - 40):
            bad = bytearray(raw)
            bad[at] ^= 1
            with pytest.raises(Exception):
                hostlib.verify(air, be.TREE_FRIENDLY, be.COIN_CAIRO, seed, bytes(bad), expected_options=opt, n_friendly_layers=N_FRIENDLY)
        # and the boundary is where the claim says: the same bytes do not verify as a tree with 21 or 23 Pedersen layers
        if log_leaves > N_FRIENDLY:
            for other in (N_FRIENDLY - 1, N_FRIENDLY + 1):
                with pytest.raises(Exception):
                    hostlib.verify(air, be.TREE_FRIENDLY, be.COIN_CAIRO, seed, raw, expected_options=opt, n_friendly_layers=other)
        print("recursive 2^%d steps, CairoVerifierClaim: proof of %d bytes in %.2f s (first call: plans and tables included), %d queries verified"
              % (log_steps, len(raw), t_prove, len(positions)))
        return raw
    finally:
        for m in keep:
            m.close()
        air.close()
        del base
        ctx.close()


def test_recursive_2p14_steps_the_shipped_example():
    """BASELINE configs[0]'s fixture under the claim the CLI picks for it; the Python mirror writes the same bytes"""
    from sandstorm_amd import backend as be, extension, public_input, wire
    from sandstorm_amd.layouts import recursive as rec
    from sandstorm_amd.prover import Claim, ProofOptions, Prover
    raw = prove_and_verify(14)
    ctx, base, air, seed, pi, log_n = statement(14)
    air.close()
    n = 1 << log_n
    pair = rec.make_air(ctx, pi, n)
    tc = rec.trace_columns(ctx, base.cols, n)
    ref = Prover(ctx, Claim(pair, be.FriendlyMerkleTree, be.COIN_CAIRO), ProofOptions()).prove(
        seed, base, lambda ch: extension.build_extension_columns("recursive", ctx, tc, ch))
    assert wire.serialize(wire.from_proof(ref, blake2s_m20_leaf_hash)) == raw
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "array_sum_recursive_cairo.proof"), "wb") as f:
        f.write(raw)
    ctx.close()


@pytest.fixture(scope="module")
def proof_2p16():
    return prove_and_verify(16)


def test_recursive_2p16_steps_cairo_verifier_claim(proof_2p16):
    """BASELINE configs[1] as the CLI defines it: 2^20 rows x 10 columns, all 21 node levels Pedersen"""
    assert len(proof_2p16) > 100_000


@pytest.mark.parametrize("log_steps", [14, 16])
def test_files_to_proof_in_one_call_writes_the_same_bytes(log_steps, request):
    """hostlib.prove_files (host_capi.cpp ssh_prove_files; what bench.py's end_to_end leg times): the generator on a thread of its own
    writes pinned columns, every column is uploaded the moment it is final (ss_upload_async), the prover extends the columns as they
    land (one transform pair per column, in arrival order) - the proof is the one the batch path writes: the committed 2^14-step
    fixture, and this run's own 2^16-step proof"""
    import numpy as np
    from sandstorm_amd import backend as be, binary, hostlib, public_input
    from sandstorm_amd.layouts import recursive as rec
    if log_steps == 14:
        with open(os.path.join(ROOT, "tests", "golden", "array_sum_recursive_cairo.proof"), "rb") as f:
            want = f.read()
    else:
        want = request.getfixturevalue("proof_2p16")
    states, memory, pi = recursive_example(log_steps)
    trace_bin, memory_bin = binary.write_register_states(states), binary.write_memory(memory)
    del states, memory
    log_n = log_steps + 4
    n = 1 << log_n
    ctx = be.Context(0)
    try:
        import torch
        pinned = [torch.empty((n, 4), dtype=torch.int64).pin_memory() if torch.cuda.is_available() else torch.empty((n, 4), dtype=torch.int64) for _ in range(7)]
        views = [t.numpy().view("uint64") for t in pinned]
    except ImportError:
        views = [np.zeros((n, 4), dtype=np.uint64) for _ in range(7)]
    dev = [ctx.alloc(32 * n) for _ in range(7)]
    air = hostlib.RecursiveHostAir(ctx, pi, log_n)
    seed = public_input.public_coin_seed(pi, be.COIN_CAIRO)
    aux_idx = (rec.COL_NPC, rec.COL_MEMORY, rec.COL_RANGE_CHECK, rec.COL_DILUTED_UNORDERED, rec.COL_DILUTED_ORDERED)
    keep = []

    def build_extension(challenges):
        keep.append(hostlib.build_extension_columns(ctx, "recursive", [dev[c] for c in aux_idx], n, challenges))
        return keep[-1].cols
    for _ in range(2):                                   # twice: tickets and events are per call
        raw, times = hostlib.prove_files(ctx, "recursive", trace_bin, memory_bin, pi, None, views, dev, air, be.TREE_FRIENDLY, N_FRIENDLY, be.COIN_CAIRO, seed,
                                         build_extension)
        assert raw == want
        assert 0 < times["trace_gen_s"] <= times["total_s"]
    for m in keep:
        m.close()
    air.close()
    ctx.close()


@pytest.mark.parametrize("log_steps", [14, 16])
def test_files_to_proof_through_the_device_generator_writes_the_same_bytes(log_steps, request):
    """hostlib.prove_files_device (host_capi.cpp ssh_prove_files_device; what bench.py's end_to_end leg times since round 6): the
    files' bytes go up as they are, the base columns are made in HBM (csrc/trace.hip), the prover goes on from there - the proof is
    the batch path's: the committed 2^14-step fixture, and this run's own 2^16-step proof"""
    from sandstorm_amd import backend as be, binary, hostlib, public_input
    from sandstorm_amd.layouts import recursive as rec
    if log_steps == 14:
        with open(os.path.join(ROOT, "tests", "golden", "array_sum_recursive_cairo.proof"), "rb") as f:
            want = f.read()
    else:
        want = request.getfixturevalue("proof_2p16")
    states, memory, pi = recursive_example(log_steps)
    trace_bin, memory_bin = binary.write_register_states(states), binary.write_memory(memory)
    del states, memory
    log_n = log_steps + 4
    n = 1 << log_n
    ctx = be.Context(0)
    dev = [ctx.alloc(32 * n) for _ in range(7)]
    air = hostlib.RecursiveHostAir(ctx, pi, log_n)
    seed = public_input.public_coin_seed(pi, be.COIN_CAIRO)
    aux_idx = (rec.COL_NPC, rec.COL_MEMORY, rec.COL_RANGE_CHECK, rec.COL_DILUTED_UNORDERED, rec.COL_DILUTED_ORDERED)
    keep = []

    def build_extension(challenges):
        keep.append(hostlib.build_extension_columns(ctx, "recursive", [dev[c] for c in aux_idx], n, challenges))
        return keep[-1].cols
    for _ in range(2):
        raw, times = hostlib.prove_files_device(ctx, "recursive", trace_bin, memory_bin, pi, None, dev, air, be.TREE_FRIENDLY, N_FRIENDLY, be.COIN_CAIRO, seed,
                                                build_extension)
        assert raw == want
        assert 0 < times["trace_gen_s"] <= times["total_s"]
    # files the generator refuses: nothing is proven, the message is the generator's, the context stays usable
    from sandstorm_amd._lib import SandstormHipError
    with pytest.raises(SandstormHipError, match="power of two"):
        hostlib.prove_files_device(ctx, "recursive", trace_bin[:24 * 3000], memory_bin, pi, None, dev, air, be.TREE_FRIENDLY, N_FRIENDLY, be.COIN_CAIRO, seed, build_extension)
    ctx.ntt([dev[0]], 10, be.FORWARD, None)
    for m in keep:
        m.close()
    air.close()
    ctx.close()


def test_files_to_proof_reports_the_generators_error():
    """ssh_prove_files with files the generator refuses (a trace that is not a power of two of cycles; a public memory that does not
    fit the run): the generator's thread fails, the prover - waiting for its first column - gives up with THAT message, the thread is
    joined, the context stays usable"""
    import dataclasses
    import numpy as np
    from sandstorm_amd import backend as be, binary, hostlib, public_input
    from sandstorm_amd._lib import SandstormHipError
    states, memory, pi = recursive_example(14)
    trace_bin, memory_bin = binary.write_register_states(states), binary.write_memory(memory)
    log_n = 18
    n = 1 << log_n
    ctx = be.Context(0)
    views = [np.zeros((n, 4), dtype=np.uint64) for _ in range(7)]
    dev = [ctx.alloc(32 * n) for _ in range(7)]
    air = hostlib.RecursiveHostAir(ctx, pi, log_n)
    seed = public_input.public_coin_seed(pi, be.COIN_CAIRO)
    args = (views, dev, air, be.TREE_FRIENDLY, N_FRIENDLY, be.COIN_CAIRO, seed, lambda ch: [])
    with pytest.raises(SandstormHipError, match="power of two"):
        hostlib.prove_files(ctx, "recursive", trace_bin[:24 * 3000], memory_bin, pi, None, [np.zeros((16 * 3000, 4), dtype=np.uint64) for _ in range(7)], *args[1:])
    bad = dataclasses.replace(pi, public_memory=pi.public_memory + [(0xFFFFFFF0, 7)])
    with pytest.raises(SandstormHipError, match="memory gaps"):
        hostlib.prove_files(ctx, "recursive", trace_bin, memory_bin, bad, None, *args)
    # and the context still serves: a transform on it
    ctx.ntt([dev[0]], 10, be.FORWARD, None)
    air.close()
    ctx.close()
