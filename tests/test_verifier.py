"""The host verifier (sandstorm_amd/verifier.py) on proofs in the reference's wire format.

CPU: the committed proofs tests/golden/mini_proof_eth_log{5,9}.bin (made on an MI355X by tests/golden/
make_mini_proof.py) verify, and do not after any single-section tampering.  GPU: a fresh proof from the C++ host."""
import json
import os

import pytest

from tests import mini_air

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def pv(*a, **k):
    """sandstorm_amd.verifier.verify at the fixtures' security level (12 queries + 8 grinding bits; the CLI's default is 80)"""
    from sandstorm_amd import verifier
    k.setdefault("required_security_bits", 20)
    return verifier.verify(*a, **k)


def cv(*a, **k):
    """the C++ host's verifier, likewise"""
    from sandstorm_amd import hostlib
    k.setdefault("required_security_bits", 20)
    return hostlib.verify(*a, **k)


def mini_verifier_air():
    from sandstorm_amd.verifier import VerifierAir
    return VerifierAir(2, 1, 1, mini_air.MASK,
                       composition=lambda n, ch, a: mini_air.composition(n, ch[0], a),
                       table_at=lambda n, x, t: mini_air.table_at(n, x))


def load_fixture(log_n):
    with open(os.path.join(GOLD, "mini_proof_meta.json")) as f:
        meta = json.load(f)
    with open(os.path.join(GOLD, "mini_proof_eth_log%d.bin" % log_n), "rb") as f:
        return f.read(), bytes.fromhex(meta["seed_hex"]), meta


@pytest.mark.parametrize("log_n", [5, 9])
def test_committed_proof_verifies(log_n):
    from sandstorm_amd import backend as be, verifier, wire
    raw, seed, meta = load_fixture(log_n)
    positions = pv(raw, mini_verifier_air(), be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed)
    w = wire.parse(raw)
    assert w.options == meta["options"] and w.trace_len == 1 << log_n
    assert len(positions) == len(w.base_openings) and positions == sorted(set(positions))
    assert wire.serialize(w) == raw


def test_tampered_proofs_are_rejected():
    from sandstorm_amd import backend as be, verifier, wire
    raw, seed, _ = load_fixture(5)
    air = mini_verifier_air()

    def rejected(mutate, match=None):
        w = wire.parse(raw)
        mutate(w)
        with pytest.raises(verifier.VerificationError, match=match):
            pv(w, air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed)

    def bump(lst, i):
        lst[i] = (lst[i] + 1) % verifier.P

    rejected(lambda w: setattr(w, "pow_nonce", w.pow_nonce + 1))                        # proof of work / positions
    rejected(lambda w: bump(w.ood_trace, 2))                                            # transcript diverges
    rejected(lambda w: bump(w.ood_composition, 1))
    rejected(lambda w: bump(w.remainder, 0))
    rejected(lambda w: setattr(w, "base_root", bytes(32)))
    # data below the transcript: the replay still succeeds, the data checks must catch it
    rejected(lambda w: bump(w.base_rows, 0), match="base trace")
    rejected(lambda w: bump(w.extension_rows, 0), match="extension trace")
    rejected(lambda w: bump(w.composition_rows, 1), match="composition trace")
    rejected(lambda w: bump(w.fri_layers[0].rows, 3), match="FRI layer 0")

    def swap_path(w):
        o = w.composition_openings[0]
        o.path[0] = bytes(32)
    rejected(swap_path, match="authentication path")
    with pytest.raises(verifier.VerificationError):
        pv(raw, air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, bytes(32))     # wrong public-coin seed
    with pytest.raises(verifier.VerificationError, match="malformed"):
        pv(raw[:-5], air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed)
    with pytest.raises(verifier.VerificationError):
        pv(raw, air, be.TREE_KECCAK, be.COIN_SOLIDITY, seed)              # unmasked tree: other hashes


def test_older_conventions_are_a_different_statement():
    """the same bytes do not verify under the older proof's conventions (natural order, normalised fold)"""
    from sandstorm_amd import backend as be, verifier
    from sandstorm_amd.prover import Conventions
    raw, seed, _ = load_fixture(5)
    with pytest.raises(verifier.VerificationError):
        pv(raw, mini_verifier_air(), be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed,
                        Conventions(bitrev_commit=False, fri_unnormalised=False, remainder_unshifted=False, fri_alpha_times_offset=False))


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [6, 10])
def test_fresh_cpp_host_proof_verifies(oracle, log_n):
    from sandstorm_amd import backend as be, hostlib, verifier
    from sandstorm_amd.coin import canonical
    from sandstorm_amd.prover import ProofOptions
    ctx = be.Context(0)
    n = 1 << log_n
    c0, c1 = mini_air.base_trace(n)
    base = be.Matrix.from_host(ctx, [oracle.to_mont(c0), oracle.to_mont(c1)])
    keep = []

    def build_extension(challenges):
        m = be.Matrix.from_host(ctx, [oracle.to_mont(mini_air.extension_trace(c0, canonical(challenges[0])))])
        keep.append(m)
        return m.cols
    opt = ProofOptions(num_queries=20, grinding_factor=10, fri_max_remainder_coeffs=8)
    seed = bytes(reversed(range(32)))
    air = hostlib.HostAir(ctx, hostlib.AIR_MINI, log_n)
    raw = hostlib.prove(ctx, air, be.TREE_KECCAK_M20, 0, be.COIN_SOLIDITY, seed, base.cols, log_n, build_extension, opt, wire=True)
    air.close()
    assert len(pv(raw, mini_verifier_air(), be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed)) <= 20
    ctx.close()


def test_committed_proof_of_the_reference_example_verifies():
    """tests/golden/array_sum_recursive_eth.proof: the reference's array-sum example (recursive layout, 2^14 steps)
    proven on an MI355X by tests/test_gpu_real_air.py with the restated 93-constraint AIR, seeded from its
    air-public-input.json; verified here on the CPU with the same AIR; any other public input is rejected"""
    import copy
    from sandstorm_amd import backend as be, public_input, verifier, wire
    from sandstorm_amd.layouts import recursive as rec
    pi = public_input.AirPublicInput.from_json(os.path.join(GOLD, "air_public_input_array_sum.json"))
    with open(os.path.join(GOLD, "array_sum_recursive_eth.proof"), "rb") as f:
        raw = f.read()
    seed = public_input.public_coin_seed(pi, be.COIN_SOLIDITY)
    positions = pv(raw, rec.verifier_air(pi), be.TREE_KECCAK, be.COIN_SOLIDITY, seed)
    w = wire.parse(raw)
    assert w.trace_len == 16 * pi.n_steps and len(w.ood_trace) == 133 and len(positions) == len(w.base_openings)
    pi2 = copy.deepcopy(pi)
    pi2.rc_max += 1                                           # a different claim: other seed, other hints
    with pytest.raises(verifier.VerificationError):
        pv(raw, rec.verifier_air(pi2), be.TREE_KECCAK, be.COIN_SOLIDITY, public_input.public_coin_seed(pi2, be.COIN_SOLIDITY))
    # same seed, wrong hint: the transcript replays, the out-of-domain identity must catch it
    with pytest.raises(verifier.VerificationError, match="out-of-domain identity"):
        pv(raw, rec.verifier_air(pi2), be.TREE_KECCAK, be.COIN_SOLIDITY, seed)


def test_cpp_verifier_on_the_committed_proofs():
    """sandstorm_amd/host/verifier.cpp (wire parser, transcript, OOD identity through the C++ AIRs, openings, DEEP, FRI):
    accepts the three committed proofs with the positions the Python verifier finds, rejects tampered bytes with the
    same diagnosis, and rejects another public input"""
    import copy
    from sandstorm_amd import backend as be, hostlib, public_input, verifier, wire
    from sandstorm_amd._lib import SandstormHipError
    from sandstorm_amd.layouts import recursive as rec
    for log_n in (5, 9):
        raw, seed, _ = load_fixture(log_n)
        air = hostlib.HostAir(None, hostlib.AIR_MINI, log_n)
        assert cv(air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, raw) == \
            pv(raw, mini_verifier_air(), be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed)
        if log_n == 5:
            w = wire.parse(raw)
            w.base_rows[0] = (w.base_rows[0] + 1) % verifier.P
            with pytest.raises(SandstormHipError, match="base trace"):
                cv(air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, wire.serialize(w))
            w = wire.parse(raw)
            w.fri_layers[0].rows[3] = (w.fri_layers[0].rows[3] + 1) % verifier.P
            with pytest.raises(SandstormHipError, match="FRI layer 0"):
                cv(air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, wire.serialize(w))
            w = wire.parse(raw)
            w.ood_composition[1] = (w.ood_composition[1] + 1) % verifier.P
            with pytest.raises(SandstormHipError):
                cv(air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, wire.serialize(w))
            with pytest.raises(SandstormHipError, match="malformed"):
                cv(air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, raw[:-3])
            with pytest.raises(SandstormHipError):
                cv(air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, raw, shipped_conventions=False)
        air.close()
    pi = public_input.AirPublicInput.from_json(os.path.join(GOLD, "air_public_input_array_sum.json"))
    with open(os.path.join(GOLD, "array_sum_recursive_eth.proof"), "rb") as f:
        raw = f.read()
    seed = public_input.public_coin_seed(pi, be.COIN_SOLIDITY)
    air = hostlib.RecursiveHostAir(None, pi, 18)
    assert cv(air, be.TREE_KECCAK, be.COIN_SOLIDITY, seed, raw) == \
        pv(raw, rec.verifier_air(pi), be.TREE_KECCAK, be.COIN_SOLIDITY, seed)
    air.close()
    pi2 = copy.deepcopy(pi)
    pi2.rc_max += 1
    air2 = hostlib.RecursiveHostAir(None, pi2, 18)
    with pytest.raises(SandstormHipError, match="out-of-domain identity"):
        cv(air2, be.TREE_KECCAK, be.COIN_SOLIDITY, seed, raw)              # same transcript, wrong hint
    air2.close()


REFERENCE_PROOF = "/root/reference/bootloader-proof.bin"


@pytest.mark.skipif(not os.path.exists(REFERENCE_PROOF), reason="the reference checkout is only mounted in the build container")
def test_reference_proof_passes_every_check_below_the_transcript(golden):
    """The reference's own shipped recursive-layout proof through this repo's verifier, minus the transcript (its public
    input is not shipped): with the out-of-domain point and DEEP coefficient recovered from its data
    (deep_pin_recursive.json) and the FRI challenges recovered from its layers (saved_proof_openings_recursive.json),
    every Merkle opening, the DEEP value of all 40 queries, the whole FRI chain and the 64-coefficient remainder check out
    under the mask of layouts/recursive.py and the default conventions; perturbing any recovered value fails."""
    from sandstorm_amd import backend as be, verifier, wire
    from sandstorm_amd.layouts import recursive as rec
    with open(REFERENCE_PROOF, "rb") as f:
        w = wire.parse(f.read())
    pin, fri = golden("deep_pin_recursive.json"), golden("saved_proof_openings_recursive.json")
    z, alpha = int(pin["z"], 16), int(pin["deep_alpha"], 16)
    consts = [int(c, 16) for c in fri["alpha_over_offset"]] + [int(fri["last_alpha_over_offset"], 16)]
    fri_alphas = [c * pow(3, 8 ** i, verifier.P) % verifier.P for i, c in enumerate(consts)]       # alpha_i = (alpha_i / offset_i) * 3^(8^i)
    assert len(fri_alphas) == len(w.fri_layers) == 4 and len(w.remainder) == 64
    args = (w, rec.mask(), 7, 3, be.TREE_KECCAK_M20)
    assert verifier.check_proof_data(*args, z, alpha, fri_alphas, pin["positions"]) == pin["positions"]
    with pytest.raises(verifier.VerificationError, match="DEEP composition value"):
        verifier.check_proof_data(*args, z, alpha + 1, fri_alphas, pin["positions"])
    with pytest.raises(verifier.VerificationError, match="DEEP composition value"):
        verifier.check_proof_data(*args, z + 1, alpha, fri_alphas, pin["positions"])
    with pytest.raises(verifier.VerificationError, match="FRI layer 1 does not fold"):
        verifier.check_proof_data(*args, z, alpha, fri_alphas[:1] + [fri_alphas[1] + 1] + fri_alphas[2:], pin["positions"])
    with pytest.raises(verifier.VerificationError, match="remainder"):
        verifier.check_proof_data(*args, z, alpha, fri_alphas[:3] + [fri_alphas[3] + 1], pin["positions"])
    with pytest.raises(verifier.VerificationError):
        verifier.check_proof_data(w, rec.mask(), 7, 3, be.TREE_KECCAK, z, alpha, fri_alphas, pin["positions"])      # unmasked hashes


def test_parsers_survive_mutated_proofs():
    """both wire parsers / verifiers take untrusted bytes: random byte flips, truncations, length-field blow-ups and
    splices of a valid proof must end in a clean rejection (VerificationError / SandstormHipError), never in a crash or
    an acceptance"""
    import random
    from sandstorm_amd import backend as be, hostlib, verifier
    from sandstorm_amd._lib import SandstormHipError
    raw, seed, _ = load_fixture(5)
    air_py, air_cpp = mini_verifier_air(), hostlib.HostAir(None, hostlib.AIR_MINI, 5)
    rng = random.Random(2024)
    accepted = 0
    for it in range(400):
        b = bytearray(raw)
        kind = it % 4
        if kind == 0:                                   # flip 1..3 random bytes
            for _ in range(rng.randrange(1, 4)):
                i = rng.randrange(len(b))
                b[i] ^= 1 << rng.randrange(8)
        elif kind == 1:                                 # truncate / extend
            b = b[:rng.randrange(len(b))] if rng.random() < 0.7 else b + bytes(rng.randrange(1, 40))
        elif kind == 2:                                 # overwrite an aligned 8-byte field with a huge or odd count
            i = rng.randrange(0, len(b) - 8)
            b[i:i + 8] = rng.choice([2**63, 2**32 + 7, 0, 2**64 - 1, len(b) + 1]).to_bytes(8, "little")
        else:                                           # splice a chunk from elsewhere
            i, j, ln = rng.randrange(len(b)), rng.randrange(len(b)), rng.randrange(1, 200)
            b[i:i + ln] = b[j:j + ln]
        b = bytes(b)
        if b == raw:
            continue
        outcomes = []
        try:
            pv(b, air_py, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed)
            outcomes.append("ok")
        except verifier.VerificationError:
            outcomes.append("rejected")
        try:
            cv(air_cpp, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, b)
            outcomes.append("ok")
        except SandstormHipError:
            outcomes.append("rejected")
        assert outcomes[0] == outcomes[1], (it, kind, outcomes)
        accepted += outcomes[0] == "ok"
    assert accepted == 0
    air_cpp.close()


def _forged_mini_proof(options, trace_len=32):
    """the advisor's forgery: no rows, no openings, arbitrary roots and an arbitrary out-of-domain
    vector; ood_composition[0] is solved from the out-of-domain identity, so everything the verifier replays from the
    transcript is consistent.  Only the query phase (absent here) ties such bytes to a trace."""
    from sandstorm_amd import air_program as ap, backend as be, verifier, wire
    from sandstorm_amd.coin import PublicCoin
    P = verifier.P
    seed = bytes(range(32))
    n = trace_len
    w = wire.WireProof(list(options), n, b"\x11" * 32, b"\x22" * 32, b"\x33" * 32)
    w.ood_trace = [5, 6, 7, 8, 9, 10]
    coin = PublicCoin(be.COIN_SOLIDITY, seed)
    coin.reseed_with_digest(w.base_root)
    gamma = wire._canon(coin.draw())
    coin.reseed_with_digest(w.extension_root)
    alpha = wire._canon(coin.draw())
    coin.reseed_with_digest(w.composition_root)
    z = wire._canon(coin.draw())
    cell = dict(zip(mini_air.MASK, w.ood_trace))
    lhs = ap.evaluate(mini_air.composition(n, gamma, alpha), P, z, lambda c, o: cell[(c, o)], lambda t: mini_air.table_at(n, z))
    h1 = 12345
    w.ood_composition = [(lhs - z * h1) % P, h1]
    nlayers, bound = 0, n
    while bound > options[4]:
        bound //= options[3]
        nlayers += 1
    assert nlayers == 0
    w.remainder = [1] * max(1, bound)
    return wire.serialize(w), seed


def test_forged_proof_without_queries_is_rejected():
    """the proof's options are untrusted.  A proof that declares zero queries and no grinding carries no
    evidence at all; both verifiers refuse it - by the explicit query-count check and by the security level
    (`claim.verify(proof, required_security_bits)`, cli/src/main.rs:176) even when asked for 0 bits."""
    from sandstorm_amd import backend as be, hostlib, verifier
    from sandstorm_amd._lib import SandstormHipError
    raw, seed = _forged_mini_proof([0, 2, 0, 8, 32])
    air = hostlib.HostAir(None, hostlib.AIR_MINI, 5)
    for bits in (0, 20, 80):
        with pytest.raises(verifier.VerificationError, match="no queries"):
            verifier.verify(raw, mini_verifier_air(), be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, required_security_bits=bits)
        with pytest.raises(SandstormHipError, match="no queries"):
            hostlib.verify(air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, raw, required_security_bits=bits)
    # one query, no grinding: 1 conjectured bit
    raw, seed = _forged_mini_proof([1, 2, 0, 8, 32])
    with pytest.raises(verifier.VerificationError, match="1 bits of conjectured security, 20 required"):
        pv(raw, mini_verifier_air(), be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed)
    with pytest.raises(SandstormHipError, match="1 bits of conjectured security, 20 required"):
        cv(air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, raw)
    # and with the security requirement waived the (absent) query data is what fails
    with pytest.raises(verifier.VerificationError, match="rows / openings count"):
        pv(raw, mini_verifier_air(), be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, required_security_bits=0)
    with pytest.raises(SandstormHipError, match="rows / openings count"):
        cv(air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, raw, required_security_bits=0)
    air.close()


def test_security_level_follows_the_reference_constants():
    """hash/keccak.rs:17,64, blake2s.rs:14,67, pedersen.rs:48; the CLI defaults (65 queries, blowup 2, 16 bits) give 81
    query-phase bits, capped at 80 by the masked-20 trees - exactly the CLI's default requirement (main.rs:66-67)"""
    from sandstorm_amd import backend as be
    from sandstorm_amd.verifier import conjectured_security_bits as sec
    assert sec([65, 2, 16, 8, 16], 1 << 21, be.TREE_KECCAK_M20, be.COIN_SOLIDITY) == 80
    assert sec([65, 2, 16, 8, 16], 1 << 21, be.TREE_KECCAK, be.COIN_SOLIDITY) == 81
    assert sec([65, 2, 16, 8, 16], 1 << 21, be.TREE_FRIENDLY, be.COIN_CAIRO) == 80
    assert sec([64, 2, 15, 8, 16], 1 << 21, be.TREE_KECCAK, be.COIN_SOLIDITY) == 79
    assert sec([200, 4, 30, 8, 16], 1 << 21, be.TREE_KECCAK, be.COIN_SOLIDITY) == 128
    assert sec([16, 2, 16, 8, 16], 1 << 21, be.TREE_KECCAK_M20, be.COIN_SOLIDITY) == 32


def test_malformed_options_are_rejected_before_any_geometry():
    """max_remainder 0 / not a power of two, a trace length that is not the remainder bound times a power of
    the folding factor, layer counts beyond the domain - all rejected up front by both verifiers (the C++ one used to
    compute `1 << (log_N - log_fold * layers)` on them)"""
    from sandstorm_amd import backend as be, hostlib, verifier, wire
    from sandstorm_amd._lib import SandstormHipError
    raw, seed = _forged_mini_proof([12, 2, 8, 8, 32])
    air = hostlib.HostAir(None, hostlib.AIR_MINI, 5)
    for opts, n, what in (([12, 2, 8, 8, 0], 8, "max remainder"), ([12, 2, 8, 8, 3], 32, "max remainder"),
                          ([12, 2, 8, 8, 2], 4, "power of the folding factor"), ([12, 2, 8, 16, 1], 32, "power of the folding factor")):
        w = wire.parse(raw)
        w.options, w.trace_len = opts, n
        b = wire.serialize(w)
        with pytest.raises(verifier.VerificationError, match=what):
            pv(b, mini_verifier_air(), be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed)
        with pytest.raises(SandstormHipError, match=what):
            cv(air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, b)
    air.close()


def test_proof_without_fri_layers_binds_the_remainder():
    """when the trace fits the remainder bound there is no FRI layer; the DEEP value of every query
    must then equal the remainder polynomial at that point - otherwise nothing ties the commitments to a low-degree
    polynomial.  tests/golden/mini_proof_eth_log5_nolayers.bin (GPU-made) verifies in both hosts; below the transcript
    (same out-of-domain point, DEEP coefficient and positions) one changed remainder coefficient is caught by that
    check, and through the front door the changed transcript fails the proof of work."""
    from sandstorm_amd import backend as be, hostlib, verifier, wire
    from sandstorm_amd._lib import SandstormHipError
    with open(os.path.join(GOLD, "mini_proof_eth_log5_nolayers.bin"), "rb") as f:
        raw = f.read()
    seed = bytes(range(32))
    w = wire.parse(raw)
    assert w.fri_layers == [] and len(w.remainder) == 32 and w.options == [12, 2, 8, 8, 32]
    air = mini_verifier_air()
    positions = pv(raw, air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed)
    cpp = hostlib.HostAir(None, hostlib.AIR_MINI, 5)
    assert cv(cpp, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, raw) == positions
    t = verifier.replay_transcript(w, air, be.COIN_SOLIDITY, seed)
    assert t["positions"] == positions and t["fri_alphas"] == []
    args = (air.mask, 2, 1, be.TREE_KECCAK_M20, t["z"], t["deep_alpha"], [], positions)
    assert verifier.check_proof_data(w, *args) == positions
    w.remainder[1] = (w.remainder[1] + 1) % verifier.P
    with pytest.raises(verifier.VerificationError, match="is not the remainder's"):
        verifier.check_proof_data(w, *args)
    with pytest.raises(verifier.VerificationError, match="proof of work"):
        pv(w, air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed)
    with pytest.raises(SandstormHipError, match="proof of work"):
        cv(cpp, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, wire.serialize(w))
    cpp.close()
