"""The host verifier (sandstorm_amd/verifier.py) on proofs in the reference's wire format.

CPU: the committed proofs tests/golden/mini_proof_eth_log{5,9}.bin (made on an MI355X by tests/golden/
make_mini_proof.py) verify, and do not after any single-section tampering.  GPU: a fresh proof from the C++ host."""
import json
import os

import pytest

from tests import mini_air

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


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
    positions = verifier.verify(raw, mini_verifier_air(), be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed)
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
            verifier.verify(w, air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed)

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
        verifier.verify(raw, air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, bytes(32))     # wrong public-coin seed
    with pytest.raises(verifier.VerificationError, match="malformed"):
        verifier.verify(raw[:-5], air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed)
    with pytest.raises(verifier.VerificationError):
        verifier.verify(raw, air, be.TREE_KECCAK, be.COIN_SOLIDITY, seed)              # unmasked tree: other hashes


def test_older_conventions_are_a_different_statement():
    """the same bytes do not verify under the older proof's conventions (natural order, normalised fold)"""
    from sandstorm_amd import backend as be, verifier
    from sandstorm_amd.prover import Conventions
    raw, seed, _ = load_fixture(5)
    with pytest.raises(verifier.VerificationError):
        verifier.verify(raw, mini_verifier_air(), be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed,
                        Conventions(bitrev_commit=False, fri_unnormalised=False, remainder_unshifted=False))


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
    assert len(verifier.verify(raw, mini_verifier_air(), be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed)) <= 20
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
    positions = verifier.verify(raw, rec.verifier_air(pi), be.TREE_KECCAK, be.COIN_SOLIDITY, seed)
    w = wire.parse(raw)
    assert w.trace_len == 16 * pi.n_steps and len(w.ood_trace) == 133 and len(positions) == len(w.base_openings)
    pi2 = copy.deepcopy(pi)
    pi2.rc_max += 1                                           # a different claim: other seed, other hints
    with pytest.raises(verifier.VerificationError):
        verifier.verify(raw, rec.verifier_air(pi2), be.TREE_KECCAK, be.COIN_SOLIDITY, public_input.public_coin_seed(pi2, be.COIN_SOLIDITY))
    # same seed, wrong hint: the transcript replays, the out-of-domain identity must catch it
    with pytest.raises(verifier.VerificationError, match="out-of-domain identity"):
        verifier.verify(raw, rec.verifier_air(pi2), be.TREE_KECCAK, be.COIN_SOLIDITY, seed)


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
        assert hostlib.verify(air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, raw) == \
            verifier.verify(raw, mini_verifier_air(), be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed)
        if log_n == 5:
            w = wire.parse(raw)
            w.base_rows[0] = (w.base_rows[0] + 1) % verifier.P
            with pytest.raises(SandstormHipError, match="base trace"):
                hostlib.verify(air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, wire.serialize(w))
            w = wire.parse(raw)
            w.fri_layers[0].rows[3] = (w.fri_layers[0].rows[3] + 1) % verifier.P
            with pytest.raises(SandstormHipError, match="FRI layer 0"):
                hostlib.verify(air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, wire.serialize(w))
            w = wire.parse(raw)
            w.ood_composition[1] = (w.ood_composition[1] + 1) % verifier.P
            with pytest.raises(SandstormHipError):
                hostlib.verify(air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, wire.serialize(w))
            with pytest.raises(SandstormHipError, match="malformed"):
                hostlib.verify(air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, raw[:-3])
            with pytest.raises(SandstormHipError):
                hostlib.verify(air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, raw, shipped_conventions=False)
        air.close()
    pi = public_input.AirPublicInput.from_json(os.path.join(GOLD, "air_public_input_array_sum.json"))
    with open(os.path.join(GOLD, "array_sum_recursive_eth.proof"), "rb") as f:
        raw = f.read()
    seed = public_input.public_coin_seed(pi, be.COIN_SOLIDITY)
    air = hostlib.RecursiveHostAir(None, pi, 18)
    assert hostlib.verify(air, be.TREE_KECCAK, be.COIN_SOLIDITY, seed, raw) == \
        verifier.verify(raw, rec.verifier_air(pi), be.TREE_KECCAK, be.COIN_SOLIDITY, seed)
    air.close()
    pi2 = copy.deepcopy(pi)
    pi2.rc_max += 1
    air2 = hostlib.RecursiveHostAir(None, pi2, 18)
    with pytest.raises(SandstormHipError, match="out-of-domain identity"):
        hostlib.verify(air2, be.TREE_KECCAK, be.COIN_SOLIDITY, seed, raw)              # same transcript, wrong hint
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
            verifier.verify(b, air_py, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed)
            outcomes.append("ok")
        except verifier.VerificationError:
            outcomes.append("rejected")
        try:
            hostlib.verify(air_cpp, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, b)
            outcomes.append("ok")
        except SandstormHipError:
            outcomes.append("rejected")
        assert outcomes[0] == outcomes[1], (it, kind, outcomes)
        accepted += outcomes[0] == "ok"
    assert accepted == 0
    air_cpp.close()
