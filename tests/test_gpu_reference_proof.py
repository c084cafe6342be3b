"""The product's proof IS the reference's proof.

`example/array-sum.proof.saved` (tests/golden/reference_array_sum_starknet.proof) is the reference's own proof of its
array-sum example under the starknet layout (EthVerifierClaim, 2^17 steps, 16 queries, 16 grinding bits).  The C++ host
here proves the same statement from the same run - trace.bin / memory.bin re-declared for the layout, base trace by
host/trace_starknet.cpp or by the device's generator (csrc/trace.hip), the real 195-constraint AIR (host/air_starknet.cpp), extension column by the device scans,
every stage a HIP kernel behind the C ABI - and emits the SAME BYTES: the three trace roots, all six FRI layer roots,
the 269 + 2 out-of-domain values, the remainder; and, given the reference's proof-of-work nonce (its grinder returns
whichever valid nonce its parallel search hits first - `find_any`, crypto/src/public_coin/solidity.rs:120-141 - ours
the smallest), the query openings and with them the entire file, byte for byte."""
import os

import pytest

from tests.test_layout_starknet import starknet_example

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.path.join(ROOT, "tests", "golden", "reference_array_sum_starknet.proof")


@pytest.fixture(scope="module", params=["host generator", "device generator"])
def statement(request):
    """the base trace by the host's generator (host/trace_starknet.cpp, uploaded) or made ON the device from the raw files' bytes
    (csrc/trace.hip through host/device_trace.hpp): the same proof bytes either way"""
    from sandstorm_amd import backend as be, binary, hostlib, public_input
    from sandstorm_amd.layouts import starknet as sk
    states, memory, spi = starknet_example(17)
    n = 1 << 21
    ctx = be.Context(0)
    if request.param == "host generator":
        cols = hostlib.starknet_base_trace(binary.write_register_states(states), binary.write_memory(memory), spi)
        assert cols[0].shape[0] == n
        base = be.Matrix.from_host(ctx, cols)
    else:
        import types
        base = types.SimpleNamespace(cols=hostlib.device_base_trace(ctx, "starknet", binary.write_register_states(states), binary.write_memory(memory), spi))
    air = hostlib.StarknetHostAir(ctx, spi, 21)
    seed = public_input.public_coin_seed(spi, be.COIN_SOLIDITY)
    keep = []

    def build_extension(challenges):
        m = hostlib.build_extension_columns(ctx, "starknet", [base.cols[sk.COL_NPC], base.cols[sk.COL_MEMORY], base.cols[sk.COL_RANGE_CHECK]],
                                            n, challenges)         # check=True: the permutation products must close
        keep.append(m)
        return m.cols

    def prove(options, **kw):
        return hostlib.prove(ctx, air, be.TREE_KECCAK_M20, 0, be.COIN_SOLIDITY, seed, base.cols, 21, build_extension, options, wire=True, **kw)
    yield prove, spi, seed
    for m in keep:
        m.close()
    air.close()
    ctx.close()


def test_cpp_host_reproduces_the_references_proof_byte_for_byte(statement):
    from sandstorm_amd import backend as be, verifier, wire
    from sandstorm_amd.layouts import starknet as sk
    from sandstorm_amd.prover import ProofOptions
    prove, spi, seed = statement
    with open(REFERENCE, "rb") as f:
        ref_raw = f.read()
    ref = wire.parse(ref_raw)
    opt = ProofOptions(*ref.options)                                  # 16 queries, blowup 2, 16 bits, fold 8, <= 16 coefficients
    # (a) our own grind: everything the transcript fixes before the proof of work is the reference's
    raw = prove(opt)
    ours = wire.parse(raw)
    assert ours.options == ref.options and ours.trace_len == ref.trace_len
    assert ours.base_root == ref.base_root, "base trace commitment"
    assert ours.extension_root == ref.extension_root, "extension trace commitment"
    assert ours.composition_root == ref.composition_root, "composition trace commitment"
    assert ours.ood_trace == ref.ood_trace and len(ref.ood_trace) == 269, "trace out-of-domain evaluations"
    assert ours.ood_composition == ref.ood_composition and len(ref.ood_composition) == 2, "composition out-of-domain evaluations"
    assert [l.root for l in ours.fri_layers] == [l.root for l in ref.fri_layers] and len(ref.fri_layers) == 6, "FRI layer commitments"
    assert ours.remainder == ref.remainder and len(ref.remainder) == 8, "FRI remainder"
    # the GPU grinder returns the SMALLEST valid nonce; the reference's is another valid one
    assert ours.pow_nonce <= ref.pow_nonce
    positions = verifier.verify(raw, sk.verifier_air(spi), be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, required_security_bits=32)
    assert len(positions) == len(ours.base_openings)
    # (b) with the reference's nonce the query positions coincide, and so does every remaining byte
    same = prove(opt, pow_nonce=ref.pow_nonce)
    assert len(same) == len(ref_raw)
    assert same == ref_raw, "first differing byte at offset %d" % next(i for i, (a, b) in enumerate(zip(same, ref_raw)) if a != b)
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "reproduced_array_sum_starknet.sha256"), "w") as f:
        import hashlib
        f.write("%s  reproduced on the GPU\n%s  tests/golden/reference_array_sum_starknet.proof\n" % (hashlib.sha256(same).hexdigest(), hashlib.sha256(ref_raw).hexdigest()))


def test_a_wrong_nonce_is_refused(statement):
    from sandstorm_amd._lib import SandstormHipError
    from sandstorm_amd.prover import ProofOptions
    prove, _, _ = statement
    with pytest.raises(SandstormHipError, match="nonce is not valid"):
        prove(ProofOptions(16, 2, 16, 8, 16), pow_nonce=12345)
