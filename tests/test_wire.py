"""The reference's proof wire format (sandstorm_amd/wire.py)."""
import os

import numpy as np
import pytest

from sandstorm_amd import wire

REF_PROOFS = ["/root/reference/example/array-sum.proof.saved", "/root/reference/bootloader-proof.bin"]


@pytest.mark.parametrize("path", REF_PROOFS)
def test_reference_proofs_round_trip(path):
    """parse -> serialize reproduces the reference's shipped proof files byte for byte (the files are not copied
    into this repo; the test runs where /root/reference is mounted)."""
    if not os.path.exists(path):
        pytest.skip("reference not mounted")
    raw = open(path, "rb").read()
    p = wire.parse(raw)
    assert wire.serialize(p) == raw
    assert p.options[1] == 2 and p.options[3] == 8 and len(p.ood_composition) == 2
    assert len(p.base_openings) == p.options[0] == len(p.composition_openings)
    assert all(o.variant == 0 for o in p.base_openings)


def test_parse_matches_golden_fixture(golden):
    path = REF_PROOFS[0]
    if not os.path.exists(path):
        pytest.skip("reference not mounted")
    p = wire.parse(open(path, "rb").read())
    g = golden("saved_proof_openings.json")
    assert p.base_root.hex() == g["roots"]["base"] and p.extension_root.hex() == g["roots"]["extension"]
    assert [l.root.hex() for l in p.fri_layers] == g["roots"]["fri_layers"]
    assert p.pow_nonce == g["pow_nonce"] and ["%x" % v for v in p.remainder] == g["remainder"]
    q0 = g["queries"][0]
    assert ["%x" % v for v in p.base_rows[:9]] == q0["base"]["row"]
    assert [p.base_openings[0].sibling.hex()] + [d.hex() for d in p.base_openings[0].path] == q0["base"]["path"]
    assert p.extension_openings[0].variant == 1 and "%x" % p.extension_openings[0].leaf == q0["extension"]["leaf"]


def test_rejects_malformed():
    with pytest.raises(ValueError):
        wire.parse(bytes(5) + (8).to_bytes(8, "little") + (31).to_bytes(8, "little") + bytes(64))
    with pytest.raises((ValueError, IndexError)):
        wire.parse(b"\x10\x02\x10\x08\x10")


def test_montgomery_conversion_round_trip():
    for x in (0, 1, 2, wire.P - 1, 2**200 + 12345):
        assert wire._canon(np.array(wire._mont_limbs(x), dtype=np.uint64)) == x
