"""Fingerprints of the reference's periodic-column polynomials (data; run in the build container where /root/reference
is mounted): sha256 over the decimal coefficients, lowest degree first, joined by commas.  The layout restatement
derives the same polynomials from first principles (doublings of the curve points, the Hades round constants) and
tests/test_layout_starknet.py compares the fingerprints."""
import hashlib
import json
import os
import re

R = "/root/reference/builtins/src/"
ARRAYS = [("pedersen_x", "pedersen/periodic.rs", "HASH_POINTS_X_COEFFS"), ("pedersen_y", "pedersen/periodic.rs", "HASH_POINTS_Y_COEFFS"),
          ("ecdsa_generator_x", "ecdsa/periodic.rs", "GENERATOR_POINTS_X_COEFFS"), ("ecdsa_generator_y", "ecdsa/periodic.rs", "GENERATOR_POINTS_Y_COEFFS"),
          ("poseidon_full_key0", "poseidon/periodic.rs", "FULL_ROUND_KEY_0_COEFFS"), ("poseidon_full_key1", "poseidon/periodic.rs", "FULL_ROUND_KEY_1_COEFFS"),
          ("poseidon_full_key2", "poseidon/periodic.rs", "FULL_ROUND_KEY_2_COEFFS"), ("poseidon_partial_key0", "poseidon/periodic.rs", "PARTIAL_ROUND_KEY_0_COEFFS"),
          ("poseidon_partial_key1", "poseidon/periodic.rs", "PARTIAL_ROUND_KEY_1_COEFFS")]


def coefficients(path, name):
    s = open(R + path).read()
    i = s.index("pub const " + name)
    return [int(x) for x in re.findall(r'Fp!\("(\d+)"\)', s[i:s.index("];", i)])]


out = {}
for key, path, name in ARRAYS:
    c = coefficients(path, name)
    out[key] = {"count": len(c), "sha256": hashlib.sha256(",".join(str(v) for v in c).encode()).hexdigest()}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "starknet_periodic_fingerprints.json"), "w") as f:
    json.dump(out, f, indent=1)
print(out)
