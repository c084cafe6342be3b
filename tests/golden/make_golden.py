#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the reference's DATA.

Run once, in the build container, where /root/reference is mounted (it does not
exist on the GPU box).  Nothing from the reference is executed: the script
reads constant tables / expected values out of the reference's own known-answer
tests and restates each test's expected side with Python big integers:

  ntt_pedersen512.json   builtins/src/pedersen/periodic.rs:1183-1250
                         (periodic_{x,y}_evals_match: fft(COEFFS) == doubling chain)
  ntt_ecdsa256.json      builtins/src/ecdsa/periodic.rs:600-638
  pedersen.json          builtins/src/pedersen/mod.rs:183-211 (hash_example0/1_works)
                         + builtins/src/pedersen/constants.rs:2064-2093 (constant_points)
  coins.json             crypto/src/public_coin/solidity.rs:172-193,
                         crypto/src/public_coin/cairo.rs:189-208
  montgomery.json        crypto/src/utils.rs:19-20 (the MONTGOMERY_R comment)
  ntt_poseidon8.json     builtins/src/poseidon/periodic.rs:241-290 (full_round_keys{0,1,2}_match:
                         fft(FULL_ROUND_KEY_k_COEFFS) == the rotated round-key halves)
  saved_proofs.json      header + out-of-domain tail of the three proof FILES the reference ships
                         (example/array-sum.proof.saved, example/bootloader/bootloader-proof.bin,
                         bootloader-proof.bin): options, trace length, base-trace root, the trace-OOD and
                         composition-OOD vectors - data for the OOD-identity pin

Every file stores field elements as decimal strings of the canonical value.
"""
import json
import os
import re
import sys

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
P = 2**251 + 17 * 2**192 + 1
BETA = 3141592653589793238462643383279502884197169399375105820974944592307816406665


def const_array(path, name):
    src = open(os.path.join(REF, path)).read()
    m = re.search(r"pub const %s: \[Fp; (\d+)\] = \[(.*?)\];" % name, src, re.S)
    vals = [int(v) for v in re.findall(r'Fp!\("(\d+)"\)', m.group(2))]
    assert len(vals) == int(m.group(1)), (name, len(vals))
    return vals


def const_point(path, name):
    src = open(os.path.join(REF, path)).read()
    m = re.search(r"pub const %s: Affine<StarkwareCurve> = Affine::new_unchecked\(\s*"
                  r'Fp!\("(\d+)"\),\s*Fp!\("(\d+)"\),' % name, src)
    return int(m.group(1)), int(m.group(2))


def ec_double(pt):
    x, y = pt
    lam = (3 * x * x + 1) * pow(2 * y, -1, P) % P
    x3 = (lam * lam - 2 * x) % P
    return x3, (lam * (x - x3) - y) % P


def ec_add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    (x1, y1), (x2, y2) = p1, p2
    if x1 == x2:
        return ec_double(p1) if y1 == y2 else None
    lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return x3, (lam * (x1 - x3) - y1) % P


def doubling_chain(pt, count):
    out = []
    for _ in range(count):
        out.append(pt)
        pt = ec_double(pt)
    return out


def eval_domain(coeffs):
    """naive P(w^k), k < n, w = 3^((p-1)/n)  (natural order)."""
    n = len(coeffs)
    w = pow(3, (P - 1) // n, P)
    out = []
    for k in range(n):
        x = pow(w, k, P)
        acc = 0
        for c in reversed(coeffs):
            acc = (acc * x + c) % P
        out.append(acc)
    return out


def dump(name, obj):
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print("wrote", name, os.path.getsize(os.path.join(OUT, name)), "bytes")


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not mounted; fixtures are already committed")
    s = lambda vs: [str(v) for v in vs]

    # ---- Pedersen periodic columns: a 512-point NTT known-answer test
    p0, p1, p2, p3, p4 = (const_point("builtins/src/pedersen/constants.rs", "P%d" % i) for i in range(5))
    for pt in (p0, p1, p2, p3, p4):
        assert (pt[1] ** 2 - (pt[0] ** 3 + pt[0] + BETA)) % P == 0
    ev = doubling_chain(p1, 248) + doubling_chain(p2, 4)
    ev += [ev[-1]] * 4
    ev += doubling_chain(p3, 248) + doubling_chain(p4, 4)
    ev += [ev[-1]] * 4
    assert len(ev) == 512
    cx = const_array("builtins/src/pedersen/periodic.rs", "HASH_POINTS_X_COEFFS")
    cy = const_array("builtins/src/pedersen/periodic.rs", "HASH_POINTS_Y_COEFFS")
    assert eval_domain(cx) == [e[0] for e in ev], "reference KAT does not reproduce"
    assert eval_domain(cy) == [e[1] for e in ev], "reference KAT does not reproduce"
    dump("ntt_pedersen512.json", {"coeffs_x": s(cx), "coeffs_y": s(cy),
                                  "evals_x": s(e[0] for e in ev), "evals_y": s(e[1] for e in ev)})

    # ---- ECDSA generator doublings: a 256-point NTT known-answer test
    src = open(os.path.join(REF, "builtins/src/utils.rs")).read()
    m = re.search(r'const GENERATOR: Affine<Self> = Affine::new_unchecked\(\s*Fp!\("(\d+)"\),\s*Fp!\("(\d+)"\)', src)
    gen = (int(m.group(1)), int(m.group(2)))
    ev = doubling_chain(gen, 251)
    ev += [ev[-1]] * 5
    gx = const_array("builtins/src/ecdsa/periodic.rs", "GENERATOR_POINTS_X_COEFFS")
    gy = const_array("builtins/src/ecdsa/periodic.rs", "GENERATOR_POINTS_Y_COEFFS")
    assert eval_domain(gx) == [e[0] for e in ev]
    assert eval_domain(gy) == [e[1] for e in ev]
    dump("ntt_ecdsa256.json", {"coeffs_x": s(gx), "coeffs_y": s(gy),
                               "evals_x": s(e[0] for e in ev), "evals_y": s(e[1] for e in ev)})

    # ---- Pedersen hash examples (+ the P1..P4 doubling-table spot checks)
    src = open(os.path.join(REF, "builtins/src/pedersen/mod.rs")).read()
    tests = re.findall(r'let a = Fp!\("(\d+)"\);\s*let b = Fp!\("(\d+)"\);.*?assert_eq!\(\s*Fp!\("(\d+)"\)', src, re.S)
    assert len(tests) == 2

    def pedersen(a, b):
        acc = p0
        for val, lo, hi in ((a, p1, p2), (b, p3, p4)):
            low, high = val & (2**248 - 1), val >> 248
            for bits, base in ((low, lo), (high, hi)):
                pt = base
                while bits:
                    if bits & 1:
                        acc = ec_add(acc, pt)
                    pt = ec_double(pt)
                    bits >>= 1
        return acc[0]
    for a, b, h in tests:
        assert pedersen(int(a), int(b)) == int(h), "reference KAT does not reproduce"
    dump("pedersen.json", {"points": {"P%d" % i: s(pt) for i, pt in enumerate((p0, p1, p2, p3, p4))},
                           "hash_examples": [{"a": a, "b": b, "hash": h} for a, b, h in tests],
                           # extra vectors from the restated definition (edge cases)
                           "extra": [{"a": str(a), "b": str(b), "hash": str(pedersen(a, b))}
                                     for a, b in ((0, 0), (1, 0), (0, 1), (P - 1, P - 1), (2**248, 2**251),
                                                  (2**160 - 1, 2**160 - 2))]})

    # ---- coins
    src = open(os.path.join(REF, "crypto/src/public_coin/solidity.rs")).read()
    draws = re.findall(r'Fp!\("(\d+)"\),\s*public_coin\.draw\(\)', src)
    assert len(draws) == 4
    src = open(os.path.join(REF, "crypto/src/public_coin/cairo.rs")).read()
    t = src[src.index("fn reseed_with_field_element()"):]
    arrays = re.findall(r"\[\s*((?:0x[0-9a-f]{2},\s*)+0x[0-9a-f]{2},?)\s*\]", t)
    seed, expected = (bytes(int(b, 16) for b in re.findall(r"0x([0-9a-f]{2})", a)) for a in arrays[:2])
    elem = re.search(r'Fp!\("(\d+)"\)', t).group(1)
    assert len(seed) == 32 and len(expected) == 32
    import hashlib
    d = (int.from_bytes(seed, "big") + 1).to_bytes(32, "big") + int(elem).to_bytes(32, "big")
    assert hashlib.blake2s(d).digest() == expected, "reference KAT does not reproduce"
    dump("coins.json", {"solidity_zero_seed_draws": draws,
                        "cairo_reseed": {"seed": seed.hex(), "element": elem, "digest": expected.hex()}})

    # ---- Poseidon round keys: three 8-point NTT known-answer tests
    src = open(os.path.join(REF, "builtins/src/poseidon/params.rs")).read()

    def key_table(name):
        m = re.search(r"pub const %s: \[\[Fp; 3\]; NUM_FULL_ROUNDS / 2\] = \[(.*?)\n\];" % name, src, re.S)
        vals = [int(v) for v in re.findall(r'Fp!\("(\d+)"\)', m.group(1))]
        assert len(vals) == 12
        return [vals[3 * i:3 * i + 3] for i in range(4)]
    h1, h2 = key_table("FULL_ROUND_KEYS_1ST_HALF"), key_table("FULL_ROUND_KEYS_2ND_HALF")
    pos = {}
    for k in range(3):
        a, b = [r[k] for r in h1], [r[k] for r in h2]
        a, b = a[1:] + [0], b[1:] + [0]                  # rotate_left(1), last = 0
        co = const_array("builtins/src/poseidon/periodic.rs", "FULL_ROUND_KEY_%d_COEFFS" % k)
        assert eval_domain(co) == a + b, "reference KAT does not reproduce"
        pos["key%d" % k] = {"coeffs": s(co), "evals": s(a + b)}
    dump("ntt_poseidon8.json", pos)

    # ---- saved proofs: header and OOD tail (ark-serialize compressed)
    def parse_proof(path):
        raw = open(os.path.join(REF, path), "rb").read()
        opts = list(raw[:5])
        trace_len = int.from_bytes(raw[5:13], "little")
        assert int.from_bytes(raw[13:21], "little") == 32
        root = raw[21:53]
        # tail: Vec<Fp> trace OOD (u64 len, 32-byte LE canonical each) then Vec<Fp> composition OOD (len 2), EOF
        assert int.from_bytes(raw[-72:-64], "little") == 2
        comp = [int.from_bytes(raw[-64 + 32 * i:len(raw) - 32 + 32 * i] if i == 0 else raw[-32:], "little") for i in range(2)]
        for n_ood in (269, 133):
            off = len(raw) - 72 - 32 * n_ood - 8
            if off > 0 and int.from_bytes(raw[off:off + 8], "little") == n_ood:
                break
        else:
            raise AssertionError("no OOD vector found in " + path)
        ood = [int.from_bytes(raw[off + 8 + 32 * i:off + 40 + 32 * i], "little") for i in range(n_ood)]
        assert all(v < P for v in ood + comp)
        return {"file": path, "bytes": len(raw), "options": {"num_queries": opts[0], "lde_blowup_factor": opts[1],
                "grinding_factor": opts[2], "fri_folding_factor": opts[3], "fri_max_remainder_coeffs": opts[4]},
                "trace_len": trace_len, "base_trace_root": root.hex(), "ood_trace": s(ood), "ood_composition": s(comp)}
    dump("saved_proofs.json", [parse_proof(f) for f in ("example/array-sum.proof.saved",
                                                        "example/bootloader/bootloader-proof.bin", "bootloader-proof.bin")])

    # ---- Montgomery constants
    src = open(os.path.join(REF, "crypto/src/utils.rs")).read()
    r = re.search(r'Fp!\("(\d+)"\)', src).group(1)
    assert int(r) == 2**256 % P
    dump("montgomery.json", {"modulus": str(P), "R_mod_p": r, "inv64": str((-pow(P, -1, 2**64)) % 2**64)})


if __name__ == "__main__":
    main()
