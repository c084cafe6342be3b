"""tests/golden/deep_pin_recursive.json: the out-of-domain point z and the DEEP coefficient alpha of the reference's shipped
recursive-layout proof (/root/reference/bootloader-proof.bin), RECOVERED FROM ITS DATA, plus everything needed to re-check
them without the reference: query positions, opened rows, out-of-domain vectors, first-FRI-layer values.

How z is found (no transcript, no public input needed): base columns 1 and 2 (diluted check) of a run without bitwise
instances do not depend on the program, so their trace polynomials T1, T2 are known (sandstorm_amd/layouts/recursive.py,
pinned by lde_offset_pin.json).  The proof's out-of-domain vector holds T2(z) and T2(z w) (mask cells (2,0), (2,1)), so z
is a common root of  T2(X) - ood[(2,0)]  and  T2(w X) - ood[(2,1)]:  their gcd (tests/golden/poly_gcd.c, classical Euclid on
degree-2^18 polynomials, ~6 min on 8 cores) is X - z.  All 33 out-of-domain values of columns 1 and 2 then equal
T_c(z w^offset): the OOD vector is in sorted (column, offset) order and evaluated at z w_n^offset.

How alpha is found: at a query point x_q the first FRI layer holds DEEP(x_q) = sum_j alpha^j (T_j(x_q) - ood_j) / (x_q - z w^o_j)
+ sum_k alpha^(133+k) (H_k(x_q) - oodc_k) / (x_q - z^2), every T_j(x_q), H_k(x_q) being an opened row value: a degree-134
polynomial in alpha per query; the gcd of two of them is linear, and its root satisfies all 40 queries — which pins the DEEP
term order and coefficients, the composition point z^2 (M5) and the whole OOD vector order.
Run in the build container: python tests/golden/make_deep_pin_golden.py [--z 0x...]   (--z skips the 6-minute gcd)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_py as oracle                         # noqa: E402
from sandstorm_amd import wire                                 # noqa: E402
from sandstorm_amd.layouts import recursive as rec             # noqa: E402
from sandstorm_amd.prover import bitrev                        # noqa: E402
from tests.test_layout_recursive import load_run               # noqa: E402

P = rec.P
with open("/root/reference/bootloader-proof.bin", "rb") as f:
    w = wire.parse(f.read())
with open(os.path.join(ROOT, "tests", "golden", "saved_proof_openings_recursive.json")) as f:
    positions = json.load(f)["positions"]
mask = rec.mask()
states, memory, pi = load_run()
cols = rec.base_trace(states, memory, pi)
n = len(cols[0])
assert n == w.trace_len and len(w.ood_trace) == len(mask) == 133
wn = pow(3, (P - 1) // n, P)

if "--z" in sys.argv:
    z = int(sys.argv[sys.argv.index("--z") + 1], 16)
else:
    work = "/tmp/deep_pin"
    os.makedirs(work, exist_ok=True)
    co = [int(v) for v in oracle.from_mont(oracle.ntt(oracle.to_mont(cols[2]), inverse=True))]
    d = list(co)
    d[0] = (d[0] - w.ood_trace[mask.index((2, 0))]) % P
    e, wk = [], 1
    for c in co:
        e.append(c * wk % P)
        wk = wk * wn % P
    e[0] = (e[0] - w.ood_trace[mask.index((2, 1))]) % P
    oracle.to_mont(d).tofile(work + "/D.bin")
    oracle.to_mont(e).tofile(work + "/E.bin")
    subprocess.check_call(["gcc", "-O3", "-fopenmp", "-o", work + "/poly_gcd", os.path.join(ROOT, "tests", "golden", "poly_gcd.c"),
                           "-I", os.path.join(ROOT, "oracle"), "-I", os.path.join(ROOT, "include"),
                           "-L", os.path.join(ROOT, "oracle", "_build"), "-loracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle", "_build")])
    subprocess.check_call([work + "/poly_gcd", work + "/D.bin", work + "/E.bin", work + "/G.bin"])
    gcd = [int(v) for v in oracle.from_mont(np.fromfile(work + "/G.bin", dtype=np.uint64).reshape(-1, 4))]
    assert len(gcd) == 2 and gcd[1] == 1, "no common root: the OOD order / evaluation-point assumption is wrong"
    z = (-gcd[0]) % P

# every out-of-domain value of the two known columns
for c in (1, 2):
    coeffs = oracle.ntt(oracle.to_mont(cols[c]), inverse=True)
    for j, (cc, o) in enumerate(mask):
        if cc == c:
            v = int(oracle.from_mont(oracle.poly_eval(coeffs, oracle.to_mont([z * pow(wn, o, P) % P])[0])[None])[0])
            assert v == w.ood_trace[j], (c, o)

# alpha from the DEEP relation at the query points
N = 2 * n
log_N = N.bit_length() - 1
w_N = pow(3, (P - 1) // N, P)
layer0 = w.fri_layers[0]
rows0 = sorted(set(q >> 3 for q in positions))
nq = len(positions)


def equation(qi, q):
    x = 3 * pow(w_N, bitrev(q, log_N), P) % P
    t = list(w.base_rows[7 * qi: 7 * qi + 7]) + list(w.extension_rows[3 * qi: 3 * qi + 3])
    c = w.composition_rows[2 * qi: 2 * qi + 2]
    a = [(t[col] - w.ood_trace[j]) * pow((x - z * pow(wn, o, P)) % P, -1, P) % P for j, (col, o) in enumerate(mask)]
    a += [(c[k] - w.ood_composition[k]) * pow((x - z * z) % P, -1, P) % P for k in range(2)]
    a[0] = (a[0] - layer0.rows[8 * rows0.index(q >> 3) + (q & 7)]) % P
    return a


def poly_gcd(a, b):
    def trim(p):
        while p and p[-1] == 0:
            p.pop()
        return p
    a, b = trim(a[:]), trim(b[:])
    while b:
        inv = pow(b[-1], -1, P)
        while len(a) >= len(b):
            q, sh = a[-1] * inv % P, len(a) - len(b)
            for i in range(len(b)):
                a[i + sh] = (a[i + sh] - q * b[i]) % P
            trim(a)
        a, b = b, a
    return a


eqs = [equation(qi, q) for qi, q in enumerate(positions)]
g = poly_gcd(eqs[0], eqs[1])
assert len(g) == 2, "no common alpha: the DEEP term order / composition point assumption is wrong"
alpha = (-g[0] * pow(g[1], -1, P)) % P
assert all(sum(c * pow(alpha, i, P) for i, c in enumerate(e)) % P == 0 for e in eqs)

out = {"file": "bootloader-proof.bin", "trace_len": n, "z": hex(z), "deep_alpha": hex(alpha), "positions": positions,
       "ood_trace": [hex(v) for v in w.ood_trace], "ood_composition": [hex(v) for v in w.ood_composition],
       "base_rows": [hex(v) for v in w.base_rows], "extension_rows": [hex(v) for v in w.extension_rows],
       "composition_rows": [hex(v) for v in w.composition_rows],
       "deep_values": [hex(layer0.rows[8 * rows0.index(q >> 3) + (q & 7)]) for q in positions]}
with open(os.path.join(ROOT, "tests", "golden", "deep_pin_recursive.json"), "w") as f:
    json.dump(out, f)
print("z", hex(z), "alpha", hex(alpha), "queries", nq)
