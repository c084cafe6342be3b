#!/usr/bin/env python3
"""Pins the reference's FRI fold convention from the proof FILES it ships (data only).

The saved proofs carry, per FRI layer, the queried rows as a flat Vec<Fp> of
`queries x 8` values.  The verifier's relation between consecutive layers - the fold of a row at the
layer's random alpha is one entry of the matching row of the next layer - can be tested WITHOUT knowing
alpha or the query positions:

  a row holds f on a coset {x w8^e};  P = its degree-<8 interpolant;  fold = c * P(alpha)  =  c * F(beta),
  F(b) = sum_k chat_k b^k,  chat = inverse DFT of the row,  beta = alpha / x.

For a candidate next-layer value v the roots beta of c F(b) - v are found (x^p - x gcd + equal-degree
splitting); the TRUE roots of all rows of a layer share  beta^L = (alpha / offset)^L  because every x is
offset times an L-th root of unity.  A hypothesis (in-row order, normalisation c) is right iff every row of
every layer has a root with the common beta^L - random data matches at most one row.

Result:
  example/array-sum.proof.saved (all 16 rows of all 5 layer pairs) and bootloader-proof.bin - the two files
  with masked-20 digests, i.e. the current code path:
  * in-row order is BIT-REVERSED:  stored[j] = f(x * w8^bitrev3(j))  - rows are 8 adjacent entries of an
    evaluation vector kept in bit-reversed order;
  * the fold is UNNORMALISED:  next = 8 * P(alpha)  (StarkWare's convention: no 1/2 per binary fold);
  * row r of a layer folds into row r >> 3, slot r & 7 of the next one (positions are indices into the
    bit-reversed vectors).
  example/bootloader/bootloader-proof.bin (unmasked digests: an OLDER code path) follows the other convention:
  natural in-row order (row j = evals[j + k L/8]) and the normalised interpolate-and-evaluate fold.
Output: fri_saved_proofs.json = per (file, layer) the rows, the matching next-layer value and beta, i.e.
known-answer vectors  fold(row; alpha = beta, offset = 1) = next  for ss_fri_fold / or_fri_fold.

Run once in the build container (needs /root/reference; ~5 minutes).  Nothing from the reference is executed.
"""
import json
import os
import random
import sys

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
P = 2**251 + 17 * 2**192 + 1
FOLD = 8
W8 = pow(3, (P - 1) // 8, P)
W8I = pow(W8, -1, P)
INV8 = pow(8, -1, P)
BITREV3 = [0, 4, 2, 6, 1, 5, 3, 7]
BIG_L = 1 << 30                      # any multiple of the layer's domain size


# ---- polynomials over F_p, coefficient lists low -> high
def pmod(a, m):
    a = a[:]
    dm = len(m) - 1
    inv = pow(m[-1], -1, P)
    while len(a) - 1 >= dm and any(a):
        if a[-1] == 0:
            a.pop()
            continue
        c = a[-1] * inv % P
        s = len(a) - 1 - dm
        for i in range(dm + 1):
            a[s + i] = (a[s + i] - c * m[i]) % P
        a.pop()
    while a and a[-1] == 0:
        a.pop()
    return a


def pmul(a, b, m):
    if not a or not b:
        return []
    r = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                r[i + j] = (r[i + j] + x * y) % P
    return pmod(r, m)


def pgcd(a, b):
    while b:
        a, b = b, pmod(a, b)
    inv = pow(a[-1], -1, P)
    return [c * inv % P for c in a]


def ppow(base, e, m):
    r = [1]
    base = pmod(base, m)
    while e:
        if e & 1:
            r = pmul(r, base, m)
        base = pmul(base, base, m)
        e >>= 1
    return r


def roots(f):
    f = f[:]
    while f and f[-1] == 0:
        f.pop()
    if len(f) <= 1:
        return []
    g = ppow([0, 1], P, f)
    g = g + [0] * (2 - len(g))
    g[1] = (g[1] - 1) % P
    while g and g[-1] == 0:
        g.pop()
    h = pgcd(f, g) if g else f
    res = []

    def split(h):
        d = len(h) - 1
        if d == 0:
            return
        if d == 1:
            res.append((-h[0]) * pow(h[1], -1, P) % P)
            return
        while True:
            r = ppow([random.randrange(P), 1], (P - 1) // 2, h)
            r = r or [0]
            r[0] = (r[0] - 1) % P
            while r and r[-1] == 0:
                r.pop()
            if not r:
                continue
            g = pgcd(h, r)
            dg = len(g) - 1
            if 0 < dg < d:
                rem, q = h[:], [0] * (d - dg + 1)
                for i in range(d - dg, -1, -1):
                    c = rem[i + dg]
                    q[i] = c
                    for j in range(dg + 1):
                        rem[i + j] = (rem[i + j] - c * g[j]) % P
                split(g)
                split(q)
                return
    split(h)
    return res


def fold_poly(row, bitrev):
    """chat of a row; P(alpha) = sum chat_k (alpha/x)^k.  bitrev: stored[j] = f(x w8^bitrev3(j))"""
    nat = [row[BITREV3[k]] for k in range(8)] if bitrev else list(row)      # nat[k] = f(x w8^k)
    return [sum(nat[j] * pow(W8I, j * k, P) for j in range(8)) * INV8 % P for k in range(8)]


def find_layers(raw):
    """flat Vec<Fp> blocks: u64 LE length n (multiple of 8), then n canonical 32-byte LE values"""
    out, pos = [], 0
    while pos < len(raw) - 8:
        n = int.from_bytes(raw[pos:pos + 8], "little")
        if 8 <= n <= 4000 and n % 8 == 0 and pos + 8 + 32 * n <= len(raw):
            vals = [int.from_bytes(raw[pos + 8 + 32 * k:pos + 40 + 32 * k], "little") for k in range(n)]
            if all(2**200 < v < P for v in vals):
                out.append(vals)
                pos += 8 + 32 * n
                continue
        pos += 1
    return out


CONVENTIONS = {                      # name: (in-row order bit-reversed, scale c in  next = c * P(alpha))
    "bitrev_unnormalised": (True, 8),        # StarkWare: rows = 8 adjacent entries of a bit-reversed vector, no 1/2 per fold
    "natural_normalised": (False, 1),        # Winterfell-style: row j = evals[j + k L/8], interpolate and evaluate
}


def match_pair(a, b, conv):
    """-> [(row, next_index, beta)] for every row of layer a, or None if the hypothesis fails"""
    bitrev, scale = CONVENTIONS[conv]
    rows_a, nb = len(a) // 8, len(b)
    cand = {}
    for r in range(rows_a):
        chat = fold_poly(a[8 * r:8 * r + 8], bitrev)
        # without collisions row r lands in next-layer row r; otherwise try every entry
        tries = list(range(8 * r, 8 * r + 8)) if 8 * rows_a == nb else list(range(nb))
        for bi in tries:
            f = [scale * c % P for c in chat]
            f[0] = (f[0] - b[bi]) % P
            for beta in roots(f):
                cand.setdefault(pow(beta, BIG_L, P), []).append((r, bi, beta))
    t, best = max(cand.items(), key=lambda kv: len({e[0] for e in kv[1]}))
    if len({e[0] for e in best}) != rows_a:
        return None
    return sorted(best)


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not mounted; fixtures are already committed")
    random.seed(1)
    h = lambda v: "%x" % v
    out = []
    for path, max_rows in (("example/array-sum.proof.saved", 16), ("bootloader-proof.bin", 8),
                           ("example/bootloader/bootloader-proof.bin", 8)):
        raw = open(os.path.join(REF, path), "rb").read()
        layers = [l for l in find_layers(raw)]
        # FRI layers are the leading run of blocks with non-increasing length (then remainder, query rows, OOD)
        nq = len(layers[0])
        fri = []
        for l in layers:
            if len(l) <= nq and (not fri or len(l) <= len(fri[-1])) and len(l) >= 64:
                fri.append(l)
            else:
                break
        print(path, "FRI layer blocks:", [len(l) for l in fri], flush=True)
        conv = None
        for name in CONVENTIONS:         # which convention produced this file: decided on its first 4 rows
            if match_pair(fri[0][:32], fri[1], name) is not None:
                conv = name
                break
        assert conv, "no known convention explains " + path
        print("  convention:", conv, flush=True)
        for li in range(len(fri) - 1):
            a, b = fri[li][:8 * max_rows], fri[li + 1]
            m = match_pair(a, b, conv)
            if m is None and li == len(fri) - 2:
                # the last block of the run is the remainder polynomial's coefficients, not a layer
                print("  block %d (%d values) is not a layer of rows: remainder" % (li + 1, len(b)), flush=True)
                continue
            assert m is not None, "convention does not hold for %s layer %d" % (path, li)
            print("  layer %d -> %d: all %d rows match" % (li, li + 1, len(a) // 8), flush=True)
            seen = set()
            for r, bi, beta in m:
                if r in seen:
                    continue
                seen.add(r)
                out.append({"file": path, "convention": conv, "layer": li, "row": r, "next_row": bi >> 3, "next_slot": bi & 7,
                            "values": [h(v) for v in a[8 * r:8 * r + 8]], "beta": h(beta), "next": h(b[bi])})
    with open(os.path.join(OUT, "fri_saved_proofs.json"), "w") as f:
        json.dump({"fold": 8, "encoding": "hex canonical", "conventions": {k: {"bitrev_rows": v[0], "scale": v[1]}
                                                                         for k, v in CONVENTIONS.items()},
                   "vectors": out}, f, separators=(",", ":"))
    print("wrote fri_saved_proofs.json", len(out), "vectors", os.path.getsize(os.path.join(OUT, "fri_saved_proofs.json")), "bytes")


if __name__ == "__main__":
    main()
