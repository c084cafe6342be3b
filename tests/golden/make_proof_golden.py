#!/usr/bin/env python3
"""Parses example/array-sum.proof.saved completely (the ark-serialize wire format of the proof, to EOF) and
verifies every Merkle opening in it against the roots with THIS repo's hash conventions - data only, nothing
of the reference runs.  What holds (all 16 queries; base, extension and composition traces; all 6 FRI layers):

  * leaf of a multi-column matrix = Keccak-256 of the row's elements as 32-byte big-endian Montgomery images,
    masked to its first 20 bytes (crypto/src/merkle/utils.rs:19-46, hash/keccak.rs:50-58, hash/mod.rs:5-13);
  * single-column matrix (extension trace): proof variant 1, leaves are the raw elements and the first layer is
    hash_elements([l0, l1]) (crypto/src/merkle/mod.rs:419-437);
  * inner node = masked Keccak(left || right); children of node k are 2k, 2k+1; paths are listed bottom-up;
  * a query position p (22 bits) opens row p of the three trace trees, row p >> 3 of FRI layer 0 (slot p & 7),
    row p >> 6 of layer 1, ... - the trace LDE and the FRI evaluation vectors share one index space;
  * the remainder polynomial (8 coefficients here) interpolates the fold of the last layer over the UNSHIFTED
    domain: 8 P_r(alpha) = R(w'^bitrev(r)) has one common alpha/offset for all rows, none when the coset offset is
    put inside R's argument (so the LDE offset itself cannot be read off the FRI data);
  * that index space is BIT-REVERSED: with beta = alpha / x recovered per row by make_fri_golden.py,
    beta * w_L^bitrev(row) is one constant (alpha / offset) for all 16 rows of a layer, while the natural map
    gives 16 different values: committed index i <-> point offset * w_L^bitrev(i), w_L = 3^((p-1)/L).

Wire format (little-endian; digests and vectors are u64-length-prefixed):
  5 x u8 options | u64 trace_len | base root | 0x01 + extension root | composition root |
  u64 #layers x { Vec<Fp> flattened rows | u64 #proofs x proof | layer root } | Vec<Fp> remainder | u64 pow nonce |
  Vec<Fp> base rows | Vec<Fp> extension rows | Vec<Fp> composition rows | proofs base | proofs ext | proofs comp |
  Vec<Fp> trace OOD | Vec<Fp> composition OOD
  proof = u8 variant (0 hashed leaves: ... sibling digest, leaf digest; 1 unhashed: ... sibling Fp, leaf Fp) after
          Vec<digest> path (bottom-up, without the sibling)
Output: saved_proof_openings.json (first 4 queries in full + every root), consumed by tests/test_oracle_golden.py
and tests/test_gpu_parity.py.  Needs /root/reference, the built oracle and fri_saved_proofs.json.
"""
import ctypes as C
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle_py as oracle      # noqa: E402

REF_DIR = "/root/reference"
FILES = [("example/array-sum.proof.saved", "saved_proof_openings.json"),
         ("bootloader-proof.bin", "saved_proof_openings_recursive.json")]      # starknet shape (9+1 columns); recursive shape (7+3)
P = 2**251 + 17 * 2**192 + 1
KECCAK_M20 = 1


def main():
    if not os.path.isdir(REF_DIR):
        sys.exit("reference not mounted; fixtures are already committed")
    for rel, out_name in FILES:
        one(rel, out_name)


def one(rel, out_name):
    print("====", rel)
    raw = open(os.path.join(REF_DIR, rel), "rb").read()
    u64 = lambda o: int.from_bytes(raw[o:o + 8], "little")

    def vec(o):
        n = u64(o)
        return [int.from_bytes(raw[o + 8 + 32 * k:o + 40 + 32 * k], "little") for k in range(n)], o + 8 + 32 * n

    def digest(o):
        assert u64(o) == 32
        return raw[o + 8:o + 40], o + 40

    def proofs(o):
        cnt = u64(o)
        o += 8
        recs = []
        for _ in range(cnt):
            tag = raw[o]
            o += 1
            n = u64(o)
            o += 8
            nodes = []
            for _k in range(n):
                d, o = digest(o)
                nodes.append(d)
            if tag == 0:
                sib, o = digest(o)
                leaf, o = digest(o)
            else:
                sib = int.from_bytes(raw[o:o + 32], "little")
                leaf = int.from_bytes(raw[o + 32:o + 64], "little")
                o += 64
            recs.append((tag, nodes, sib, leaf))
        return recs, o

    lib = oracle.lib()

    def merge(a, b):
        out = (C.c_uint8 * 32)()
        lib.or_hash_merge(C.c_int(KECCAK_M20), (C.c_uint8 * 32).from_buffer_copy(a), (C.c_uint8 * 32).from_buffer_copy(b), out)
        return bytes(out)

    def rowhash(vals):
        m = oracle.to_mont(vals)
        return bytes(oracle.hash_rows(KECCAK_M20, [m[k:k + 1] for k in range(len(vals))])[0])

    def climb(cur, sibs, pos):
        for lvl, s in enumerate(sibs):
            cur = merge(cur, s) if ((pos >> lvl) & 1) == 0 else merge(s, cur)
        return cur

    opts = list(raw[:5])
    trace_len = u64(5)
    o = 13
    base_root, o = digest(o)
    assert raw[o] == 1
    ext_root, o = digest(o + 1)
    comp_root, o = digest(o)
    nl = u64(o)
    o += 8
    layers = []
    for _ in range(nl):
        vals, o = vec(o)
        pr, o = proofs(o)
        root, o = digest(o)
        layers.append((vals, pr, root))
    remainder, o = vec(o)
    nonce = u64(o)
    o += 8
    base_rows, o = vec(o)
    ext_rows, o = vec(o)
    comp_rows, o = vec(o)
    base_pr, o = proofs(o)
    ext_pr, o = proofs(o)
    comp_pr, o = proofs(o)
    ood, o = vec(o)
    oodc, o = vec(o)
    assert o == len(raw), "wire format not consumed to EOF"
    nq = len(base_pr)
    ncb, nce = len(base_rows) // nq, len(ext_rows) // nq
    print("parsed to EOF: options", opts, "trace_len", trace_len, "queries", nq, "base cols", ncb, "ext cols", nce, "layers", nl)
    # slot of each layer-i row inside its layer-(i+1) row and beta = alpha / x per row: the fold pin, on every row
    sys.path.insert(0, HERE)
    import make_fri_golden as G
    nxt = {}
    for li in range(nl - 1):
        m = G.match_pair(layers[li][0], layers[li + 1][0], "bitrev_unnormalised")
        assert m is not None, "fold convention does not hold on layer %d" % li
        for r, bi, beta in m:
            nxt.setdefault((li, r), (bi >> 3, bi & 7, beta))
    log_rows0 = (trace_len * opts[1]).bit_length() - 1 - 3
    top_bits = log_rows0 - 3 * (nl - 1)
    positions = []
    for q in range(nq):
        r, sl = q, []
        for li in range(nl - 1):
            r, s, _b = nxt[(li, r)]
            sl.append(s)
        vals, pr, root = layers[0]
        leaf = rowhash(vals[8 * q:8 * q + 8])
        assert leaf == pr[q][3]
        hit = None
        for top in range(1 << top_bits):  # the position bits above the recovered slots
            r0 = 0
            for s in reversed(sl):
                r0 = r0 * 8 + s
            r0 += top << (3 * (nl - 1))
            if climb(leaf, [pr[q][2]] + pr[q][1], r0) == root:
                hit = r0
        assert hit is not None, "FRI layer 0 path of query %d does not verify" % q
        bleaf = rowhash(base_rows[ncb * q:ncb * q + ncb])
        assert bleaf == base_pr[q][3]
        p = None
        for s in range(8):
            if climb(bleaf, [base_pr[q][2]] + base_pr[q][1], 8 * hit + s) == base_root:
                p = 8 * hit + s
        assert p is not None, "base trace path of query %d does not verify" % q
        positions.append(p)
    assert positions == sorted(positions)
    print("query positions:", positions)

    for q, p in enumerate(positions):
        cleaf = rowhash(comp_rows[2 * q:2 * q + 2])
        assert cleaf == comp_pr[q][3] and climb(cleaf, [comp_pr[q][2]] + comp_pr[q][1], p) == comp_root
        tag, nodes, sib, leaf = ext_pr[q]
        if nce == 1:                      # single column: raw-element leaves, first layer = hash_elements([l0, l1])
            assert tag == 1 and leaf == ext_rows[q]
            pair = oracle.to_mont([leaf, sib] if (p & 1) == 0 else [sib, leaf])
            first = bytes(oracle.hash_rows(KECCAK_M20, [pair[0:1], pair[1:2]])[0])
            assert climb(first, nodes, p >> 1) == ext_root
        else:
            eleaf = rowhash(ext_rows[nce * q:nce * q + nce])
            assert tag == 0 and eleaf == leaf and climb(eleaf, [sib] + nodes, p) == ext_root
    for li, (vals, pr, root) in enumerate(layers):
        ps = sorted(set(pp >> (3 * (li + 1)) for pp in positions))
        assert len(ps) == len(vals) // 8
        for r, pos in enumerate(ps):
            leaf = rowhash(vals[8 * r:8 * r + 8])
            assert leaf == pr[r][3] and climb(leaf, [pr[r][2]] + pr[r][1], pos) == root
    print("every opening verifies: base, extension, composition, %d FRI layers" % nl)

    # index <-> point map
    brev = lambda x, bits: int(format(x, "0%db" % bits)[::-1], 2) if bits else 0
    consts = []
    for li in range(nl - 1):
        rows_log = (trace_len * opts[1]).bit_length() - 1 - 3 * (li + 1)
        w = pow(3, (P - 1) >> (rows_log + 3), P)
        ps = sorted(set(pp >> (3 * (li + 1)) for pp in positions))
        br = {nxt[(li, r)][2] * pow(w, brev(ps[r], rows_log), P) % P for r in range(len(ps))}
        nat = {nxt[(li, r)][2] * pow(w, ps[r], P) % P for r in range(len(ps))}
        assert len(br) == 1 and len(nat) > 1
        consts.append(next(iter(br)))
    print("index i <-> offset * w^bitrev(i) holds on every layer (natural order does not)")

    # remainder: the last layer folds onto a polynomial in the UNSHIFTED variable t = y / offset:
    #   8 * P_r(alpha) = R(w'^bitrev(r))  has one alpha/offset common to all rows; with the coset offset 3^(8^k)
    #   inside R's argument there is none.
    last_vals, _pr, _root = layers[-1]
    rows_log = (trace_len * opts[1]).bit_length() - 1 - 3 * nl
    wl = pow(3, (P - 1) >> (rows_log + 3), P)
    ps = sorted(set(pp >> (3 * nl) for pp in positions))

    def evalp(c, x):
        a = 0
        for k in reversed(c):
            a = (a * x + k) % P
        return a

    def common_alpha(offset):
        sets = []
        for r, pos in enumerate(ps):
            x = offset * pow(wl, brev(pos, rows_log), P) % P
            f = [8 * c % P for c in G.fold_poly(last_vals[8 * r:8 * r + 8], True)]
            f[0] = (f[0] - evalp(remainder, pow(x, 8, P))) % P
            sets.append({b * x % P for b in G.roots(f)})
        return set.intersection(*sets)
    unshifted, shifted = common_alpha(1), common_alpha(pow(3, 8 ** (nl - 1), P))      # layer nl-1 sits on 3^(8^(nl-1)) * <w>
    assert len(unshifted) == 1 and len(shifted) == 0
    rem_const = next(iter(unshifted))
    print("remainder = interpolant of the folded last layer over the unshifted domain (%d coefficients)" % len(remainder))

    hx = lambda b: b.hex()
    hv = lambda v: "%x" % v
    K = 4
    out = {"file": rel, "options": opts, "trace_len": trace_len, "pow_nonce": nonce,
           "roots": {"base": hx(base_root), "extension": hx(ext_root), "composition": hx(comp_root),
                     "fri_layers": [hx(l[2]) for l in layers]},
           "remainder": [hv(v) for v in remainder], "positions": positions,
           "alpha_over_offset": [hv(c) for c in consts], "last_alpha_over_offset": hv(rem_const),
           "last_layer_rows": [[hv(v) for v in last_vals[8 * r:8 * r + 8]] for r in range(len(last_vals) // 8)],
           "queries": []}
    for q in range(K):
        p = positions[q]
        e = {"position": p,
             "base": {"row": [hv(v) for v in base_rows[ncb * q:ncb * q + ncb]],
                      "path": [hx(base_pr[q][2])] + [hx(d) for d in base_pr[q][1]]},
             "composition": {"row": [hv(v) for v in comp_rows[2 * q:2 * q + 2]],
                             "path": [hx(comp_pr[q][2])] + [hx(d) for d in comp_pr[q][1]]},
             "extension": ({"leaf": hv(ext_pr[q][3]), "sibling": hv(ext_pr[q][2]), "path": [hx(d) for d in ext_pr[q][1]]}
                           if nce == 1 else
                           {"row": [hv(v) for v in ext_rows[nce * q:nce * q + nce]],
                            "path": [hx(ext_pr[q][2])] + [hx(d) for d in ext_pr[q][1]]}),
             "fri": []}
        for li, (vals, pr, root) in enumerate(layers):
            ps = sorted(set(pp >> (3 * (li + 1)) for pp in positions))
            r = ps.index(p >> (3 * (li + 1)))
            e["fri"].append({"position": ps[r], "row": [hv(v) for v in vals[8 * r:8 * r + 8]],
                             "path": [hx(pr[r][2])] + [hx(d) for d in pr[r][1]]})
        out["queries"].append(e)
    with open(os.path.join(HERE, out_name), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", out_name, os.path.getsize(os.path.join(HERE, out_name)), "bytes")


if __name__ == "__main__":
    main()
