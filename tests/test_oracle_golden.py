"""Pins the CPU oracle against the reference's own known-answer tests
(SURVEY.md §4 / §8c).  Runs without a GPU."""
import hashlib

import numpy as np
import pytest

from tests.util import P, felt_int, int_limbs, random_column


def test_montgomery_constants(oracle, golden):
    g = golden("montgomery.json")
    assert int(g["modulus"]) == P
    one = oracle.to_mont([1])[0]
    assert felt_int(one) == int(g["R_mod_p"])
    assert int(g["inv64"]) == 2**64 - 1
    assert oracle.from_mont(one) == 1


def test_field_ops_vs_python(oracle):
    rng = np.random.default_rng(1)
    vals = [int.from_bytes(rng.bytes(32), "little") % P for _ in range(64)] + [0, 1, P - 1, P - 2, 2**251]
    a = oracle.to_mont(vals)
    # a*b via the Pedersen-free path: poly_eval of degree-1 polynomial c0 + c1*x
    for i in range(0, len(vals) - 2, 3):
        c = np.stack([a[i], a[i + 1]])
        got = oracle.from_mont(oracle.poly_eval(c, a[i + 2]))
        assert got == (vals[i] + vals[i + 1] * vals[i + 2]) % P


@pytest.mark.parametrize("name,n", [("ntt_pedersen512.json", 512), ("ntt_ecdsa256.json", 256)])
def test_ntt_periodic_column_kat(oracle, golden, name, n):
    g = golden(name)
    for axis in ("x", "y"):
        coeffs = oracle.to_mont([int(v) for v in g["coeffs_" + axis]])
        want = [int(v) for v in g["evals_" + axis]]
        assert len(want) == n
        got = oracle.from_mont(oracle.ntt(coeffs))
        assert list(got) == want
        # ... and back: interpolation is the inverse
        back = oracle.ntt(oracle.to_mont(want), inverse=True)
        assert np.array_equal(back, coeffs)


def test_ntt_poseidon_round_key_kat(oracle, golden):
    """full_round_keys{0,1,2}_match (builtins/src/poseidon/periodic.rs:241-290): 8-point NTTs."""
    g = golden("ntt_poseidon8.json")
    for k in range(3):
        coeffs = oracle.to_mont([int(v) for v in g["key%d" % k]["coeffs"]])
        want = [int(v) for v in g["key%d" % k]["evals"]]
        assert list(oracle.from_mont(oracle.ntt(coeffs))) == want
        assert np.array_equal(oracle.ntt(oracle.to_mont(want), inverse=True), coeffs)


def test_fri_fold_matches_reference_proofs(oracle, golden):
    """The reference's OWN proofs pin the fold: every queried row of every FRI layer of the three shipped proof
    files, folded at beta = alpha / x (recovered from the data alone, tests/golden/make_fri_golden.py), gives
    the matching entry of the next layer - under bit-reversed rows + the unnormalised fold for the two files of
    the current code path, natural rows + the normalised fold for the older one."""
    g = golden("fri_saved_proofs.json")
    one = oracle.to_mont([1])[0]
    seen = set()
    for v in g["vectors"]:
        conv = g["conventions"][v["convention"]]
        flags = (oracle.FRI_BITREV_ROWS if conv["bitrev_rows"] else 0) | (oracle.FRI_UNNORMALISED if conv["scale"] == 8 else 0)
        row = oracle.to_mont([int(x, 16) for x in v["values"]])
        beta = oracle.to_mont([int(v["beta"], 16)])[0]
        got = oracle.from_mont(oracle.fri_fold(row, 8, beta, one, flags))
        assert got[0] == int(v["next"], 16), (v["file"], v["layer"], v["row"])
        # ... and NOT under the other convention (the pin discriminates)
        assert oracle.from_mont(oracle.fri_fold(row, 8, beta, one, flags ^ 3))[0] != int(v["next"], 16)
        seen.add((v["file"], v["convention"]))
    assert len(seen) == 3 and len(g["vectors"]) >= 100


def _climb(oracle, cur, path, pos, kind=1):
    import ctypes as C
    lib = oracle.lib()
    for lvl, sib in enumerate(path):
        a, b = (cur, sib) if ((pos >> lvl) & 1) == 0 else (sib, cur)
        out = (C.c_uint8 * 32)()
        lib.or_hash_merge(C.c_int(kind), (C.c_uint8 * 32).from_buffer_copy(a), (C.c_uint8 * 32).from_buffer_copy(b), out)
        cur = bytes(out)
    return cur


def _rowhash(oracle, vals_hex, kind=1):
    m = oracle.to_mont([int(v, 16) for v in vals_hex])
    return bytes(oracle.hash_rows(kind, [m[k:k + 1] for k in range(len(vals_hex))])[0])


PROOF_FIXTURES = ["saved_proof_openings.json", "saved_proof_openings_recursive.json"]


@pytest.mark.parametrize("fixture", PROOF_FIXTURES)
def test_merkle_openings_of_reference_proof(oracle, golden, fixture):
    """Every Merkle opening of the reference's shipped proofs - example/array-sum.proof.saved (starknet shape, 9+1
    columns, 16 queries, 6 FRI layers) and bootloader-proof.bin (recursive shape, 7+3 columns, 40 queries, 4 layers);
    tests/golden/make_proof_golden.py checked every query, 4 are committed - verifies against the proof's roots with
    the oracle's row hash (H1), unhashed-leaf first layer (H2), node hash and pairing (H3/H4): the reference's own
    data pins them.  Position p opens row p of the trace trees and row p >> 3(i+1) of FRI layer i."""
    g = golden(fixture)
    roots = g["roots"]
    for q in g["queries"]:
        p = q["position"]
        for name in ("base", "composition"):
            leaf = _rowhash(oracle, q[name]["row"])
            assert _climb(oracle, leaf, [bytes.fromhex(d) for d in q[name]["path"]], p).hex() == roots[name]
        if "leaf" in q["extension"]:                                                   # single-column extension trace
            pair = [q["extension"]["leaf"], q["extension"]["sibling"]]
            first = _rowhash(oracle, pair if (p & 1) == 0 else pair[::-1])            # hash_elements([l0, l1])
            assert _climb(oracle, first, [bytes.fromhex(d) for d in q["extension"]["path"]], p >> 1).hex() == roots["extension"]
        else:
            leaf = _rowhash(oracle, q["extension"]["row"])
            assert _climb(oracle, leaf, [bytes.fromhex(d) for d in q["extension"]["path"]], p).hex() == roots["extension"]
        for li, f in enumerate(q["fri"]):
            assert f["position"] == p >> (3 * (li + 1))
            leaf = _rowhash(oracle, f["row"])
            assert _climb(oracle, leaf, [bytes.fromhex(d) for d in f["path"]], f["position"]).hex() == roots["fri_layers"][li]
        # a wrong sibling order must not verify (the pin discriminates)
        leaf = _rowhash(oracle, q["base"]["row"])
        assert _climb(oracle, leaf, [bytes.fromhex(d) for d in q["base"]["path"]], p ^ 1).hex() != roots["base"]


@pytest.mark.parametrize("fixture", PROOF_FIXTURES)
def test_committed_order_is_bit_reversed(golden, fixture):
    """beta = alpha / x per queried FRI row (fri_saved_proofs.json) times w_L^bitrev(row position) is ONE constant
    per layer - and 16 different values under the natural map: index i of a committed vector is the point
    offset * w_L^bitrev(i), w_L = 3^((p-1)/L)."""
    g, fri = golden(fixture), golden("fri_saved_proofs.json")
    positions = g["positions"]
    log_N = (g["trace_len"] * g["options"][1]).bit_length() - 1
    brev = lambda x, bits: int(format(x, "0%db" % bits)[::-1], 2) if bits else 0
    for li, want in enumerate(g["alpha_over_offset"]):
        rows_log = log_N - 3 * (li + 1)
        w = pow(3, (P - 1) >> (rows_log + 3), P)
        ps = sorted(set(pp >> (3 * (li + 1)) for pp in positions))
        betas = {v["row"]: int(v["beta"], 16) for v in fri["vectors"]
                 if v["file"] == g["file"] and v["layer"] == li}
        assert {betas[r] * pow(w, brev(ps[r], rows_log), P) % P for r in betas} == {int(want, 16)}
        assert len({betas[r] * pow(w, ps[r], P) % P for r in betas}) > 1


@pytest.mark.parametrize("fixture", PROOF_FIXTURES)
def test_remainder_of_reference_proof(oracle, golden, fixture):
    """The last FRI layer of the saved proof folds (bit-reversed rows, unnormalised) onto its remainder polynomial
    taken in the UNSHIFTED variable: fold(row; alpha/offset as recovered, x = w'^bitrev(pos)) == R(x^8)."""
    g = golden(fixture)
    rem = [int(v, 16) for v in g["remainder"]]
    c = int(g["last_alpha_over_offset"], 16)
    nl = len(g["roots"]["fri_layers"])
    rows_log = (g["trace_len"] * g["options"][1]).bit_length() - 1 - 3 * nl
    w = pow(3, (P - 1) >> (rows_log + 3), P)
    ps = sorted(set(pp >> (3 * nl) for pp in g["positions"]))
    brev = lambda x, bits: int(format(x, "0%db" % bits)[::-1], 2) if bits else 0
    one = oracle.to_mont([1])[0]
    for r, pos in enumerate(ps):
        x = pow(w, brev(pos, rows_log), P)
        row = oracle.to_mont([int(v, 16) for v in g["last_layer_rows"][r]])
        beta = oracle.to_mont([c * pow(x, -1, P) % P])[0]
        got = oracle.from_mont(oracle.fri_fold(row, 8, beta, one, oracle.FRI_BITREV_ROWS | oracle.FRI_UNNORMALISED))[0]
        y, want = pow(x, 8, P), 0
        for k in reversed(rem):
            want = (want * y + k) % P
        assert got == want, r


def test_saved_proof_fixture_shape(golden):
    """Header and out-of-domain tail of the reference's three saved proofs (data only):
    mask sizes 269 (starknet) / 133 (recursive) are the masks of the restated AIRs (tests/survey_masks.py), the
    values are canonical field elements, and the option bytes are the CLI's.  The OOD identity itself needs the
    restated constraint sets."""
    proofs = golden("saved_proofs.json")
    assert [len(p["ood_trace"]) for p in proofs] == [269, 269, 133]
    from tests import survey_masks as sm
    assert sum(sm.STARKNET_CELLS_PER_COLUMN) == 269 and sum(len(v) for v in sm.RECURSIVE_MASK.values()) == 133
    for p in proofs:
        assert len(p["ood_composition"]) == 2
        assert all(0 <= int(v) < P for v in p["ood_trace"] + p["ood_composition"])
        assert p["options"]["lde_blowup_factor"] == 2 and p["options"]["fri_folding_factor"] == 8
        assert p["trace_len"] in (1 << 21, 1 << 18)


def test_ntt_coset_matches_definition(oracle):
    n = 64
    col = random_column(n, 3)
    g = oracle.to_mont([3])[0]
    ev = oracle.ntt(col, offset=g)
    w = pow(3, (P - 1) // n, P)
    coeffs = list(oracle.from_mont(col))
    for k in (0, 1, 7, 63):
        x = 3 * pow(w, k, P) % P
        want = sum(c * pow(x, i, P) for i, c in enumerate(coeffs)) % P
        assert oracle.from_mont(ev[k]) == want
    assert np.array_equal(oracle.ntt(ev, inverse=True, offset=g), col)


def test_lde_matches_definition(oracle):
    n, lb = 32, 1
    col = random_column(n, 5)
    g = oracle.to_mont([3])[0]
    ev, co = oracle.lde(col, lb, g)
    # interpolant reproduces the trace on <w_n> ...
    assert np.array_equal(oracle.ntt(co), col)
    # ... and the LDE is its evaluation on 3<w_2n>
    w = pow(3, (P - 1) // (n << lb), P)
    coeffs = list(oracle.from_mont(co))
    for k in (0, 1, 2, 33, 63):
        x = 3 * pow(w, k, P) % P
        assert oracle.from_mont(ev[k]) == sum(c * pow(x, i, P) for i, c in enumerate(coeffs)) % P


def test_keccak_and_blake2s_vectors(oracle):
    assert oracle.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert oracle.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    for ln in (0, 1, 55, 63, 64, 65, 127, 128, 129, 135, 136, 137, 271, 272, 320, 1000):
        msg = bytes((7 * i + ln) & 0xff for i in range(ln))
        assert oracle.blake2s256(msg) == hashlib.blake2s(msg).digest(), ln
    # multi-block Keccak self-consistency against the sponge definition is covered by the coin KAT


def test_solidity_coin_kat(oracle, golden):
    g = golden("coins.json")
    coin = oracle.Coin(0, bytes(32))
    for want in g["solidity_zero_seed_draws"]:
        assert oracle.from_mont(coin.draw()) == int(want)
    assert coin.counter == 4


def test_cairo_coin_kat(oracle, golden):
    g = golden("coins.json")["cairo_reseed"]
    coin = oracle.Coin(1, bytes.fromhex(g["seed"]))
    coin.reseed_bytes(int(g["element"]).to_bytes(32, "big"))
    assert coin.digest.hex() == g["digest"]
    assert coin.counter == 0


def test_pedersen_kat(oracle, golden):
    g = golden("pedersen.json")
    for case in g["hash_examples"] + g["extra"]:
        a, b = oracle.to_mont([int(case["a"]), int(case["b"])])
        assert oracle.from_mont(oracle.pedersen_hash(a, b)) == int(case["hash"]), case
    # doubling chains 2^i P_k against the periodic-column evaluations
    k = golden("ntt_pedersen512.json")
    xs, ys = oracle.pedersen_doublings(1, 248)
    assert list(oracle.from_mont(xs)) == [int(v) for v in k["evals_x"][:248]]
    assert list(oracle.from_mont(ys)) == [int(v) for v in k["evals_y"][:248]]
    xs, _ = oracle.pedersen_doublings(4, 4)
    assert list(oracle.from_mont(xs)) == [int(v) for v in k["evals_x"][504:508]]


def test_hash_rows_layout(oracle):
    """hash_elements absorbs to_montgomery(e).to_be_bytes::<32>() per element
    (crypto/src/hash/keccak.rs:50-58); masks per hash/mod.rs:5-23."""
    cols = [random_column(8, c) for c in range(3)]
    for kind, fn, mask in ((0, oracle.keccak256, None), (1, oracle.keccak256, "k"),
                           (2, oracle.blake2s256, None), (3, oracle.blake2s256, "b")):
        got = oracle.hash_rows(kind, cols)
        for r in range(8):
            msg = b"".join(felt_int(c[r]).to_bytes(32, "big") for c in cols)
            d = bytearray(fn(msg))
            if mask == "k":
                d[20:] = bytes(12)
            if mask == "b":
                d[:12] = bytes(12)
            assert bytes(got[r]) == bytes(d)


def test_pow_grind_and_verify(oracle):
    for kind in (0, 1):
        coin = oracle.Coin(kind, bytes(range(32)))
        nonce = coin.grind(8)
        assert nonce >= 1 and coin.verify_pow(8, nonce)
        assert all(not coin.verify_pow(8, k) for k in range(1, nonce))


def test_draw_queries(oracle):
    sol = oracle.Coin(0, bytes(32)).draw_queries(16, 1 << 10)
    assert len(sol) <= 16 and all(q < 1024 for q in sol)
    c0, c1 = oracle.Coin(1, bytes(32)), oracle.Coin(1, bytes(32))
    a, b = c0.draw_queries(5, 1 << 10), c1.draw_queries(8, 1 << 10)
    assert c0.counter == c1.counter == 2          # batches of 4 (cairo.rs:124-130)
    assert set(a) <= set(b)


def test_lde_offset_pinned_by_the_reference_proof(oracle, golden):
    """SURVEY Appendix A, M2 (LDE coset offset), from data alone: the diluted-check columns of a run without bitwise
    instances do not depend on the program; regenerated (layouts/recursive.py) and extended over 3 * <w_N>, they equal
    what the reference's shipped recursive proof opens at all 40 of its query positions, position q holding the point
    3 * w_N^bitrev(q) — and they do not for another offset or the natural order."""
    from sandstorm_amd.layouts import recursive as rec
    from sandstorm_amd.prover import bitrev
    from tests.test_layout_recursive import load_run
    fx = golden("lde_offset_pin.json")
    states, memory, pi = load_run()
    cols = rec.base_trace(states, memory, pi)
    n = len(cols[0])
    assert n == fx["trace_len"]
    log_N = (n * fx["lde_blowup"]).bit_length() - 1
    want = {1: [int(v, 16) for v in fx["column1"]], 2: [int(v, 16) for v in fx["column2"]]}

    def hits(offset, order):
        g = oracle.to_mont([offset])[0]
        total = 0
        for c in (1, 2):
            lde = oracle.lde(oracle.to_mont(cols[c]), 1, g)[0]
            for q, v in zip(fx["positions"], want[c]):
                idx = bitrev(q, log_N) if order == "bitrev" else q
                total += int(oracle.from_mont(lde[idx][None])[0]) == v
        return total
    assert hits(3, "bitrev") == 2 * len(fx["positions"]) == 80
    assert hits(3, "natural") == 0 and hits(1, "bitrev") == 0 and hits(9, "bitrev") == 0


def test_deep_composition_pinned_by_the_reference_proof(oracle, golden):
    """SURVEY Appendix A M5 / M6 and the order of the out-of-domain vector, from the reference's shipped recursive proof
    alone (tests/golden/make_deep_pin_golden.py recovers z and alpha from its data):
      * the 33 out-of-domain values of the two program-independent columns equal T_c(z w_n^offset), in sorted
        (column, offset) order;
      * at every one of its 40 query points the first FRI layer holds
          sum_j alpha^j (T_j(x) - ood_j) / (x - z w^o_j) + sum_k alpha^(133+k) (H_k(x) - oodc_k) / (x - z^2)
        with the mask of sandstorm_amd/layouts/recursive.py in sorted order, then the composition columns;
    (the same expression is the big-integer definition the oracle's or_deep_compose is held to in tests/test_oracle_defs.py,
    and the GPU kernel is held to the oracle in tests/test_gpu_parity.py)."""
    from sandstorm_amd.layouts import recursive as rec
    from sandstorm_amd.prover import bitrev
    from tests.test_layout_recursive import load_run
    fx = golden("deep_pin_recursive.json")
    h = lambda key: [int(v, 16) for v in fx[key]]
    z, alpha, n = int(fx["z"], 16), int(fx["deep_alpha"], 16), fx["trace_len"]
    ood_t, ood_c, base, ext, comp, deep = h("ood_trace"), h("ood_composition"), h("base_rows"), h("extension_rows"), h("composition_rows"), h("deep_values")
    mask = rec.mask()
    assert len(mask) == len(ood_t) == 133
    wn = pow(3, (P - 1) // n, P)
    states, memory, pi = load_run()
    cols = rec.base_trace(states, memory, pi)
    for c in (1, 2):
        coeffs = oracle.ntt(oracle.to_mont(cols[c]), inverse=True)
        for j, (cc, o) in enumerate(mask):
            if cc == c:
                assert int(oracle.from_mont(oracle.poly_eval(coeffs, oracle.to_mont([z * pow(wn, o, P) % P])[0])[None])[0]) == ood_t[j]
    N = 2 * n
    w_N = pow(3, (P - 1) // N, P)
    for qi, q in enumerate(fx["positions"]):
        x = 3 * pow(w_N, bitrev(q, N.bit_length() - 1), P) % P
        t = base[7 * qi: 7 * qi + 7] + ext[3 * qi: 3 * qi + 3]
        acc = 0
        for j, (col, o) in enumerate(mask):
            acc += pow(alpha, j, P) * (t[col] - ood_t[j]) * pow((x - z * pow(wn, o, P)) % P, -1, P)
        for k in range(2):
            acc += pow(alpha, 133 + k, P) * (comp[2 * qi + k] - ood_c[k]) * pow((x - z * z) % P, -1, P)
        assert acc % P == deep[qi]
        # a perturbed convention fails: composition point z instead of z^2
        bad = acc - sum(pow(alpha, 133 + k, P) * (comp[2 * qi + k] - ood_c[k]) * (pow((x - z * z) % P, -1, P) - pow((x - z) % P, -1, P)) for k in range(2))
        assert bad % P != deep[qi]


def test_starknet_ood_vector_is_in_column_order(golden):
    """The starknet-layout proofs' 269 out-of-domain values are grouped by column with the per-column cell counts of
    SURVEY.md 8a (16, 5, 4, 9, 2, ...): in `example/array-sum.proof.saved`, a run without Pedersen instances, the
    Pedersen partial-sum columns 1 and 2 are the constants P0.x and P0.y and columns 3 and 4 (suffix, slope) are zero —
    exactly the 5, 4, 9 and 2 entries that follow the 16 flag cells, and no entry next to them."""
    from sandstorm_amd.layouts.recursive import PEDERSEN_POINTS
    entry = next(e for e in golden("saved_proofs.json") if e["file"] == "example/array-sum.proof.saved")
    ood = [int(v) for v in entry["ood_trace"]]
    px, py = PEDERSEN_POINTS[0]
    counts = [16, 5, 4, 9, 2]
    start = [sum(counts[:c]) for c in range(len(counts) + 1)]
    assert ood[start[1]:start[2]] == [px] * 5 and ood[start[2]:start[3]] == [py] * 4
    assert ood[start[3]:start[4]] == [0] * 9 and ood[start[4]:start[5]] == [0] * 2
    assert ood[start[1] - 1] not in (px, py, 0) and ood[start[5]] != 0


def test_sha256_is_the_fips_180_4_function(oracle):
    """the oracle's SHA-256 (oracle/hash.c: the hash of the trees the reference names for its 64-bit-field claim, cli/src/main.rs:105,119)
    against the standard's own example vectors (FIPS 180-4 / NIST CSRC "SHA-256 examples": one block, two blocks, the empty message,
    one million 'a') and, at every length across the padding boundaries, against OpenSSL's (hashlib)"""
    import hashlib
    kat = {b"abc": "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad",
           b"": "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855",
           b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq": "248d6a61d20638b8e5c026930c3e6039a33ce45964ff2167f6ecedd419db06c1",
           b"a" * 1000000: "cdc76e5c9914fb9281a1c7e284d73e67f1809a48a497200e046d39ccc7112cd0"}
    for msg, want in kat.items():
        assert oracle.sha256(msg).hex() == want
    blob = bytes((i * 131 + 7) & 0xff for i in range(300))
    for n in range(0, 200):
        assert oracle.sha256(blob[:n]) == hashlib.sha256(blob[:n]).digest(), n
