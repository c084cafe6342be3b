"""Cross-checks the C oracle's unpinned-by-reference parts (Merkle layout,
FRI fold, DEEP, program interpreter) against pure-Python big-integer
restatements of their mathematical definitions.  No GPU."""
import numpy as np
import pytest

from tests import pyref
from tests.pyref import P
from tests.util import felt_int, random_column


@pytest.fixture(scope="module")
def pedersen(golden):
    pts = golden("pedersen.json")["points"]
    return pyref.make_pedersen([(int(pts["P%d" % i][0]), int(pts["P%d" % i][1])) for i in range(5)])


def _mont_be(limbs):
    return felt_int(limbs).to_bytes(32, "big")


@pytest.mark.parametrize("tree", [0, 1])
@pytest.mark.parametrize("leaf_kind", [0, 1])
def test_keccak_trees(oracle, tree, leaf_kind):
    n = 8
    mask = pyref.mask_keccak if tree == 1 else (lambda d: d)
    H = lambda b: mask(oracle.keccak256(b))
    if leaf_kind == 0:
        leaves = oracle.hash_rows(tree, [random_column(n, 0), random_column(n, 1)])
        level = [bytes(l) for l in leaves]
    else:
        leaves = random_column(n, 2)
        level = [_mont_be(l) for l in leaves]
    nodes, _ = oracle.merkle_build(tree, 0, leaf_kind, leaves)
    want = {n + i: v for i, v in enumerate(level)}
    for k in range(n - 1, 0, -1):
        want[k] = H(want[2 * k] + want[2 * k + 1])
    for k in range(1, 2 * n):
        assert bytes(nodes[k]) == want[k], k


@pytest.mark.parametrize("n_friendly", [0, 1, 2, 3, 22])
def test_friendly_tree_multicol(oracle, pedersen, n_friendly):
    """crypto/src/merkle/mod.rs:529-634 build trees with 0..3 Pedersen layers over 8 rows."""
    n = 8
    leaves = oracle.hash_rows(3, [random_column(n, 0), random_column(n, 1)])
    nodes, tags = oracle.merkle_build(2, n_friendly, 0, leaves)
    val = {n + i: bytes(l) for i, l in enumerate(leaves)}
    for k in range(n - 1, 0, -1):
        depth = k.bit_length() - 1
        a, b = val[2 * k], val[2 * k + 1]
        if depth < n_friendly:
            h = pedersen(int.from_bytes(a, "big") % P, int.from_bytes(b, "big") % P)
            val[k] = h.to_bytes(32, "big")
            assert tags[k] == 0
        else:
            val[k] = pyref.mask_blake(pyref.blake2s(a + b))
            assert tags[k] == 1
        assert bytes(nodes[k]) == val[k], (k, depth)


def test_friendly_tree_singlecol(oracle, pedersen):
    n = 8
    leaves = oracle.to_mont(list(range(n)))      # the reference's test column 0..7 (mod.rs:500-520)
    nodes, tags = oracle.merkle_build(2, 1, 1, leaves)
    val = {}
    for k in range(n // 2, n):
        l0, l1 = 2 * k - n, 2 * k - n + 1
        h = pedersen(pedersen(pedersen(0, l0), l1), 2)        # PedersenHashFn::hash_elements
        val[k] = h
    for k in range(n // 2 - 1, 0, -1):
        val[k] = pedersen(val[2 * k], val[2 * k + 1])
    for k in range(1, n):
        assert int.from_bytes(bytes(nodes[k]), "big") == val[k], k
        assert tags[k] == 0


@pytest.mark.parametrize("fold", [2, 4, 8, 16])
def test_fri_fold_definition(oracle, fold):
    L = 64
    ev = random_column(L, 9)
    vals = list(oracle.from_mont(ev))
    alpha, off = 0x1234567 ** 5 % P, 3
    out = oracle.fri_fold(ev, fold, oracle.to_mont([alpha])[0], oracle.to_mont([off])[0])
    w, wf = pyref.root_of_unity(L), pyref.root_of_unity(fold)
    rows = L // fold
    for j in range(rows):
        xj = off * pow(w, j, P) % P
        xs = [xj * pow(wf, k, P) % P for k in range(fold)]
        # x_j * wf^k = offset * w^(j + k*rows): the row really is evals[j + k*rows]
        assert xs[1] == off * pow(w, j + rows, P) % P
        ys = [vals[j + k * rows] for k in range(fold)]
        assert oracle.from_mont(out[j]) == pyref.interpolate_eval(xs, ys, alpha)


def test_fri_fold_lowers_degree(oracle):
    """folding evaluations of a degree<d polynomial gives evaluations of a degree<d/8 one
    on the domain offset^8 * <w^8>."""
    n, fold = 256, 8
    coeffs = np.zeros((n, 4), dtype=np.uint64)
    coeffs[:64] = random_column(64, 4)
    g = oracle.to_mont([3])[0]
    ev = oracle.ntt(coeffs, offset=g)
    folded = oracle.fri_fold(ev, fold, oracle.to_mont([77])[0], g)
    c2 = oracle.ntt(folded, inverse=True, offset=oracle.to_mont([pow(3, 8, P)])[0])
    assert not np.any(c2[8:]) and np.any(c2[:8])
    # coefficient m of the folded poly = sum_k alpha^k c[8m + k]
    cs = list(oracle.from_mont(coeffs))
    for m in range(8):
        assert oracle.from_mont(c2[m]) == sum(pow(77, k, P) * cs[8 * m + k] for k in range(8)) % P


def test_deep_definition(oracle):
    log_n, lb = 4, 1
    n, N = 1 << log_n, 1 << (log_n + lb)
    g = oracle.to_mont([3])[0]
    cols, coeffs = [], []
    for c in range(3):
        ev, co = oracle.lde(random_column(n, c), lb, g)
        cols.append(ev); coeffs.append(list(oracle.from_mont(co)))
    comp_coeffs = [random_column(n, 10 + k) for k in range(2)]
    comp = [oracle.ntt(np.concatenate([c, np.zeros((N - n, 4), dtype=np.uint64)]), offset=g) for c in comp_coeffs]
    z = 0xABCDEF0123456789 ** 3 % P
    wn = pyref.root_of_unity(n)
    mask = [(0, 0), (0, 1), (1, 0), (2, 3), (2, 5)]
    poly = lambda cs, x: sum(c * pow(x, i, P) for i, c in enumerate(cs)) % P
    ood_t = [poly(coeffs[c], z * pow(wn, o, P) % P) for c, o in mask]
    cc = [list(oracle.from_mont(c)) for c in comp_coeffs]
    ood_c = [poly(c, pow(z, 2, P)) for c in cc]
    alpha = 987654321987654321
    ct = [pow(alpha, j, P) for j in range(len(mask))]
    ccf = [pow(alpha, len(mask) + k, P) for k in range(2)]
    out = oracle.deep_compose(cols, comp, log_n, lb, g, [m[0] for m in mask], [m[1] for m in mask],
                              oracle.to_mont(ood_t), oracle.to_mont(ct), oracle.to_mont(ood_c),
                              oracle.to_mont(ccf), oracle.to_mont([z])[0])
    wN = pyref.root_of_unity(N)
    colv = [list(oracle.from_mont(c)) for c in cols]
    compv = [list(oracle.from_mont(c)) for c in comp]
    for i in range(N):
        x = 3 * pow(wN, i, P) % P
        acc = 0
        for j, (c, o) in enumerate(mask):
            acc += ct[j] * (colv[c][i] - ood_t[j]) * pow(x - z * pow(wn, o, P), -1, P)
        for k in range(2):
            acc += ccf[k] * (compv[k][i] - ood_c[k]) * pow(x - z * z, -1, P)
        assert oracle.from_mont(out[i]) == acc % P
    # the DEEP quotient of consistent OOD values is a polynomial of degree < n
    co = oracle.ntt(out, inverse=True, offset=g)
    assert not np.any(co[n:])


def test_program_interpreter(oracle):
    """(T0(x)*T1(next) - c0) * table0 + x, then 1/that via INV, against Python."""
    from sandstorm_amd.air_program import OP, SRC, instr
    log_n, lb = 3, 1
    n, N = 8, 16
    g = oracle.to_mont([3])[0]
    cols = [oracle.lde(random_column(n, c), lb, g)[0] for c in range(2)]
    table = random_column(4, 7)
    consts = random_column(2, 8)
    code = []
    code += instr(OP.MOV, 0, SRC.TRACE, (0 << 24) | 0)
    code += instr(OP.MUL, 0, SRC.TRACE, (1 << 24) | 1)
    code += instr(OP.SUB, 0, SRC.CONST, 0)
    code += instr(OP.MUL, 0, SRC.TABLE, 0)
    code += instr(OP.ADD, 0, SRC.X, 0)
    code += instr(OP.ST, 0, 0, 1)
    code += instr(OP.MOV, 1, SRC.SLOT, 1)
    code += instr(OP.INV, 1, 0, 0)
    code += instr(OP.RSUB, 1, SRC.ACC, 0)
    code += instr(OP.OUT, 1, 0, 0)
    out = oracle.eval_program(code, consts, table, [0, 2], 2, cols, log_n, lb, g)
    v = [list(oracle.from_mont(c)) for c in cols]
    t, c0 = list(oracle.from_mont(table)), oracle.from_mont(consts[0])
    wN = pyref.root_of_unity(N)
    for i in range(N):
        x = 3 * pow(wN, i, P) % P
        a = ((v[0][i] * v[1][(i + 2) % N] - c0) * t[i % 4] + x) % P
        assert oracle.from_mont(out[i]) == (a - pow(a, -1, P)) % P


def test_friendly_tree_large_threaded(oracle, pedersen):
    """2048 leaves: the oracle's OpenMP paths (regression: Pedersen tables must be ready before them)"""
    n = 2048
    leaves = oracle.hash_rows(3, [random_column(n, 0), random_column(n, 1)])
    nodes, tags = oracle.merkle_build(2, 22, 0, leaves)
    for k in (n // 2, n // 2 + 1, n - 1, 1500):
        a, b = bytes(nodes[2 * k]), bytes(nodes[2 * k + 1])
        assert int.from_bytes(bytes(nodes[k]), "big") == pedersen(int.from_bytes(a, "big") % P, int.from_bytes(b, "big") % P)


def test_fri_fold_conventions_agree(oracle):
    """BITREV_ROWS is the natural-order fold seen through the bit-reversal permutation of both layers;
    UNNORMALISED is `fold` times the normalised value."""
    log_len, fold = 9, 8
    n = 1 << log_len
    ev = random_column(n, 31)
    alpha, off = oracle.to_mont([0xABCDEF12345])[0], oracle.to_mont([3])[0]
    nat = oracle.fri_fold(ev, fold, alpha, off)
    br = lambda i, bits: int(format(i, "0%db" % bits)[::-1], 2)
    ev_br = ev[[br(i, log_len) for i in range(n)]]
    got = oracle.fri_fold(ev_br, fold, alpha, off, oracle.FRI_BITREV_ROWS)
    assert np.array_equal(got, nat[[br(i, log_len - 3) for i in range(n // fold)]])
    un = oracle.from_mont(oracle.fri_fold(ev, fold, alpha, off, oracle.FRI_UNNORMALISED))
    assert [int(x) for x in un] == [int(x) * fold % P for x in oracle.from_mont(nat)]


def test_pedersen_window_tables_equal_the_bitwise_sum(oracle):
    """oracle/pedersen.c computes the hash from 4-bit window tables (the shape of starknet-crypto's pedersen_hash, which the CPU leg of
    bench.py stands in for); the bit-by-bit point sum of the published definition is kept beside it: equal on random inputs, on inputs
    with empty and full digits, with only high bits, and on the reference's own examples (tests/test_oracle_golden.py pins those)"""
    import random
    rng = random.Random(66)
    p = 2**251 + 17 * 2**192 + 1
    cases = [(0, 0), (1, 0), (0, 1), (p - 1, p - 1), (15 << 248, 15 << 248), ((1 << 248) - 1, 1 << 251), (0xf0f0f0f0 << 100, 0x0f0f0f0f << 17)]
    cases += [(rng.randrange(p), rng.randrange(p)) for _ in range(40)]
    for a, b in cases:
        am, bm = oracle.to_mont([a % p])[0], oracle.to_mont([b % p])[0]
        assert np.array_equal(oracle.pedersen_hash(am, bm), oracle.pedersen_hash_bitwise(am, bm)), (hex(a), hex(b))
    assert 5 < oracle.mulmod_ns(200_000) < 2000
