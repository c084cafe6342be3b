"""The 64-bit field variant (BASELINE.json configs[4]): p = 2^64 - 2^32 + 1 with Fq3 = Fp[X]/(X^3 - 2).

PARITY UNPINNED: the reference's field crate (ministark-gpu, p18446744069414584321) is un-vendored and the reference holds
no vector, proof or constant for this field.  CPU: the oracle (oracle/goldilocks.c) against the definitions in Python
integers, and the one constant that is public knowledge (the 2^32-th root of unity 7^((p-1)/2^32) = 1753635133440165772).
GPU: ss_ntt_gl64 / ss_lde_gl64 / ss_fri_fold_gl64x3 against the oracle, bit for bit, and size-independent properties at
2^24-2^25 points."""
import numpy as np
import pytest

GL_P = 2**64 - 2**32 + 1


def rand_fp(rng, n):
    return (rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)) % np.uint64(GL_P)


def test_oracle_is_the_definition(oracle):
    assert oracle.gl_root_of_unity(32) == 1753635133440165772 == pow(7, (GL_P - 1) >> 32, GL_P)
    assert pow(2, (GL_P - 1) // 3, GL_P) != 1                       # 2 is a cubic non-residue: X^3 - 2 is irreducible
    rng = np.random.default_rng(1)
    n, off = 16, 7
    a = rand_fp(rng, n)
    w = pow(7, (GL_P - 1) // n, GL_P)
    want = [sum(int(a[i]) * pow(off * pow(w, k, GL_P), i, GL_P) for i in range(n)) % GL_P for k in range(n)]
    got = oracle.gl_ntt(a, offset=off)
    assert [int(v) for v in got] == want
    assert np.array_equal(oracle.gl_ntt(got, inverse=True, offset=off), a)
    ev, co = oracle.gl_lde(a, 1, off)
    assert np.array_equal(co, oracle.gl_ntt(a, inverse=True))
    W = pow(7, (GL_P - 1) // (2 * n), GL_P)
    assert [int(v) for v in ev] == [sum(int(co[i]) * pow(off * pow(W, k, GL_P), i, GL_P) for i in range(n)) % GL_P for k in range(2 * n)]
    # FRI fold over Fq3: the folded layer of a degree < L/blowup polynomial is a polynomial of degree < that / fold
    L, fold = 64, 8
    coeffs = np.stack([np.concatenate([rand_fp(rng, L // 2), np.zeros(L // 2, dtype=np.uint64)]) for _ in range(3)], axis=1)
    evals = np.stack([oracle.gl_ntt(coeffs[:, c].copy(), offset=off) for c in range(3)], axis=1)
    alpha = rand_fp(rng, 3)
    folded = oracle.gl3_fri_fold(evals, fold, alpha, off)
    back = np.stack([oracle.gl_ntt(folded[:, c].copy(), inverse=True, offset=pow(off, fold, GL_P)) for c in range(3)], axis=1)
    assert not back[L // 2 // fold:].any() and back[:L // 2 // fold].any()
    # ... and it is the definition: f(x) = sum_k x^k f_k(x^fold), folded = sum_k alpha^k f_k
    def mul3(x, y):
        d = [0] * 5
        for i in range(3):
            for j in range(3):
                d[i + j] += int(x[i]) * int(y[j])
        return [(d[0] + 2 * d[3]) % GL_P, (d[1] + 2 * d[4]) % GL_P, d[2] % GL_P]
    apow = [[1, 0, 0]]
    for _ in range(fold - 1):
        apow.append(mul3(apow[-1], alpha))
    want = np.zeros((L // fold, 3), dtype=object)
    for t in range(L // fold):                                         # coefficient t of the folded polynomial
        acc = [0, 0, 0]
        for k in range(fold):
            term = mul3(apow[k], coeffs[fold * t + k])
            acc = [(acc[c] + term[c]) % GL_P for c in range(3)]
        want[t] = acc
    assert [[int(v) for v in row] for row in back[:L // fold]] == [[int(v) for v in row] for row in want]


@pytest.fixture(scope="module")
def ctx():
    from sandstorm_amd.backend import Context
    c = Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [1, 2, 5, 10, 13, 14, 17, 20])
def test_ntt_and_lde_vs_oracle(ctx, oracle, log_n):
    from sandstorm_amd import backend as be
    rng = np.random.default_rng(log_n)
    n = 1 << log_n
    cols = [rand_fp(rng, n) for _ in range(3)]
    for off in (1, 7):
        d = [ctx.column(c) for c in cols]
        ctx.ntt_gl64(d, log_n, be.FORWARD, off)
        for c, dc in zip(cols, d):
            assert np.array_equal(dc.download(np.uint64, (n,)), oracle.gl_ntt(c, offset=off)), (log_n, off)
        ctx.ntt_gl64(d, log_n, be.INVERSE, off)
        for c, dc in zip(cols, d):
            assert np.array_equal(dc.download(np.uint64, (n,)), c)
    # bit-reversed orders are permutations of the natural ones
    d = [ctx.column(cols[0])]
    ctx.ntt_gl64(d, log_n, be.INVERSE, 7, be.NATURAL, be.BITREV)
    rev = np.array([int(format(i, "0%db" % log_n)[::-1], 2) for i in range(n)])
    assert np.array_equal(d[0].download(np.uint64, (n,))[rev], oracle.gl_ntt(cols[0], inverse=True, offset=7))
    ctx.ntt_gl64(d, log_n, be.FORWARD, 7, be.BITREV, be.NATURAL)
    assert np.array_equal(d[0].download(np.uint64, (n,)), cols[0])
    if log_n <= 17:
        for lb in (1, 2):
            ev = [ctx.alloc(8 << (log_n + lb)) for _ in cols]
            co = [ctx.alloc(8 * n) for _ in cols]
            ctx.lde_gl64([ctx.column(c) for c in cols], log_n, lb, 7, ev, co)
            for c, e, k in zip(cols, ev, co):
                want_ev, want_co = oracle.gl_lde(c, lb, 7)
                assert np.array_equal(e.download(np.uint64, (n << lb,)), want_ev)
                assert np.array_equal(k.download(np.uint64, (n,))[rev], want_co)


@pytest.mark.gpu
def test_lde_at_the_benchmark_size_is_consistent(ctx, oracle):
    """2^24-row columns, blowup 2 (BASELINE configs[4]): the even rows of the extension over the coset 1 * <w_2n> are the
    input; over the coset 7 * <w_2n> interpolating back gives zero upper coefficients and the same lower ones"""
    from sandstorm_amd import backend as be
    log_n = 24
    n = 1 << log_n
    rng = np.random.default_rng(9)
    col = rand_fp(rng, n)
    d_in, ev, co = ctx.column(col), ctx.alloc(16 * n), ctx.alloc(8 * n)
    ctx.lde_gl64([d_in], log_n, 1, 1, [ev], [co])
    assert np.array_equal(ev.download(np.uint64, (2 * n,))[0::2], col)
    ctx.lde_gl64([d_in], log_n, 1, 7, [ev], None)
    ctx.ntt_gl64([ev], log_n + 1, be.INVERSE, 7, be.NATURAL, be.BITREV)
    got = ev.download(np.uint64, (2 * n,))
    assert not got[1::2].any()                                           # bit-reversed: odd slots are the upper coefficients
    assert np.array_equal(got[0::2], co.download(np.uint64, (n,)))


@pytest.mark.gpu
@pytest.mark.parametrize("fold,log_len", [(2, 6), (4, 8), (8, 12), (16, 12), (8, 18)])
def test_fri_fold_vs_oracle(ctx, oracle, fold, log_len):
    from sandstorm_amd import backend as be
    rng = np.random.default_rng(fold + log_len)
    L = 1 << log_len
    evals = np.stack([rand_fp(rng, L) for _ in range(3)], axis=1)
    alpha = rand_fp(rng, 3)
    for flags, un in ((0, False), (be.FRI_UNNORMALISED, True)):
        out = ctx.alloc(24 * (L // fold))
        ctx.fri_fold_gl64x3(ctx.column(evals), log_len, fold, alpha, 7, out, flags)
        assert np.array_equal(out.download(np.uint64, (L // fold, 3)), oracle.gl3_fri_fold(evals, fold, alpha, 7, un))
