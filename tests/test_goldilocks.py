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


# ---- DEEP over the cubic extension ---------------------------------------------------------------------------------------
def _mul3(x, y):
    d = [0] * 5
    for i in range(3):
        for j in range(3):
            d[i + j] += int(x[i]) * int(y[j])
    return [(d[0] + 2 * d[3]) % GL_P, (d[1] + 2 * d[4]) % GL_P, d[2] % GL_P]


def _deep_case(rng, oracle, log_n, lb, ncols, ncomp, offsets):
    """random trace / composition polynomials, a mask over `offsets`, honest out-of-domain values"""
    n = 1 << log_n
    trace = [rand_fp(rng, n) for _ in range(ncols)]
    lde, coeffs = zip(*[oracle.gl_lde(t, lb, 7) for t in trace])
    comp_coeffs = [rand_fp(rng, n) for _ in range(ncomp)]
    comp_lde = [oracle.gl_ntt(np.concatenate([c, np.zeros((n << lb) - n, dtype=np.uint64)]), offset=7) for c in comp_coeffs]
    mask = [(c, o) for c in range(ncols) for o in offsets if (c + o) % 3 != 1] or [(0, 0)]
    mc, mo = [c for c, _ in mask], [o for _, o in mask]
    z, zc = rand_fp(rng, 3), rand_fp(rng, 3)
    ood_t = oracle.gl3_ood_eval(coeffs, mc, mo, z)
    ood_c = oracle.gl3_ood_eval(comp_coeffs, list(range(ncomp)), [0] * ncomp, zc) if ncomp else np.zeros((0, 3), dtype=np.uint64)
    ct = np.stack([rand_fp(rng, 3) for _ in mask])
    cc = np.stack([rand_fp(rng, 3) for _ in range(ncomp)]) if ncomp else np.zeros((0, 3), dtype=np.uint64)
    return trace, lde, coeffs, comp_coeffs, comp_lde, mc, mo, z, zc, ood_t, ood_c, ct, cc


def test_deep_oracle_is_the_definition(oracle):
    rng = np.random.default_rng(5)
    a = rand_fp(rng, 3)
    assert _mul3(a, oracle.gl3_inv(a)) == [1, 0, 0]
    log_n, lb = 4, 1
    n, N = 1 << log_n, 2 << log_n
    trace, lde, coeffs, comp_coeffs, comp_lde, mc, mo, z, zc, ood_t, ood_c, ct, cc = _deep_case(rng, oracle, log_n, lb, 2, 1, [0, 1, 5])
    w = pow(7, (GL_P - 1) // n, GL_P)
    for j, (c, o) in enumerate(zip(mc, mo)):                         # Horner in Fq3 == the power sum in Python integers
        pt = [int(v) * pow(w, o, GL_P) % GL_P for v in z]
        acc, pw = [0, 0, 0], [1, 0, 0]
        for k in range(n):
            acc = [(acc[t] + int(coeffs[c][k]) * pw[t]) % GL_P for t in range(3)]
            pw = _mul3(pw, pt)
        assert [int(v) for v in ood_t[j]] == acc
    # honest out-of-domain values make every DEEP term a polynomial: the composition has degree < n on the LDE domain
    deep = oracle.gl3_deep_compose(lde, comp_lde, log_n, lb, 7, mc, mo, ood_t, ct, ood_c, cc, z, zc)
    for t in range(3):
        back = oracle.gl_ntt(deep[:, t].copy(), inverse=True, offset=7)
        assert not back[n:].any() and back[:n].any()
    # ... and a wrong one does not
    bad = ood_t.copy()
    bad[0, 0] = (int(bad[0, 0]) + 1) % GL_P
    deep = oracle.gl3_deep_compose(lde, comp_lde, log_n, lb, 7, mc, mo, bad, ct, ood_c, cc, z, zc)
    assert oracle.gl_ntt(deep[:, 0].copy(), inverse=True, offset=7)[n:].any()


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,lb,ncols,ncomp,offsets", [(3, 1, 1, 0, [0]), (6, 1, 3, 2, [0, 1, 2, 17]), (10, 2, 5, 3, [0, 1, 4, 15, 16, 513]),
                                                           (12, 1, 6, 6, [0, 1, 2, 3, 8, 70, 2058])])
def test_deep_and_ood_vs_oracle(ctx, oracle, log_n, lb, ncols, ncomp, offsets):
    rng = np.random.default_rng(100 + log_n)
    n, N = 1 << log_n, (1 << log_n) << lb
    trace, lde, coeffs, comp_coeffs, comp_lde, mc, mo, z, zc, ood_t, ood_c, ct, cc = _deep_case(rng, oracle, log_n, lb, ncols, ncomp, offsets)
    rev = np.array([int(format(i, "0%db" % log_n)[::-1], 2) for i in range(n)])
    d_co = [ctx.column(c[rev]) for c in coeffs]                      # bit-reversed, as ss_lde_gl64 leaves them
    assert np.array_equal(ctx.ood_eval_gl64x3(d_co, log_n, mc, mo, z), ood_t)
    # columns read at z only take the one-pass path (a dot product with the power table) instead of three transforms
    at_z = list(range(ncols))
    assert np.array_equal(ctx.ood_eval_gl64x3(d_co, log_n, at_z, [0] * ncols, zc), oracle.gl3_ood_eval(coeffs, at_z, [0] * ncols, zc))
    out = ctx.alloc(24 * N)
    ctx.deep_compose_gl64x3([ctx.column(c) for c in lde], [ctx.column(c) for c in comp_lde], log_n, lb, 7, mc, mo, ood_t, ct, ood_c, cc, z, zc, out)
    assert np.array_equal(out.download(np.uint64, (N, 3)), oracle.gl3_deep_compose(lde, comp_lde, log_n, lb, 7, mc, mo, ood_t, ct, ood_c, cc, z, zc))
    # dishonest out-of-domain values: still the definition (the quotients are then not polynomials; the sub-coset composition +
    # extension agrees with the term-by-term values only on the sub-coset rows)
    bad = ood_t.copy()
    bad[0, 1] = (int(bad[0, 1]) + 1) % GL_P
    ctx.deep_compose_gl64x3([ctx.column(c) for c in lde], [ctx.column(c) for c in comp_lde], log_n, lb, 7, mc, mo, bad, ct, ood_c, cc, z, zc, out)
    want = oracle.gl3_deep_compose(lde, comp_lde, log_n, lb, 7, mc, mo, bad, ct, ood_c, cc, z, zc)
    assert np.array_equal(out.download(np.uint64, (N, 3))[:: 1 << lb], want[:: 1 << lb])


@pytest.mark.gpu
def test_deep_at_the_benchmark_size_feeds_fri(ctx, oracle):
    """2^20-row columns (5 trace + 1 extension-field column = 8 Fp columns, blowup 2): the composed evaluations interpolate to
    degree < n per component, and one FRI fold of them has degree < n / 8"""
    from sandstorm_amd import backend as be
    log_n, lb, ncols = 20, 1, 8
    n, N = 1 << log_n, 2 << log_n
    rng = np.random.default_rng(77)
    cols = [ctx.column(rand_fp(rng, n)) for _ in range(ncols)]
    ev = [ctx.alloc(8 * N) for _ in range(ncols)]
    co = [ctx.alloc(8 * n) for _ in range(ncols)]
    ctx.lde_gl64(cols, log_n, lb, 7, ev, co)
    mask = [(c, o) for c in range(ncols) for o in (0, 1, 2, 16, 33)]
    mc, mo = [c for c, _ in mask], [o for _, o in mask]
    z = rand_fp(rng, 3)
    ood = ctx.ood_eval_gl64x3(co, log_n, mc, mo, z)
    coef = np.stack([rand_fp(rng, 3) for _ in mask])
    out = ctx.alloc(24 * N)
    ctx.deep_compose_gl64x3(ev, [], log_n, lb, 7, mc, mo, ood, coef, np.zeros((0, 3), dtype=np.uint64), np.zeros((0, 3), dtype=np.uint64), z, z, out)
    deep = out.download(np.uint64, (N, 3))
    for t in range(3):
        comp = ctx.column(np.ascontiguousarray(deep[:, t]))
        ctx.ntt_gl64([comp], log_n + lb, be.INVERSE, 7, be.NATURAL, be.NATURAL)
        got = comp.download(np.uint64, (N,))
        assert not got[n:].any() and got[:n].any()
    folded = ctx.alloc(24 * (N // 8))
    ctx.fri_fold_gl64x3(out, log_n + lb, 8, rand_fp(rng, 3), 7, folded)
    f = folded.download(np.uint64, (N // 8, 3))
    comp = ctx.column(np.ascontiguousarray(f[:, 0]))
    ctx.ntt_gl64([comp], log_n + lb - 3, be.INVERSE, pow(7, 8, GL_P), be.NATURAL, be.NATURAL)
    got = comp.download(np.uint64, (N // 8,))
    assert not got[n // 8:].any() and got[:n // 8].any()


# ---- the constraint program over the cubic extension ---------------------------------------------------------------------
def _gl_program(seed, size, ncols):
    """a random expression DAG lowered for this field; every constant gets pseudo-random extension coordinates, and a tail
    exercises INV and the scratch slots"""
    import random
    from sandstorm_amd import air_program as ap
    from tests.test_air_program import random_dag
    prog = ap.lower(random_dag(random.Random(seed), ncols, 2, 5, size), GL_P)
    consts = np.array([[c % GL_P, (c * 0x9E3779B97F4A7C15 + 1) % GL_P, (c * c + 7) % GL_P] for c in prog.consts] + [[3, 1, 4]], dtype=np.uint64)
    code = [int(w) for w in prog.code]
    assert code[-2] & 0xff == 7                                     # ... OUT acc_d
    d = (code[-2] >> 8) & 0xf
    e = (d + 1) % 4
    I = lambda op, dst, kind, payload: [op | (dst << 8) | (kind << 12), payload]
    tail = (I(6, d, 0, prog.n_slots) + I(0, e, 2, len(consts) - 1) + I(1, e, 1, prog.n_slots) + I(5, e, 0, 0) + I(4, e, 5, 0)
            + I(3, e, 3, 1) + I(4, e, 0, d) + I(7, e, 0, 0))        # st; e = (3,1,4) + slot; e = 1/e; e *= x; e = T0[i+1] - e; e *= d; out e
    return np.array(code[:-2] + tail, dtype=np.uint32), consts, prog.n_slots + 1


def _py_program(code, consts, n_slots, tables, desc, lde, log_N, lb, offset, points):
    """the program at a few points, in Python integers"""
    N = 1 << log_N
    w = pow(7, (GL_P - 1) >> log_N, GL_P)
    add = lambda a, b: [(x + y) % GL_P for x, y in zip(a, b)]
    sub = lambda a, b: [(x - y) % GL_P for x, y in zip(a, b)]

    def inv3(a):
        r, base, e = [1, 0, 0], list(a), GL_P ** 3 - 2
        while e:
            if e & 1:
                r = _mul3(r, base)
            base, e = _mul3(base, base), e >> 1
        return r if any(a) else [0, 0, 0]
    out = {}
    for i in points:
        acc, slots, x = [[0, 0, 0] for _ in range(4)], [[0, 0, 0] for _ in range(n_slots)], offset * pow(w, i, GL_P) % GL_P
        for pc in range(len(code) // 2):
            w0, w1 = int(code[2 * pc]), int(code[2 * pc + 1])
            op, d, kind = w0 & 0xff, (w0 >> 8) & 0xf, (w0 >> 12) & 0xf
            src = [0, 0, 0]
            if op <= 4:
                src = (acc[w1 & 3] if kind == 0 else slots[w1] if kind == 1 else [int(v) for v in consts[w1]] if kind == 2
                       else [int(lde[w1 >> 24][(i + ((w1 & 0xffffff) << lb)) % N]), 0, 0] if kind == 3
                       else [int(tables[desc[2 * w1] + i % (1 << desc[2 * w1 + 1])]), 0, 0] if kind == 4 else [x, 0, 0])
            if op == 0: acc[d] = list(src)
            elif op == 1: acc[d] = add(acc[d], src)
            elif op == 2: acc[d] = sub(acc[d], src)
            elif op == 3: acc[d] = sub(src, acc[d])
            elif op == 4: acc[d] = _mul3(acc[d], src)
            elif op == 5: acc[d] = inv3(acc[d])
            elif op == 6: slots[w1] = list(acc[d])
            else: out[i] = list(acc[d])
    return out


def test_program_oracle_is_the_definition(oracle):
    rng = np.random.default_rng(21)
    log_n, lb, ncols = 4, 1, 3
    N = 2 << log_n
    code, consts, n_slots = _gl_program(4, 60, ncols)
    lde = [rand_fp(rng, N) for _ in range(ncols)]
    tables, desc = np.concatenate([rand_fp(rng, 4), rand_fp(rng, 8)]), [0, 2, 4, 3]
    got = oracle.gl3_eval_program(code, consts, n_slots, tables, desc, lde, log_n, lb, 7)
    want = _py_program(code, consts, n_slots, tables, desc, lde, log_n + lb, lb, 7, range(N))
    assert [[int(v) for v in row] for row in got] == [want[i] for i in range(N)]
    assert got.any()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,size,log_n", [(1, 30, 3), (2, 120, 8), (3, 400, 12), (5, 250, 17)])
def test_program_vs_oracle(ctx, oracle, seed, size, log_n):
    rng = np.random.default_rng(seed)
    lb, ncols = 1, 3
    N = 2 << log_n
    code, consts, n_slots = _gl_program(seed, size, ncols)
    lde = [rand_fp(rng, N) for _ in range(ncols)]
    tables, desc = np.concatenate([rand_fp(rng, 4), rand_fp(rng, 8)]), [0, 2, 4, 3]
    out = ctx.alloc(24 * N)
    ctx.eval_quotient_gl64x3(code, consts, n_slots, ctx.column(tables), desc, [ctx.column(c) for c in lde], log_n, lb, 7, out)
    assert np.array_equal(out.download(np.uint64, (N, 3)), oracle.gl3_eval_program(code, consts, n_slots, tables, desc, lde, log_n, lb, 7))
    # a program that reaches outside its inputs is refused before any launch
    from sandstorm_amd._lib import SandstormHipError
    bad = code.copy()
    assert int(bad[-16]) & 0xff == 6                                  # the tail's ST
    bad[-15] = n_slots + 5
    with pytest.raises(SandstormHipError):
        ctx.eval_quotient_gl64x3(bad, consts, n_slots, ctx.column(tables), desc, [ctx.column(c) for c in lde], log_n, lb, 7, out)


# ---- the kernels compose: a small AIR over this field, end to end up to (not including) the commitments -----------------------
@pytest.mark.gpu
def test_the_kernels_compose_into_a_proof_skeleton(ctx, oracle):
    """A two-column AIR (a' = b, b' = a b + a on every row but the last), challenges in Fq3: LDE -> constraint program ->
    the composition is a polynomial of degree < n -> out-of-domain values satisfy the AIR identity at z (what a verifier
    recomputes) -> DEEP composition of trace and composition columns has degree < n -> FRI folds bring it down to a constant.
    Every arrow is a kernel of this file's entry points; the checks are done in Python integers on the host."""
    from sandstorm_amd import backend as be
    log_n, lb = 10, 1
    n, N = 1 << log_n, 2 << log_n
    g_n = pow(7, (GL_P - 1) >> log_n, GL_P)
    a, b = [3], [5]
    for _ in range(n - 1):
        a, b = a + [b[-1]], b + [(a[-1] * b[-1] + a[-1]) % GL_P]
    cols = [ctx.column(np.array(c, dtype=np.uint64)) for c in (a, b)]
    ev, co = [ctx.alloc(8 * N) for _ in range(2)], [ctx.alloc(8 * n) for _ in range(2)]
    ctx.lde_gl64(cols, log_n, lb, 7, ev, co)
    rng = np.random.default_rng(8)
    alpha = [[int(v) for v in rand_fp(rng, 3)] for _ in range(2)]
    last = pow(g_n, n - 1, GL_P)
    # tables: 1 / (x^n - 1) over the LDE domain has period `blowup` (x^n = 7^n w_N^(n i))
    zi = np.array([pow((pow(7, n, GL_P) * pow(pow(7, (GL_P - 1) >> (log_n + lb), GL_P), n * i, GL_P) - 1) % GL_P, GL_P - 2, GL_P) for i in range(2)], dtype=np.uint64)
    I = lambda op, d, kind, payload: [op | (d << 8) | (kind << 12), payload]
    T = lambda col, off: (col << 24) | off
    consts = np.array(alpha + [[last, 0, 0]], dtype=np.uint64)
    code = np.array(
        I(0, 0, 3, T(0, 1)) + I(2, 0, 3, T(1, 0)) + I(0, 3, 2, 0) + I(4, 3, 0, 0)                               # acc3 = alpha0 (a' - b)
        + I(0, 1, 3, T(0, 0)) + I(4, 1, 3, T(1, 0)) + I(1, 1, 3, T(0, 0)) + I(3, 1, 3, T(1, 1))                  # acc1 = b' - (a b + a)
        + I(0, 2, 2, 1) + I(4, 2, 0, 1) + I(1, 3, 0, 2)                                                          # acc3 += alpha1 acc1
        + I(0, 0, 5, 0) + I(2, 0, 2, 2) + I(4, 3, 0, 0) + I(4, 3, 4, 0) + I(7, 3, 0, 0), dtype=np.uint32)        # * (x - last) / (x^n - 1)
    q = ctx.alloc(24 * N)
    ctx.eval_quotient_gl64x3(code, consts, 0, ctx.column(zi), [0, 1], ev, log_n, lb, 7, q)
    qh = q.download(np.uint64, (N, 3))
    comp_ev = [ctx.column(np.ascontiguousarray(qh[:, t])) for t in range(3)]
    comp_co = [ctx.column(np.ascontiguousarray(qh[:, t])) for t in range(3)]
    ctx.ntt_gl64(comp_co, log_n + lb, be.INVERSE, 7, be.NATURAL, be.BITREV)
    for c in comp_co:                                                                 # degree < n: the odd bit-reversed slots are zero
        got = c.download(np.uint64, (N,))
        assert not got[1::2].any() and got[0::2].any()
    # out-of-domain: the verifier's identity at z
    z = [int(v) for v in rand_fp(rng, 3)]
    mask = [(0, 0), (0, 1), (1, 0), (1, 1)]
    mc, mo = [c for c, _ in mask], [o for _, o in mask]
    ood_t = ctx.ood_eval_gl64x3(co, log_n, mc, mo, z)
    ood_q = ctx.ood_eval_gl64x3(comp_co, log_n + lb, [0, 1, 2], [0, 0, 0], z)        # the components, as polynomials over the N-domain
    add = lambda u, v: [(x + y) % GL_P for x, y in zip(u, v)]
    sub = lambda u, v: [(x - y) % GL_P for x, y in zip(u, v)]
    Xk = [[1, 0, 0], [0, 1, 0], [0, 0, 1]]
    q_at_z = [0, 0, 0]
    for k in range(3):
        q_at_z = add(q_at_z, _mul3(Xk[k], ood_q[k]))
    a0, a1, b0, b1 = ([int(v) for v in row] for row in ood_t)
    c0, c1 = sub(a1, b0), sub(b1, add(_mul3(a0, b0), a0))
    num = _mul3(add(_mul3(alpha[0], c0), _mul3(alpha[1], c1)), sub(z, [last, 0, 0]))
    zn = [1, 0, 0]
    for _ in range(n):
        zn = _mul3(zn, z)
    assert _mul3(q_at_z, sub(zn, [1, 0, 0])) == num
    # DEEP: trace cells + the composition (three component columns, coefficients c X^k, out-of-domain value on the first)
    coef_t = rand_fp(rng, 12).reshape(4, 3)
    cq = [int(v) for v in rand_fp(rng, 3)]
    coef_c = np.array([_mul3(cq, Xk[k]) for k in range(3)], dtype=np.uint64)
    ood_c = np.array([q_at_z, [0, 0, 0], [0, 0, 0]], dtype=np.uint64)
    deep = ctx.alloc(24 * N)
    ctx.deep_compose_gl64x3(ev, comp_ev, log_n, lb, 7, mc, mo, ood_t, coef_t, ood_c, coef_c, z, z, deep)
    dh = deep.download(np.uint64, (N, 3))
    for t in range(3):
        c = ctx.column(np.ascontiguousarray(dh[:, t]))
        ctx.ntt_gl64([c], log_n + lb, be.INVERSE, 7, be.NATURAL, be.NATURAL)
        got = c.download(np.uint64, (N,))
        assert not got[n:].any() and got[:n].any()
    # FRI: fold 8, 8, 8, 2 (2^11 -> 2^8 -> 2^5 -> 2^2 -> 2^1 points): the last layer is a constant polynomial
    layer, ll, off = deep, log_n + lb, 7
    for fold in (8, 8, 8, 2):
        nxt = ctx.alloc(24 * ((1 << ll) // fold))
        ctx.fri_fold_gl64x3(layer, ll, fold, rand_fp(rng, 3), off, nxt)
        layer, ll, off = nxt, ll - (fold.bit_length() - 1), pow(off, fold, GL_P)
    fin = layer.download(np.uint64, (2, 3))
    assert np.array_equal(fin[0], fin[1]) and fin.any()


@pytest.mark.gpu
@pytest.mark.parametrize("nseg,seg_len", [(1, 1), (5, 1), (6, 1), (7, 1), (8, 1), (9, 1), (14, 1), (15, 1), (16, 1), (8, 3), (2, 17), (1, 34)])
def test_row_hashing_and_trees_of_8_byte_elements(ctx, nseg, seg_len):
    """ss_hash_rows_gl64 (Keccak-256 / Blake2s-256 / SHA-256 over the rows' little-endian bytes, block and padding boundaries
    included: 8, 9, 16 elements ... against SHA-256's 55 / 56 / 64-byte edges), the Blake2s and SHA-256 trees over the digests,
    ss_gather_rows_gl64 - against hashlib (OpenSSL's SHA-256: the FIPS 180-4 function) and the library's host Keccak"""
    import hashlib
    from sandstorm_amd import backend as be
    from sandstorm_amd.coin import keccak256
    rng = np.random.default_rng(nseg * 100 + seg_len)
    nrows = 64
    segs = [rand_fp(rng, nrows * seg_len) for _ in range(nseg)]
    d_segs = [ctx.column(s) for s in segs]
    row_bytes = lambda i: b"".join(int(s[i * seg_len + e]).to_bytes(8, "little") for s in segs for e in range(seg_len))
    for kind, h in ((be.HASH_KECCAK, keccak256), (be.HASH_SHA256, lambda d: hashlib.sha256(d).digest()), (be.HASH_BLAKE2S, lambda d: hashlib.blake2s(d).digest())):
        out = ctx.alloc(32 * nrows)
        ctx.hash_rows_gl64(d_segs, seg_len, nrows, out, kind)
        got = out.download(np.uint8, (nrows, 32))
        assert [bytes(r) for r in got] == [h(row_bytes(i)) for i in range(nrows)]
        if kind == be.HASH_SHA256:              # MatrixMerkleTreeImpl<Sha256HashFn> (cli/src/main.rs:119): node = SHA-256(left || right)
            nodes = ctx.alloc(64 * nrows)
            root, _ = ctx.merkle_build(be.TREE_SHA256, 0, be.LEAF_DIGEST, out, nrows, nodes)
            level = [hashlib.sha256(row_bytes(i)).digest() for i in range(nrows)]
            while len(level) > 1:
                level = [hashlib.sha256(level[2 * k] + level[2 * k + 1]).digest() for k in range(len(level) // 2)]
            assert root == level[0]
            paths, _ = ctx.merkle_open(nodes, None, nrows, [3, 62])
            assert bytes(paths[0][0]) == hashlib.sha256(row_bytes(2)).digest() and bytes(paths[1][0]) == hashlib.sha256(row_bytes(63)).digest()
    nodes = ctx.alloc(64 * nrows)
    root, _ = ctx.merkle_build(be.TREE_BLAKE2S, 0, be.LEAF_DIGEST, out, nrows, nodes)
    level = [hashlib.blake2s(row_bytes(i)).digest() for i in range(nrows)]
    while len(level) > 1:
        level = [hashlib.blake2s(level[2 * k] + level[2 * k + 1]).digest() for k in range(len(level) // 2)]
    assert root == level[0]
    idx = [0, 5, 63, 17]
    rows = ctx.gather_rows_gl64(d_segs, seg_len, nrows, idx)
    from sandstorm_amd._lib import SandstormHipError
    with pytest.raises(SandstormHipError):
        ctx.gather_rows_gl64(d_segs, seg_len, nrows, [nrows])
    assert rows.shape == (4, nseg, seg_len)
    for q, i in enumerate(idx):
        assert [int(v) for v in rows[q].reshape(-1)] == [int(s[i * seg_len + e]) for s in segs for e in range(seg_len)]
