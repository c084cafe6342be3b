"""Run ONLY against the host build of the device code (tests/test_device_code_on_host.py passes this file to pytest explicitly): DEEP
composition with masks built to sit on the kernel's bookkeeping boundaries - 15 / 16 / 17 / 32 / 33 / 97 / 130 cells in ONE column (a
fused dot product is reduced every 16 terms, the reduced parts every 6), many columns, offsets beyond the trace length (they wrap),
the same cell named twice, no composition columns at all / the maximum of four - both fields, against the oracle's term-by-term sums."""
import numpy as np
import pytest

from tests.test_gpu_parity import g3
from tests.util import P, random_column

GL_P = 2**64 - 2**32 + 1


@pytest.fixture(scope="module")
def ctx():
    from sandstorm_amd.backend import Context
    c = Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def be():
    from sandstorm_amd import backend
    return backend


def masks(rng, n, ncols):
    out = []
    for cells_in_col0 in (1, 15, 16, 17, 32, 33, 97, 130):
        m = [(0, int(o)) for o in rng.choice(4 * n, size=cells_in_col0, replace=cells_in_col0 > 4 * n)]       # offsets beyond n wrap
        m += [(int(c), int(rng.integers(0, n))) for c in rng.integers(0, ncols, size=int(rng.integers(0, 40)))]
        if len(m) > 2:
            m.append(m[1])                                                                                        # a cell named twice
        out.append(m)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,ncols,ncomp", [(6, 1, 0), (7, 3, 2), (8, 10, 4), (9, 16, 1)])
def test_deep_compose_on_the_bookkeeping_boundaries(ctx, be, oracle, log_n, ncols, ncomp):
    rng = np.random.default_rng(100 * log_n + ncols)
    lb = 1
    n, N = 1 << log_n, 1 << (log_n + lb)
    g = g3(oracle)
    cols = [random_column(n, c + 1000 * log_n) for c in range(ncols)]
    ev, co = be.Matrix.from_host(ctx, cols).lde(lb, g)
    comp_coeffs = [random_column(n, 2000 + k) for k in range(ncomp)]
    cm = be.Matrix.from_host(ctx, [np.concatenate([c, np.zeros((N - n, 4), dtype=np.uint64)]) for c in comp_coeffs]) if ncomp else None
    if cm is not None:
        cm.evaluate(g)
    z = int(rng.integers(2, 2**62)) ** 3 % P
    zm = oracle.to_mont([z])[0]
    zc = oracle.to_mont([pow(z, max(1, ncomp), P)])[0]
    for mask in masks(rng, n, ncols):
        mc, mo = [c for c, _ in mask], [o for _, o in mask]
        ood_t = ctx.ood_eval(co.cols, log_n, mc, mo, zm)
        ood_c = np.stack([oracle.poly_eval(c, zc) for c in comp_coeffs]) if ncomp else np.zeros((0, 4), dtype=np.uint64)
        ct = oracle.to_mont([int(v) for v in rng.integers(1, 2**62, size=len(mask))])
        cc = oracle.to_mont([int(v) for v in rng.integers(1, 2**62, size=ncomp)]) if ncomp else np.zeros((0, 4), dtype=np.uint64)
        out = ctx.alloc(32 * N)
        ctx.deep_compose(ev.cols, cm.cols if cm is not None else [], log_n, lb, g, mc, mo, ood_t, ct, ood_c, cc, zm, out)
        want = oracle.deep_compose(ev.to_host(), cm.to_host() if cm is not None else [], log_n, lb, g, mc, mo, ood_t, ct, ood_c, cc, zm)
        assert np.array_equal(out.download(np.uint64, (N, 4)), want), (log_n, ncols, ncomp, len(mask))


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,ncols,ncomp", [(5, 1, 0), (7, 4, 3), (8, 8, 6), (9, 16, 12)])
def test_deep_compose_gl64x3_on_the_boundaries(ctx, oracle, log_n, ncols, ncomp):
    """the 64-bit field's DEEP (wide sums: one reduction per column's taps, per point): long columns, many columns, wrapped offsets,
    doubled cells, 0 / 12 composition columns - and out-of-domain evaluation of the same masks"""
    rng = np.random.default_rng(7 * log_n + ncols)
    lb = 1
    n, N = 1 << log_n, 2 << log_n
    rand = lambda k: rng.integers(0, GL_P, size=k, dtype=np.uint64)
    trace = [rand(n) for _ in range(ncols)]
    trace[0][: n // 2] = np.uint64(GL_P - 1)                                                                    # words at the top of the field
    lde, coeffs = zip(*[oracle.gl_lde(t, lb, 7) for t in trace])
    comp_coeffs = [rand(n) for _ in range(ncomp)]
    comp_lde = [oracle.gl_ntt(np.concatenate([c, np.zeros(N - n, dtype=np.uint64)]), offset=7) for c in comp_coeffs]
    rev = np.array([int(format(i, "0%db" % log_n)[::-1], 2) for i in range(n)])
    d_co = [ctx.column(c[rev]) for c in coeffs]
    z, zc = rand(3), rand(3)
    for mask in masks(rng, n, ncols):
        mask = [(c, o % (1 << 24)) for c, o in mask]
        mc, mo = [c for c, _ in mask], [o for _, o in mask]
        ood_t = oracle.gl3_ood_eval(coeffs, mc, mo, z)
        assert np.array_equal(ctx.ood_eval_gl64x3(d_co, log_n, mc, mo, z), ood_t), (log_n, ncols, len(mask))
        ood_c = oracle.gl3_ood_eval(comp_coeffs, list(range(ncomp)), [0] * ncomp, zc) if ncomp else np.zeros((0, 3), dtype=np.uint64)
        ct = np.stack([rand(3) for _ in mask])
        ct[0] = np.uint64(GL_P - 1)
        cc = np.stack([rand(3) for _ in range(ncomp)]) if ncomp else np.zeros((0, 3), dtype=np.uint64)
        out = ctx.alloc(24 * N)
        ctx.deep_compose_gl64x3([ctx.column(c) for c in lde], [ctx.column(c) for c in comp_lde], log_n, lb, 7, mc, mo, ood_t, ct, ood_c, cc, z, zc, out)
        want = oracle.gl3_deep_compose(lde, comp_lde, log_n, lb, 7, mc, mo, ood_t, ct, ood_c, cc, z, zc)
        assert np.array_equal(out.download(np.uint64, (N, 3)), want), (log_n, ncols, ncomp, len(mask))


@pytest.mark.gpu
def test_constraint_vm_on_many_random_programs(ctx, be, oracle):
    """the interpreter of both fields against the oracle's VM on forty more random expression DAGs than the MI355X suite runs"""
    from tests.test_goldilocks import test_program_vs_oracle
    from tests.test_gpu_parity import test_eval_quotient_vs_oracle
    for seed in range(10, 30):
        test_eval_quotient_vs_oracle(ctx, be, oracle, seed, 40 + 17 * (seed % 9), 3 + seed % 7)
        test_program_vs_oracle(ctx, oracle, seed, 30 + 23 * (seed % 8), 3 + seed % 8)
