"""Run ONLY against the host build of the device code (tests/test_device_code_on_host.py passes this file to pytest explicitly):
sweeps over the sizes the MI355X suite samples - every transform size up to 2^16 in both directions and on a coset, every extension
factor, every fold factor at every small layer length, every row width of every hash, every tree size of every tree kind - each
against the oracle, bit for bit."""
import numpy as np
import pytest

from tests.test_gpu_parity import _down, _up, g3
from tests.util import P, random_column


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


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", list(range(1, 17)))
def test_every_transform_size(ctx, be, oracle, log_n):
    n = 1 << log_n
    col = random_column(n, 300 + log_n)
    for off in (None, g3(oracle)):
        d = _up(ctx, [col])
        ctx.ntt(d, log_n, be.FORWARD, off)
        assert np.array_equal(_down(d, n)[0], oracle.ntt(col, offset=off)), (log_n, off is not None)
        ctx.ntt(d, log_n, be.INVERSE, off)
        assert np.array_equal(_down(d, n)[0], col)
    d = _up(ctx, [col])
    ctx.ntt(d, log_n, be.INVERSE, None, be.NATURAL, be.BITREV)
    assert np.array_equal(_down(d, n)[0], oracle.bitrev_permute(oracle.ntt(col, inverse=True)))


@pytest.mark.gpu
@pytest.mark.parametrize("log_blowup", [1, 2, 3, 4])
@pytest.mark.parametrize("log_n", [1, 2, 5, 9, 12, 13, 14])
def test_every_extension_factor(ctx, be, oracle, log_n, log_blowup):
    n = 1 << log_n
    cols = [random_column(n, 500 + c) for c in range(2)]
    ev, co = be.Matrix.from_host(ctx, cols).lde(log_blowup, g3(oracle))
    ev_h, co_h = ev.to_host(), co.to_host()
    for c in range(2):
        want_ev, want_co = oracle.lde(cols[c], log_blowup, g3(oracle))
        assert np.array_equal(ev_h[c], want_ev) and np.array_equal(co_h[c], oracle.bitrev_permute(want_co)), (log_n, log_blowup, c)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_every_row_width(ctx, be, oracle, kind):
    for ncols in range(1, 17):
        for nrows in (1, 3, 64, 257):
            cols = [random_column(nrows, 700 + 16 * kind + c) for c in range(ncols)]
            m = be.Matrix.from_host(ctx, cols)
            got = m.hash_rows(kind).download(np.uint8, (nrows, 32))
            assert np.array_equal(got, oracle.hash_rows(kind, cols)), (kind, ncols, nrows)


@pytest.mark.gpu
@pytest.mark.parametrize("fold", [2, 4, 8, 16])
def test_every_fold_at_every_small_length(ctx, be, oracle, fold):
    log_fold = fold.bit_length() - 1
    alpha = oracle.to_mont([0x1234567890ABCDEF ** 3 % P])[0]
    for log_len in range(log_fold, 12):
        n = 1 << log_len
        ev = random_column(n, 900 + fold + log_len)
        out = ctx.alloc(32 * (n // fold))
        ctx.fri_fold(ctx.column(ev), log_len, fold, alpha, g3(oracle), out)
        assert np.array_equal(out.download(np.uint64, (n // fold, 4)), oracle.fri_fold(ev, fold, alpha, g3(oracle))), (fold, log_len)


@pytest.mark.gpu
@pytest.mark.parametrize("tree,leaf_kind,nf", [(0, 0, 0), (0, 1, 0), (1, 0, 0), (1, 1, 0), (2, 0, 0), (2, 0, 2), (2, 0, 22), (2, 1, 22)])
def test_every_tree_size(ctx, be, oracle, tree, leaf_kind, nf):
    from tests.test_gpu_parity import test_merkle_vs_oracle
    for log_n in range(1, 8 if tree == 2 and nf else 11):         # Pedersen levels: 32 curve additions per node, lane by lane
        test_merkle_vs_oracle(ctx, be, oracle, tree, leaf_kind, nf, log_n)


@pytest.mark.gpu
def test_every_small_poly_eval_ood_and_deep_size(ctx, be, oracle):
    from tests.test_gpu_parity import test_deep_compose_vs_oracle, test_ood_eval_vs_oracle, test_poly_eval_vs_oracle
    for log_n in range(1, 13):
        test_poly_eval_vs_oracle(ctx, be, oracle, log_n)
    for log_n in range(2, 12):
        test_ood_eval_vs_oracle(ctx, be, oracle, log_n)
    for log_n in range(3, 12):
        test_deep_compose_vs_oracle(ctx, be, oracle, log_n)
