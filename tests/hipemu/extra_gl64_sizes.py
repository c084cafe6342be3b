"""Run ONLY against the host build of the device code (tests/test_device_code_on_host.py passes this file to pytest explicitly; its
name keeps it out of the default collection): the 64-bit field's transforms through the C ABI for EVERY size up to 2^17 - every
shape of register group (1-4 stages; first, middle, last, only), contiguous and strided passes - and words at both ends of the
field, against the oracle."""
import numpy as np
import pytest

GL_P = 2**64 - 2**32 + 1


@pytest.fixture(scope="module")
def ctx():
    from sandstorm_amd.backend import Context
    c = Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", list(range(1, 18)))
def test_every_transform_size(ctx, oracle, log_n):
    from sandstorm_amd import backend as be
    rng = np.random.default_rng(1000 + log_n)
    n = 1 << log_n
    cols = [rng.integers(0, GL_P, size=n, dtype=np.uint64) for _ in range(2)]
    cols[1] = np.where(rng.integers(0, 3, size=n) == 0, rng.integers(0, 4, size=n, dtype=np.uint64), np.uint64(GL_P - 1) - rng.integers(0, 4, size=n, dtype=np.uint64))
    rev = np.array([int(format(i, "0%db" % log_n)[::-1], 2) for i in range(n)])
    for off in (1, 7):
        d = [ctx.column(c) for c in cols]
        ctx.ntt_gl64(d, log_n, be.FORWARD, off)
        for c, dc in zip(cols, d):
            assert np.array_equal(dc.download(np.uint64, (n,)), oracle.gl_ntt(c, offset=off)), (log_n, off)
        ctx.ntt_gl64(d, log_n, be.INVERSE, off)
        for c, dc in zip(cols, d):
            assert np.array_equal(dc.download(np.uint64, (n,)), c)
    for lb in (1, 2):
        ev = [ctx.alloc(8 << (log_n + lb)) for _ in cols]
        co = [ctx.alloc(8 * n) for _ in cols]
        ctx.lde_gl64([ctx.column(c) for c in cols], log_n, lb, 7, ev, co)
        for c, e, k in zip(cols, ev, co):
            want_ev, want_co = oracle.gl_lde(c, lb, 7)
            assert np.array_equal(e.download(np.uint64, (n << lb,)), want_ev)
            assert np.array_equal(k.download(np.uint64, (n,))[rev], want_co)
