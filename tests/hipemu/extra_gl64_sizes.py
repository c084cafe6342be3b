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


@pytest.mark.gpu
@pytest.mark.parametrize("fold", [2, 4, 8, 16])
def test_every_fold_at_every_small_length(ctx, oracle, fold):
    from tests.test_goldilocks import test_fri_fold_vs_oracle
    for log_len in range(fold.bit_length() - 1, 12):
        test_fri_fold_vs_oracle(ctx, oracle, fold, log_len)


@pytest.mark.gpu
def test_every_row_shape_of_8_byte_elements(ctx):
    from tests.test_goldilocks import test_row_hashing_and_trees_of_8_byte_elements
    for nseg in range(1, 17):
        for seg_len in (1, 2, 3, 5, 8, 17):
            if nseg * seg_len <= 64:
                test_row_hashing_and_trees_of_8_byte_elements(ctx, nseg, seg_len)


@pytest.mark.gpu
def test_running_products_of_every_small_length(ctx, oracle):
    """ss_running_product_gl64x3 (reduce / scan / apply levels of 8 items per lane) against the plain loop in Python integers
    (oracle/gl_cpu_context.py), for every count around the level boundaries, strides, single- and two-column terms, output strides"""
    import torch
    from oracle.gl_cpu_context import GlCpuContext
    ref = GlCpuContext()
    rng = np.random.default_rng(12)
    z, alpha = [int(v) for v in rng.integers(1, GL_P, size=3, dtype=np.uint64)], [int(v) for v in rng.integers(1, GL_P, size=3, dtype=np.uint64)]
    for count in list(range(1, 20)) + [63, 64, 65, 511, 512, 513, 4095, 4097]:
        for stride, two_columns, out_stride, out_offset in ((1, False, 1, 0), (2, True, 2, 1), (3, True, 1, 0)):
            a = torch.from_numpy(rng.integers(0, GL_P, size=count * stride + 1, dtype=np.uint64).view(np.int64).copy())
            b = torch.from_numpy(rng.integers(0, GL_P, size=count * stride + 1, dtype=np.uint64).view(np.int64).copy())
            args = (a, a[1:] if two_columns else None, b, b[1:] if two_columns else None, stride, count, z, alpha if two_columns else None)
            got = [torch.zeros(count * out_stride + out_offset, dtype=torch.int64) for _ in range(3)]
            want = [torch.zeros(count * out_stride + out_offset, dtype=torch.int64) for _ in range(3)]
            last = ctx.running_product_gl64x3(*args, got, out_stride, out_offset)
            want_last = ref.running_product_gl64x3(*args, want, out_stride, out_offset)
            assert last == tuple(int(v) for v in want_last), (count, stride, two_columns)
            for g, w in zip(got, want):
                assert torch.equal(g, w), (count, stride, two_columns)
