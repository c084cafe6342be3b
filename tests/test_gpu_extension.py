"""GPU parity of the extension-trace scans (SURVEY.md §8a row A2 / "next" row X1): ss_permutation_product,
ss_diluted_aggregate and the host mirror of Trace::build_extension_columns against the oracle's restatement of
layouts/src/recursive/trace.rs:699-814 and layouts/src/starknet/trace.rs:997-1100.  Bit-exact."""
import numpy as np
import pytest

from tests.test_extension_cpu import challenges, permuted_trace
from tests.util import P, random_column

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from sandstorm_amd.backend import Context
    c = Context(0)
    yield c
    c.close()


SENTINEL = 0xABCDEF


def _filled(oracle, n):
    out = np.zeros((n, 4), dtype=np.uint64)
    out[:] = oracle.to_mont([SENTINEL])[0]
    return out


@pytest.mark.parametrize("count", [1, 2, 63, 64, 65, 4096, 4097, (1 << 16) + 3])
@pytest.mark.parametrize("shape", ["address_value", "single"])
def test_permutation_product_matches_oracle(ctx, oracle, count, shape):
    z, alpha = random_column(2, 90 + count % 7)
    if shape == "address_value":        # memory: array_chunks::<2>() of two columns, output every 2nd row
        a, b = random_column(2 * count, 11), random_column(2 * count, 12)
        num, den, os_, oo = (a, 2, 0, 1), (b, 2, 0, 1), 2, 0
    else:                               # range check: offsets 0 / 2 of every 4 cells of ONE column, output rows 4k + 1
        a = random_column(4 * count, 13)
        num, den, os_, oo = (a, 4, 0, -1), (a, 4, 2, -1), 4, 1
        b = a
    want = _filled(oracle, count * os_)
    last_want = oracle.permutation_product(num, den, count, z, alpha, want, os_, oo)
    da, db = ctx.column(a), ctx.column(b)
    dout = ctx.column(_filled(oracle, count * os_))
    last = ctx.permutation_product((da,) + num[1:], (db,) + den[1:], count, z, alpha, dout, os_, oo)
    got = dout.download(np.uint64, (count * os_, 4))
    assert np.array_equal(got, want)
    assert np.array_equal(last, last_want)


def test_permutation_product_zero_denominator(ctx, oracle):
    """a zero denominator term: ark-ff batch_inversion semantics (zero stays zero, earlier entries unaffected)"""
    count = 1000
    z = random_column(1, 77)[0]
    a = random_column(4 * count, 14)
    a[4 * 321 + 2] = z                   # z - ordered = 0 at item 321
    want = np.zeros((count, 4), dtype=np.uint64)
    oracle.permutation_product((a, 4, 0, -1), (a, 4, 2, -1), count, z, np.zeros(4, dtype=np.uint64), want)
    assert want[320].any() and not want[321:].any()
    da, dout = ctx.column(a), ctx.alloc(32 * count)
    ctx.zero(dout)
    last = ctx.permutation_product((da, 4, 0, -1), (da, 4, 2, -1), count, z, None, dout)
    assert np.array_equal(dout.download(np.uint64, (count, 4)), want) and not last.any()


@pytest.mark.parametrize("count", [1, 2, 64, 65, 4097, (1 << 16) + 3])
@pytest.mark.parametrize("dense", [True, False])
def test_diluted_aggregate_matches_oracle(ctx, oracle, count, dense):
    stride, off, os_, oo = (1, 0, 1, 0) if dense else (8, 5, 8, 3)
    x = random_column(stride * count, 15)
    z, alpha = random_column(2, 16)
    want = _filled(oracle, count * os_)
    oracle.diluted_aggregate(x, stride, off, count, z, alpha, want, os_, oo)
    dx, dout = ctx.column(x), ctx.column(_filled(oracle, count * os_))
    ctx.diluted_aggregate(dx, stride, off, count, z, alpha, dout, os_, oo)
    assert np.array_equal(dout.download(np.uint64, (count * os_, 4)), want)


@pytest.mark.parametrize("layout", ["recursive", "starknet"])
def test_build_extension_columns_matches_oracle(ctx, oracle, layout):
    from sandstorm_amd import extension as ext
    n = 1 << 12
    host = permuted_trace(oracle, layout, n, seed=5)
    ch = challenges(oracle)
    want, lasts = oracle.build_extension_columns(layout, host, ch, n)
    dev = {k: ctx.column(v) for k, v in host.items()}
    cols = ext.TraceColumns(dev["npc"], dev["memory"], dev["range_check"], n, dev.get("diluted_unordered"), dev.get("diluted_ordered"))
    m = ext.build_extension_columns(layout, ctx, cols, ch)         # check=True: the products close to one
    got = m.to_host()
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    # not a permutation -> the reference's assert, as an exception
    bad = dict(host)
    rc = bad["range_check"].copy()
    rc[2] = oracle.to_mont([12345])[0]
    dev_bad = ctx.column(rc)
    cols_bad = ext.TraceColumns(dev["npc"], dev["memory"], dev_bad, n, dev.get("diluted_unordered"), dev.get("diluted_ordered"))
    with pytest.raises(ext.PermutationCheckError):
        ext.build_extension_columns(layout, ctx, cols_bad, ch)
    assert ext.build_extension_columns(layout, ctx, cols_bad, ch, check=False).nrows == n


@pytest.mark.parametrize("layout", ["recursive", "starknet"])
def test_cpp_host_build_extension_columns(ctx, oracle, layout):
    """the C++ host's Trace::build_extension_columns (sandstorm_amd/host/extension.cpp) against the oracle"""
    from sandstorm_amd import hostlib
    from sandstorm_amd._lib import SandstormHipError
    n = 1 << 11
    host = permuted_trace(oracle, layout, n, seed=9)
    ch = challenges(oracle)
    want, _ = oracle.build_extension_columns(layout, host, ch, n)
    names = ["npc", "memory", "range_check"] + (["diluted_unordered", "diluted_ordered"] if layout == "recursive" else [])
    dev = [ctx.column(host[k]) for k in names]
    m = hostlib.build_extension_columns(ctx, layout, dev, n, ch)
    got = m.to_host()
    m.close()
    assert len(got) == len(want) and all(np.array_equal(g, w) for g, w in zip(got, want))
    rc = host["range_check"].copy()
    rc[2] = oracle.to_mont([4242])[0]
    dev[2] = ctx.column(rc)
    with pytest.raises(SandstormHipError, match="range-check permutation product"):
        hostlib.build_extension_columns(ctx, layout, dev, n, ch)


@pytest.mark.parametrize("dense", [True, False])
@pytest.mark.parametrize("world", [2, 8])
def test_block_scans_compose_to_the_whole_scan(ctx, oracle, world, dense):
    """ABI 12: ss_diluted_aggregate_block + ss_affine_apply and ss_permutation_product + ss_scale_strided on `world` row blocks, the blocks
    before a block folded in on the host as host/extension.cpp build_extension_blocks does - the single scan's cells, bit for bit"""
    count = 512                                         # items per block
    stride, off, os_, oo = (1, 0, 1, 0) if dense else (8, 5, 8, 3)
    total = count * world
    x = random_column(stride * total, 21)
    z, alpha = random_column(2, 22)
    want = _filled(oracle, total * os_)
    oracle.diluted_aggregate(x, stride, off, total, z, alpha, want, os_, oo)
    zc, ac = (int(v) for v in oracle.from_mont(np.stack([z, alpha])))
    xs = [int(v) for v in oracle.from_mont(x[off::stride])]
    value = None
    got = _filled(oracle, total * os_)
    for r in range(world):
        dx = ctx.column(x[r * count * stride:(r + 1) * count * stride])
        maps = ctx.alloc(64 * count)
        m, c = (int(oracle.from_mont(t[None])[0]) for t in ctx.diluted_aggregate_block(dx, stride, off, count, r == 0, z, alpha, maps))
        start = 0
        if r:
            u = (xs[r * count] - xs[r * count - 1]) % P
            start = (value * (1 + zc * u) + ac * u * u) % P
        dout = ctx.column(_filled(oracle, count * os_))
        ctx.affine_apply(maps, count, oracle.to_mont([start])[0], dout, os_, oo)
        got[r * count * os_:(r + 1) * count * os_] = dout.download(np.uint64, (count * os_, 4))
        value = (m * start + c) % P
    assert np.array_equal(got, want)
    # a running product: the block's own product, then times the blocks before it
    a = random_column(4 * total, 23)
    want = _filled(oracle, total * 4)
    oracle.permutation_product((a, 4, 0, -1), (a, 4, 2, -1), total, z, alpha, want, 4, 1)
    got = _filled(oracle, total * 4)
    before = 1
    for r in range(world):
        da = ctx.column(a[4 * r * count:4 * (r + 1) * count])
        dout = ctx.column(_filled(oracle, count * 4))
        last = ctx.permutation_product((da, 4, 0, -1), (da, 4, 2, -1), count, z, None, dout, 4, 1)
        if r:
            ctx.scale_strided(dout, 4, 1, count, oracle.to_mont([before])[0])
        got[4 * r * count:4 * (r + 1) * count] = dout.download(np.uint64, (count * 4, 4))
        before = before * int(oracle.from_mont(last[None])[0]) % P
    assert np.array_equal(got, want)


@pytest.mark.parametrize("layout", ["recursive", "starknet"])
@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_cpp_host_build_extension_blocks(oracle, layout, world):
    """the C++ host's build_extension_blocks (the scans of Trace::build_extension_columns divided over `world` ranks, here threads
    with a context each, one all-gather of the blocks' totals): rank r's matrix = rows [r n / world, (r + 1) n / world) of the
    oracle's extension columns; a column that is no permutation is refused on EVERY rank"""
    import threading
    from sandstorm_amd import backend as be, hostlib
    from sandstorm_amd._lib import SandstormHipError
    n = 1 << 11
    nb = n // world
    host = permuted_trace(oracle, layout, n, seed=9)
    ch = challenges(oracle)
    want, _ = oracle.build_extension_columns(layout, host, ch, n)
    names = ["npc", "memory", "range_check"] + (["diluted_unordered", "diluted_ordered"] if layout == "recursive" else [])
    bad = dict(host)
    bad["range_check"] = host["range_check"].copy()
    bad["range_check"][2] = oracle.to_mont([4242])[0]

    def run(columns, check=True):
        group = hostlib.LocalGroup(world)
        out, errs = [None] * world, [None] * world

        def body(r):
            c = be.Context(0)
            try:
                dev = [c.column(columns[k][r * nb:(r + 1) * nb]) for k in names]
                m = hostlib.build_extension_blocks(c, layout, dev, n, r, world, group, ch, check=check)
                out[r] = m.to_host()
                m.close()
            except BaseException as e:           # noqa: BLE001 - compared below
                errs[r] = e
            finally:
                c.close()
        threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        group.close()
        return out, errs
    out, errs = run(host)
    assert errs == [None] * world, errs
    for r in range(world):
        assert len(out[r]) == len(want)
        for g, w in zip(out[r], want):
            assert np.array_equal(g, w[r * nb:(r + 1) * nb]), (r, world)
    out, errs = run(bad)
    # (every rank computes the same closing value and throws; one that is still inside the gather's last barrier when the first
    # failure marks the thread group learns it as "another rank failed")
    assert all(isinstance(e, SandstormHipError) and ("range-check permutation product" in str(e) or "another rank failed" in str(e)) for e in errs), errs
    assert any("range-check permutation product" in str(e) for e in errs), errs
    out, errs = run(bad, check=False)
    assert errs == [None] * world and all(o is not None for o in out)


def test_extension_scans_at_full_size(ctx, oracle):
    """2^22 items (a 2^23-row memory column, the starknet 2^19-step shape): a true permutation closes to one, and
    sampled neighbours satisfy out_{i+1} * den_{i+1} = out_i * num_{i+1} (big integers)."""
    count = 1 << 22
    rng = np.random.default_rng(3)
    vals = rng.integers(0, 1 << 62, size=(count, 2), dtype=np.uint64)
    order = np.lexsort((vals[:, 1], vals[:, 0]))

    def col(v):                         # small canonical values -> Montgomery limbs (vectorised: x * R mod p via python ints is too slow)
        flat = v.reshape(-1)
        out = np.zeros((flat.shape[0], 4), dtype=np.uint64)
        out[:, 0] = flat                # canonical integers < 2^62, NOT Montgomery images: the kernels only see field elements,
        return out                      # and any 4-limb value < p is one; the check below decodes them the same way
    a, b = col(vals), col(vals[order])
    z, alpha = random_column(2, 33)
    da, db, dout = ctx.column(a), ctx.column(b), ctx.alloc(32 * count)
    last = ctx.permutation_product((da, 2, 0, 1), (db, 2, 0, 1), count, z, alpha, dout)
    assert int(oracle.from_mont(last[None])[0]) == 1
    got = dout.download(np.uint64, (count, 4))
    zc, ac = (int(v) for v in oracle.from_mont(np.stack([z, alpha])))
    for i in [0, 1, 62, 63, 64, 4095, 4096, 123456, count - 2]:
        o0, o1 = (int(v) for v in oracle.from_mont(got[i:i + 2]))
        an, vn, ad, vd = (int(v) for v in oracle.from_mont(np.stack([a[2 * i + 2], a[2 * i + 3], b[2 * i + 2], b[2 * i + 3]])))
        assert o1 * ((zc - (ac * vd + ad)) % P) % P == o0 * ((zc - (ac * vn + an)) % P) % P


def test_extension_scan_argument_errors(ctx):
    from sandstorm_amd._lib import SandstormHipError
    buf = ctx.alloc(32 * 16)
    z = random_column(1, 1)[0]
    with pytest.raises(SandstormHipError):
        ctx.permutation_product((buf, 2, 2, -1), (buf, 2, 0, -1), 4, z, None, buf)          # offset >= stride
    with pytest.raises(SandstormHipError):
        ctx.permutation_product((buf, 2, 0, 1), (buf, 2, 0, 1), 4, z, None, buf)            # (a, v) terms need alpha
    with pytest.raises(SandstormHipError):
        ctx.permutation_product((buf, 2, 0, -1), (buf, 2, 1, -1), 0, z, None, buf)          # empty
    with pytest.raises(SandstormHipError):
        ctx.diluted_aggregate(buf, 1, 0, 4, z, z, buf, 2, 2)                                # out offset >= out stride


def test_extension_column_of_the_reference_proof(ctx, oracle):
    """A2 pinned by reference OUTPUT, on the device (the CPU twin is tests/test_layout_starknet.py::
    test_extension_column_is_the_one_the_references_proof_opens): the six challenges replayed from the transcript of
    `example/array-sum.proof.saved`, the extension column built by the device scans (both hosts'
    build_extension_columns), its LDE over 3<w> by ss_lde_fp252 - the 16 extension leaves that proof opens."""
    import os
    from sandstorm_amd import backend as be, binary, extension, hostlib, public_input, wire
    from sandstorm_amd.coin import PublicCoin
    from sandstorm_amd.layouts import starknet as sk
    from sandstorm_amd.prover import bitrev
    from tests.test_layout_starknet import ROOT, reference_query_positions, starknet_example
    with open(os.path.join(ROOT, "tests", "golden", "reference_array_sum_starknet.proof"), "rb") as f:
        w = wire.parse(f.read())
    states, memory, spi = starknet_example(17)
    coin = PublicCoin(be.COIN_SOLIDITY, public_input.public_coin_seed(spi, be.COIN_SOLIDITY))
    coin.reseed_with_digest(w.base_root)
    challenges_ = [coin.draw() for _ in range(6)]
    natural = [bitrev(p, 22) for p in reference_query_positions(w, spi)]
    cols = hostlib.starknet_base_trace(binary.write_register_states(states), binary.write_memory(memory), spi)
    n = cols[0].shape[0]
    base = be.Matrix.from_host(ctx, cols)
    g = oracle.to_mont([3])[0]
    want = [wire._mont_limbs(int(v)) for v in w.extension_rows]
    # Python host mirror
    m = extension.build_extension_columns("starknet", ctx, sk.trace_columns(ctx, base.cols, n), challenges_)
    lde, _ = m.lde(1, g)
    got = ctx.gather_rows(lde.cols, natural)
    assert [[int(x) for x in r[0]] for r in got] == want
    # C++ host
    hm = hostlib.build_extension_columns(ctx, "starknet", [base.cols[sk.COL_NPC], base.cols[sk.COL_MEMORY], base.cols[sk.COL_RANGE_CHECK]], n, challenges_)
    assert np.array_equal(hm.to_host()[0], m.to_host()[0])
    hm.close()
