"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on
the same seeded inputs, against the committed golden vectors, and — at sizes the
oracle cannot reach in seconds — through size-independent properties.

Bar: bit-exact (integer / byte work).  Run with `pytest -m gpu` on an MI355X.
"""
import os
import numpy as np
import pytest

from tests.util import P, felt_int, random_column

pytestmark = pytest.mark.gpu


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


def _up(ctx, cols):
    return [ctx.column(c) for c in cols]


def _down(bufs, n):
    return [b.download(np.uint64, (n, 4)) for b in bufs]


G3 = None


def g3(oracle):
    global G3
    if G3 is None:
        G3 = oracle.to_mont([3])[0]
    return G3


# ----------------------------------------------------------------------------- NTT
@pytest.mark.parametrize("name,n", [("ntt_pedersen512.json", 512), ("ntt_ecdsa256.json", 256)])
def test_ntt_golden_kat(ctx, be, oracle, golden, name, n):
    """The reference's own periodic-column NTT known-answer tests, on the GPU."""
    g = golden(name)
    for axis in ("x", "y"):
        coeffs = oracle.to_mont([int(v) for v in g["coeffs_" + axis]])
        want = oracle.to_mont([int(v) for v in g["evals_" + axis]])
        d = _up(ctx, [coeffs])
        ctx.ntt(d, n.bit_length() - 1, be.FORWARD)
        got = _down(d, n)[0]
        assert np.array_equal(got, want)
        ctx.ntt(d, n.bit_length() - 1, be.INVERSE)
        assert np.array_equal(_down(d, n)[0], coeffs)


def test_ntt_poseidon_round_key_kat(ctx, be, oracle, golden):
    """The reference's 8-point Poseidon round-key NTT known-answer tests, on the GPU."""
    g = golden("ntt_poseidon8.json")
    for k in range(3):
        coeffs = oracle.to_mont([int(v) for v in g["key%d" % k]["coeffs"]])
        want = oracle.to_mont([int(v) for v in g["key%d" % k]["evals"]])
        d = _up(ctx, [coeffs])
        ctx.ntt(d, 3, be.FORWARD)
        assert np.array_equal(_down(d, 8)[0], want)
        ctx.ntt(d, 3, be.INVERSE)
        assert np.array_equal(_down(d, 8)[0], coeffs)


@pytest.mark.parametrize("log_n", [1, 2, 3, 4, 7, 10, 11, 12, 13, 15, 18])
@pytest.mark.parametrize("coset", [False, True])
def test_ntt_vs_oracle(ctx, be, oracle, log_n, coset):
    n = 1 << log_n
    cols = [random_column(n, c) for c in range(2)]
    off = g3(oracle) if coset else None
    d = _up(ctx, cols)
    ctx.ntt(d, log_n, be.FORWARD, off)
    got = _down(d, n)
    for c in range(2):
        assert np.array_equal(got[c], oracle.ntt(cols[c], offset=off)), (log_n, c)
    ctx.ntt(d, log_n, be.INVERSE, off)
    back = _down(d, n)
    for c in range(2):
        assert np.array_equal(back[c], cols[c])


def test_ntt_orders(ctx, be, oracle):
    log_n, n = 12, 4096
    col = random_column(n, 7)
    want = oracle.ntt(col)
    # bit-reversed input
    d = _up(ctx, [oracle.bitrev_permute(col)])
    ctx.ntt(d, log_n, be.FORWARD, None, be.BITREV, be.NATURAL)
    assert np.array_equal(_down(d, n)[0], want)
    # bit-reversed output
    d = _up(ctx, [col])
    ctx.ntt(d, log_n, be.FORWARD, None, be.NATURAL, be.BITREV)
    assert np.array_equal(_down(d, n)[0], oracle.bitrev_permute(want))
    # inverse to bit-reversed coefficients, then forward from them
    d = _up(ctx, [want])
    ctx.ntt(d, log_n, be.INVERSE, None, be.NATURAL, be.BITREV)
    assert np.array_equal(_down(d, n)[0], oracle.bitrev_permute(col))
    ctx.ntt(d, log_n, be.FORWARD, None, be.BITREV, be.NATURAL)
    assert np.array_equal(_down(d, n)[0], want)


def test_ntt_many_columns(ctx, be, oracle):
    """more columns than one launch's pointer table (16)"""
    log_n, n = 9, 512
    cols = [random_column(n, c) for c in range(19)]
    d = _up(ctx, cols)
    ctx.ntt(d, log_n, be.FORWARD, g3(oracle))
    got = _down(d, n)
    for c in range(19):
        assert np.array_equal(got[c], oracle.ntt(cols[c], offset=g3(oracle))), c


def _exchange(parts, R):
    """the equal-split all-to-all between the block layout and the exchanged one (its own inverse): chunk m of rank t's buffer
    becomes chunk t of rank m's"""
    chunk = parts[0].shape[0] // R
    return [np.concatenate([parts[t][m * chunk:(m + 1) * chunk] for t in range(R)]) for m in range(R)]


@pytest.mark.parametrize("log_n,log_ranks", [(2, 1), (4, 2), (6, 3), (7, 1), (9, 2), (10, 3), (12, 1), (13, 3), (14, 2), (16, 3)])
@pytest.mark.parametrize("coset", [False, True])
def test_ntt_spread_over_ranks(ctx, be, oracle, log_n, log_ranks, coset):
    """ss_ntt_shard_fp252: ONE transform over R ranks - local stages on blocks, the cross stages on the exchanged layout (here the
    ranks are buffers of one device and the all-to-all is numpy) - is ss_ntt_fp252 on the gathered array, bit for bit: the
    inverse (natural in, bit-reversed out; the Cooley-Tukey tree over the subgroup, Gentleman-Sande over a coset) and the
    forward transform from the bit-reversed coefficients, with and without the LDE's zero padding"""
    n, R = 1 << log_n, 1 << log_ranks
    off = g3(oracle) if coset else None
    col = random_column(n, 40 + log_n)
    evals = oracle.ntt(col, offset=off)                      # natural order
    # inverse: blocks of evaluations -> blocks of bit-reversed coefficients
    parts = _exchange(np.split(evals, R), R)
    bufs = _up(ctx, parts)
    for m in range(R):
        ctx.ntt_shard([bufs[m]], log_n, log_ranks, m, be.INVERSE, off, be.NTT_PART_CROSS)
    parts = _exchange(_down(bufs, n // R), R)
    bufs = _up(ctx, parts)
    for r in range(R):
        ctx.ntt_shard([bufs[r]], log_n, log_ranks, r, be.INVERSE, off, be.NTT_PART_LOCAL)
    coeff_br = np.concatenate(_down(bufs, n // R))
    assert np.array_equal(coeff_br, oracle.bitrev_permute(col))
    # forward from them, and the LDE's forward half (twice the points, the zero padding never stored)
    for log_expand in (0, 1):
        N, log_N = n << log_expand, log_n + log_expand
        want = oracle.ntt(np.concatenate([col, np.zeros_like(col)]), offset=off) if log_expand else evals
        src = _up(ctx, np.split(coeff_br, R))
        dst = [ctx.alloc(32 * (N // R)) for _ in range(R)]
        for r in range(R):
            ctx.ntt_shard([src[r]], log_N, log_ranks, r, be.FORWARD, off, be.NTT_PART_LOCAL, log_expand, [dst[r]])
        bufs = _up(ctx, _exchange(_down(dst, N // R), R))
        for m in range(R):
            ctx.ntt_shard([bufs[m]], log_N, log_ranks, m, be.FORWARD, off, be.NTT_PART_CROSS)
        got = np.concatenate(_exchange(_down(bufs, N // R), R))
        assert np.array_equal(got, want), log_expand


def test_ntt_spread_over_ranks_takes_any_number_of_columns(ctx, be, oracle):
    """ss_ntt_shard_fp252 with 17 vectors in ONE call (it refused more than 16 where ss_ntt_fp252 / ss_lde_fp252 chunk): each
    column of the call is what a call of its own gives, both parts, both directions"""
    log_n, log_ranks, rank, ncols = 8, 2, 3, 17
    B = (1 << log_n) >> log_ranks
    cols = [random_column(B, 300 + c) for c in range(ncols)]
    off = g3(oracle)
    for direction, part, log_expand in ((be.INVERSE, be.NTT_PART_CROSS, 0), (be.INVERSE, be.NTT_PART_LOCAL, 0), (be.FORWARD, be.NTT_PART_CROSS, 0),
                                        (be.FORWARD, be.NTT_PART_LOCAL, 0), (be.FORWARD, be.NTT_PART_LOCAL, 1)):
        src = [c[:B >> log_expand] for c in cols]
        one_by_one = []
        for c in src:
            d, out = _up(ctx, [c])[0], ctx.alloc(32 * B)
            ctx.ntt_shard([d], log_n, log_ranks, rank, direction, off, part, log_expand, [out])
            one_by_one.append(out.download(np.uint64, (B, 4)))
        ds, outs = _up(ctx, src), [ctx.alloc(32 * B) for _ in range(ncols)]
        ctx.ntt_shard(ds, log_n, log_ranks, rank, direction, off, part, log_expand, outs)
        for c in range(ncols):
            assert np.array_equal(outs[c].download(np.uint64, (B, 4)), one_by_one[c]), (direction, part, log_expand, c)
        if not log_expand:                                   # and in place
            ctx.ntt_shard(ds, log_n, log_ranks, rank, direction, off, part)
            for c in range(ncols):
                assert np.array_equal(ds[c].download(np.uint64, (B, 4)), one_by_one[c]), (direction, part, "in place", c)


def test_uploads_on_the_copy_stream_are_ordered_by_their_tickets(ctx, be, oracle):
    """ss_upload_async / ss_wait_upload (the trace generator's thread uploads a column the moment it is final, the prover's stream
    waits for it before the column's first use): a transform enqueued after the wait sees the uploaded column; a ticket serves one
    wait; ticket 0 and a ticket never issued are refused"""
    from sandstorm_amd._lib import SandstormHipError
    log_n = 12
    cols = [random_column(1 << log_n, 900 + c) for c in range(3)]
    bufs = [ctx.alloc(32 << log_n) for _ in cols]
    tickets = [ctx.upload_async(b, c) for b, c in zip(bufs, cols)]
    assert len(set(tickets)) == 3 and all(t > 0 for t in tickets)
    for t in reversed(tickets):                              # any order
        ctx.wait_upload(t)
    ctx.ntt(bufs, log_n, be.FORWARD, None)
    for b, c in zip(bufs, cols):
        assert np.array_equal(b.download(np.uint64, (1 << log_n, 4)), oracle.ntt(c))
    with pytest.raises(SandstormHipError, match="ticket"):
        ctx.wait_upload(tickets[0])
    for bad in (0, 1 << 40):
        with pytest.raises(SandstormHipError, match="ticket"):
            ctx.wait_upload(bad)
    again = ctx.upload_async(bufs[0], cols[1])               # slots are reused
    ctx.wait_upload(again)
    assert np.array_equal(bufs[0].download(np.uint64, (1 << log_n, 4)), cols[1])


@pytest.mark.parametrize("width,src_pitch,dst_pitch", [(8, 24, 8), (8, 8, 16), (32, 96, 32), (24, 24, 40), (64, 72, 64), (72, 80, 72), (12, 16, 12), (4, 8, 4)])
def test_rows_copied_between_pitches(ctx, width, src_pitch, dst_pitch):
    """ss_dev_copy_2d: a coordinate out of interleaved triples, a column spread over every other slot, a leaf block's comb - rows of whole
    64-bit words go through the library's own kernel, every other shape through the runtime's rectangular copy; both against numpy, at
    offsets inside the buffers, with what lies between the rows of the destination left alone"""
    rng = np.random.default_rng(width * 1000 + src_pitch)
    for rows in (1, 257, 5000):
        so, do = 8 * 3, 8 * 2
        src = rng.integers(0, 256, so + rows * src_pitch, dtype=np.uint8)
        dst0 = rng.integers(0, 256, do + rows * dst_pitch + 16, dtype=np.uint8)
        d_src, d_dst = ctx.alloc(src.nbytes), ctx.alloc(dst0.nbytes)
        d_src.upload(src)
        d_dst.upload(dst0)
        ctx.dev_copy_2d(d_dst, dst_pitch, d_src, src_pitch, width, rows, dst_offset=do, src_offset=so)
        want = dst0.copy()
        for r in range(rows):
            want[do + r * dst_pitch:do + r * dst_pitch + width] = src[so + r * src_pitch:so + r * src_pitch + width]
        assert np.array_equal(d_dst.download(np.uint8, (dst0.nbytes,)), want), (width, rows)


def test_profiled_launches_are_counted_timed_and_clocked(ctx, be, oracle):
    """ss_profile_enable / ss_profile_reset / ss_profile_read / ss_profile_read_clock (what bench.py's stage times, launch counts and
    stage clocks are read from): launches of a family are counted and timed only while profiling is on; at level 2 the stamps around them
    give shader cycles and reference ticks whose ratio is a clock an MI355X can run at (the reference counter ticks at 100 MHz); results do
    not depend on the level"""
    log_n = 14
    col = random_column(1 << log_n, 4242)
    want = oracle.ntt(col)
    buf = ctx.alloc(32 << log_n)
    emulated = os.environ.get("SS_TEST_HIPEMU") == "1"         # (no shader clock to stamp on the host: levels 0 and 1 only)
    from sandstorm_amd._lib import SandstormHipError
    for level in (0, 1) if emulated else (0, 1, 2):
        try:
            ctx.profile(level)
        except SandstormHipError as e:                         # the library found no stream for its monitor wave and said so
            assert level == 2 and "clock monitor" in str(e)
            continue
        ctx.profile_reset()
        for _ in range(3):
            buf.upload(col)
            ctx.ntt([buf], log_n, be.FORWARD, None)
        assert np.array_equal(buf.download(np.uint64, (1 << log_n, 4)), want), level
        ms, launches = ctx.profile_read(be.PROF_NTT_PASS)
        cycles, ref = ctx.profile_read_clock(be.PROF_NTT_PASS)
        if level == 0:
            assert launches == 0 and ms == 0.0 and cycles == 0.0 and ref == 0.0
            continue
        assert launches >= 3 and launches % 3 == 0 and (ms > 0.0 or emulated), (level, launches, ms)    # (the emulated events carry no time)
        assert ctx.profile_read(be.PROF_DEEP) == (0.0, 0)       # a family that did not run
        if level == 1:
            assert cycles == 0.0 and ref == 0.0
        else:
            ghz = cycles / (ref / 100e6) / 1e9
            assert ref > 0 and 0.3 < ghz < 3.5, (cycles, ref, ghz)
        ctx.profile_reset()
        assert ctx.profile_read(be.PROF_NTT_PASS) == (0.0, 0)
    ctx.profile(False)


@pytest.mark.parametrize("ncols", [1, 9, 16, 17, 37])
def test_opened_rows_of_a_column_set(ctx, ncols):
    """ss_gather_rows: the rows at the query positions, row after row (one launch per 16 columns writes them in that order), repeated and
    unsorted positions included"""
    n = 1 << 10
    cols = [random_column(n, 7000 + c) for c in range(ncols)]
    d = _up(ctx, cols)
    idx = np.array([5, 0, n - 1, 5, 77, 512, 513, 1], dtype=np.uint64)
    rows = ctx.gather_rows(d, idx)
    assert rows.shape == (len(idx), ncols, 4)
    for qi, q in enumerate(idx):
        for c in range(ncols):
            assert np.array_equal(rows[qi, c], cols[c][int(q)]), (qi, c)
    assert ctx.gather_rows(d, np.zeros(0, dtype=np.uint64)).shape[0] == 0


def test_query_phase_gathers_in_one_round_trip(ctx, be):
    """ss_gather_batch: every job's entries as the single calls return them - opened rows (ss_gather_rows, also past 16 columns),
    authentication paths and tag bytes (ss_merkle_open: the siblings' node numbers over the node array), leaf digests - with empty jobs
    skipped and outputs of odd byte counts next to each other; malformed jobs are refused"""
    from sandstorm_amd._lib import SandstormHipError
    n = 1 << 9
    cols = [random_column(n, 8100 + c) for c in range(19)]
    d = _up(ctx, cols)
    leaves = np.random.default_rng(5).integers(0, 256, (n, 32), dtype=np.uint8)
    d_leaves = ctx.alloc(32 * n)
    d_leaves.upload(leaves)
    d_nodes, d_tags = ctx.alloc(64 * n), ctx.alloc(2 * n)
    ctx.merkle_build(be.TREE_FRIENDLY, 3, be.LEAF_DIGEST, d_leaves, n, d_nodes, tags=d_tags)
    pos = np.array([3, 500, 3, 0, n - 1], dtype=np.uint64)
    log_n = 9
    sib = np.array([((n + int(q)) >> l) ^ 1 for q in pos for l in range(log_n)], dtype=np.uint64)
    few = np.array([7, 1, 300], dtype=np.uint64)
    outs = ctx.gather_batch([(d, 32, pos), ([d_nodes], 32, sib), ([d_tags], 1, sib), (d[:3], 32, np.zeros(0, dtype=np.uint64)),
                             ([d_leaves], 32, pos), ([d_tags], 1, few), (d[2:4], 32, few)])
    assert np.array_equal(outs[0], ctx.gather_rows(d, pos))
    paths, tags = ctx.merkle_open(d_nodes, d_tags, n, pos)
    assert outs[1].tobytes() == np.asarray(paths).tobytes() and outs[2].tobytes() == np.asarray(tags).tobytes()
    assert outs[3].shape[0] == 0
    assert outs[4].tobytes() == leaves[pos.astype(np.int64)].tobytes()
    all_tags = d_tags.download(np.uint8, (2 * n,))
    assert np.array_equal(outs[5], all_tags[few.astype(np.int64)])
    assert np.array_equal(outs[6], ctx.gather_rows(d[2:4], few))
    assert ctx.gather_batch([]) == []
    for bad in ([(d[:2], 1, few)], [(d[:1], 16, few)]):
        with pytest.raises(SandstormHipError):
            ctx.gather_batch(bad)


def test_fri_fold_rows_of_a_layer(ctx, be, oracle):
    """ss_fri_fold_rows: a layer folded range by range (each range's entries column after column) is the layer folded whole"""
    log_len, fold = 12, 8
    n, rows = 1 << log_len, (1 << log_len) // 8
    ev = random_column(n, 77)
    alpha, off = oracle.to_mont([987654321])[0], g3(oracle)
    for flags in (0, be.FRI_UNNORMALISED):
        want = oracle.fri_fold(ev, fold, alpha, off, flags)
        for R in (1, 2, 4, 8):
            cnt = rows // R
            got = []
            for m in range(R):
                local = np.concatenate([ev[k * rows + m * cnt:k * rows + (m + 1) * cnt] for k in range(fold)])
                d, out = ctx.column(local), ctx.alloc(32 * cnt)
                ctx.fri_fold_rows(d, log_len, fold, alpha, off, m * cnt, cnt, out, flags)
                got.append(out.download(np.uint64, (cnt, 4)))
            assert np.array_equal(np.concatenate(got), want), (flags, R)


@pytest.mark.parametrize("log_n,log_blowup,ncols", [(4, 1, 2), (10, 1, 3), (11, 1, 1), (12, 2, 2), (14, 1, 10), (16, 1, 2)])
def test_lde_vs_oracle(ctx, be, oracle, log_n, log_blowup, ncols):
    n = 1 << log_n
    cols = [random_column(n, c + 100) for c in range(ncols)]
    m = be.Matrix.from_host(ctx, cols)
    ev, co = m.lde(log_blowup, g3(oracle))
    ev_h, co_h = ev.to_host(), co.to_host()
    for c in range(ncols):
        want_ev, want_co = oracle.lde(cols[c], log_blowup, g3(oracle))
        assert np.array_equal(ev_h[c], want_ev), c
        assert np.array_equal(co_h[c], oracle.bitrev_permute(want_co)), c
    # without keeping coefficients
    ev2, none = m.lde(log_blowup, g3(oracle), keep_coeffs=False)
    assert none is None
    assert np.array_equal(ev2.to_host()[0], ev_h[0])


def test_ntt_2_20_vs_oracle(ctx, be, oracle):
    """recursive layout at 2^16 steps: n = 2^20 rows (BASELINE configs[1])"""
    log_n, n = 20, 1 << 20
    col = random_column(n, 1)
    d = _up(ctx, [col])
    ctx.ntt(d, log_n, be.FORWARD, g3(oracle))
    assert np.array_equal(_down(d, n)[0], oracle.ntt(col, offset=g3(oracle)))


def test_ntt_large_impulse_and_roundtrip(ctx, be, oracle):
    """2^24 points (starknet layout at 2^20 steps): forward(x + d*e_j) - forward(x) is the
    impulse response d * (g w^k)^j at sampled k, and inverse(forward(x)) == x."""
    log_n, n = 24, 1 << 24
    base = random_column(n, 1)
    js = [0, 1, 4097, n // 2 + 3, n - 1]
    delta = 0x1234567
    cols = [base]
    for j in js:
        c = base.copy()
        c[j] = oracle.to_mont([(int(oracle.from_mont(base[j])) + delta) % P])[0]
        cols.append(c)
    d = _up(ctx, cols)
    ctx.ntt(d, log_n, be.FORWARD, g3(oracle))
    w = pow(3, (P - 1) >> log_n, P)
    ks = [0, 1, 2, 2048, 2049, 1 << 18, n // 2, n - 1]
    idx = np.array(ks, dtype=np.uint64)
    rows = ctx.gather_rows(d, idx)                       # (len(ks), ncols, 4)
    for qi, k in enumerate(ks):
        f0 = int(oracle.from_mont(rows[qi, 0]))
        x = 3 * pow(w, k, P) % P
        for ci, j in enumerate(js):
            fj = int(oracle.from_mont(rows[qi, 1 + ci]))
            assert (fj - f0) % P == delta * pow(x, j, P) % P, (k, j)
    ctx.ntt(d[:2], log_n, be.INVERSE, g3(oracle))
    back = _down(d[:2], n)
    assert np.array_equal(back[0], cols[0]) and np.array_equal(back[1], cols[1])


def test_lde_large_consistency(ctx, be, oracle):
    """2^20-row trace, blowup 2 (BASELINE config #2): the LDE interpolates back to a
    degree < n polynomial whose coefficients are the kept ones, and the trace-domain
    evaluations reappear at the right coset when offset = 1."""
    log_n, n = 20, 1 << 20
    col = random_column(n, 42)
    m = be.Matrix.from_host(ctx, [col])
    ev, co = m.lde(1, g3(oracle))
    back = be.Matrix(ctx, [ev.cols[0]], 2 * n)
    back.interpolate(g3(oracle), out_order=be.BITREV)          # bit-reversed coefficients of length 2n
    c2 = back.to_host()[0]
    # bit-reversed layout: position 2q holds coefficient bitrev(q) (< n), odd positions the top half
    assert not np.any(c2[1::2])
    assert np.array_equal(c2[0::2], co.to_host()[0])
    ev1, _ = m.lde(1, oracle.to_mont([1])[0], keep_coeffs=False)
    assert np.array_equal(ev1.to_host()[0][0::2], col)


# ------------------------------------------------------------------------- hashing
@pytest.mark.parametrize("kind", [0, 1, 2, 3])
@pytest.mark.parametrize("ncols", [1, 2, 3, 4, 7, 8, 9, 10, 16])
def test_hash_rows_vs_oracle(ctx, be, oracle, kind, ncols):
    n = 300 if ncols % 2 else 1024                     # a ragged and a full grid
    cols = [random_column(n, c + 7 * kind) for c in range(ncols)]
    m = be.Matrix.from_host(ctx, cols)
    got = m.hash_rows(kind).download(np.uint8, (n, 32))
    assert np.array_equal(got, oracle.hash_rows(kind, cols))


def test_hash_rows_large_checksum(ctx, be, oracle):
    """2^20 rows x 7 columns (recursive base trace): spot rows against the oracle and
    an order-independent XOR checksum against a second GPU run on a permuted copy."""
    n = 1 << 20
    cols = [random_column(n, c) for c in range(7)]
    m = be.Matrix.from_host(ctx, cols)
    got = m.hash_rows(be.HASH_BLAKE2S_M20).download(np.uint8, (n, 32))
    sel = np.array([0, 1, 255, 256, 65535, n - 1])
    want = oracle.hash_rows(be.HASH_BLAKE2S_M20, [c[sel] for c in cols])
    assert np.array_equal(got[sel], want)
    perm = np.random.default_rng(0).permutation(n)
    m2 = be.Matrix.from_host(ctx, [c[perm] for c in cols])
    got2 = m2.hash_rows(be.HASH_BLAKE2S_M20).download(np.uint8, (n, 32))
    assert np.array_equal(got2, got[perm])


@pytest.mark.parametrize("tree,leaf_kind,nf", [(0, 0, 0), (0, 1, 0), (1, 0, 0), (1, 1, 0),
                                              (2, 0, 0), (2, 0, 1), (2, 0, 3), (2, 0, 22), (2, 1, 22)])
@pytest.mark.parametrize("log_n", [1, 3, 8])
def test_merkle_vs_oracle(ctx, be, oracle, tree, leaf_kind, nf, log_n):
    n = 1 << log_n
    if leaf_kind == 0:
        kind = tree if tree < 2 else be.HASH_BLAKE2S_M20
        leaves = oracle.hash_rows(kind, [random_column(n, 0), random_column(n, 1)])
    else:
        leaves = random_column(n, 5)
    want_nodes, want_tags = oracle.merkle_build(tree, nf, leaf_kind, leaves)
    d_leaves = ctx.alloc(32 * n).upload(leaves)
    nodes, tags = ctx.alloc(64 * n), ctx.alloc(2 * n)
    root, root_tag = ctx.merkle_build(tree, nf, leaf_kind, d_leaves, n, nodes, tags)
    got_nodes = nodes.download(np.uint8, (2 * n, 32))
    assert np.array_equal(got_nodes[1:], want_nodes[1:])
    assert root == bytes(want_nodes[1])
    if tree == 2:
        assert np.array_equal(tags.download(np.uint8, (2 * n,))[1:], want_tags[1:])
        assert root_tag == want_tags[1]
    # openings
    idx = sorted(set([0, n - 1, n // 2, min(3, n - 1)]))
    paths, ptags = ctx.merkle_open(nodes, tags, n, idx)
    for q, i in enumerate(idx):
        k = n + i
        for lvl in range(log_n):
            assert bytes(paths[q, lvl]) == bytes(want_nodes[k ^ 1])
            k >>= 1


def test_friendly_merkle_large_and_small_levels(ctx, be, oracle):
    """A 2^15-leaf all-Pedersen tree crosses both Pedersen paths: one lane per hash (+ the batched inversion) on the level of 16384
    hashes, 32 lanes per hash (one window each, additions and inversion dealt out to quads of lanes) from 8192 hashes down."""
    n = 1 << 15
    leaves = oracle.hash_rows(be.HASH_BLAKE2S_M20, [random_column(n, 11), random_column(n, 12)])
    want_nodes, want_tags = oracle.merkle_build(2, 22, 0, leaves)
    d_leaves = ctx.alloc(32 * n).upload(leaves)
    nodes, tags = ctx.alloc(64 * n), ctx.alloc(2 * n)
    root, _ = ctx.merkle_build(2, 22, 0, d_leaves, n, nodes, tags)
    assert np.array_equal(nodes.download(np.uint8, (2 * n, 32))[1:], want_nodes[1:])
    assert root == bytes(want_nodes[1])


def test_pedersen_levels_on_sparse_digests(ctx, be, oracle):
    """The 32-lanes-per-hash kernel gives every lane ONE window of a scalar: a zero digit is a lane that holds the point at infinity,
    and the butterfly's additions meet it on either side, on both, and all the way up (a scalar of zero).  Leaves with few non-zero
    windows - 0, 1, single bits at and around the 16 / 18 / 20 / 22 / 24-bit window boundaries, all-ones runs, p - 1 - in every pairing
    with each other and with full-width digests."""
    vals = [0, 1, 2, P - 1, P - 2, (1 << 251) - 1, 1 << 251, (1 << 248) - 1]
    for w in (16, 18, 20, 22, 24):
        vals += [1 << w, (1 << w) - 1, 1 << (2 * w), (1 << (10 * w)) | 1, ((1 << w) - 1) << (3 * w)]
    vals += [1 << k for k in (31, 32, 63, 64, 127, 128, 191, 192, 239, 240, 247, 250)]
    rnd = [int.from_bytes(bytes(r), "big") % P for r in oracle.hash_rows(be.HASH_BLAKE2S_M20, [random_column(16, 77)])]
    vals += rnd
    pairs = [(a, b) for a in vals for b in (vals[:6] + rnd[:2])] + [(b, a) for a in vals for b in vals[:4]] + [(a, a) for a in vals]
    n = 1
    while n < 2 * len(pairs):
        n <<= 1
    flat = [v for ab in pairs for v in ab] + [0] * (n - 2 * len(pairs))
    leaves = np.frombuffer(b"".join(v.to_bytes(32, "big") for v in flat), dtype=np.uint8).reshape(n, 32).copy()
    want_nodes, want_tags = oracle.merkle_build(2, 22, 0, leaves)
    d_leaves = ctx.alloc(32 * n).upload(leaves)
    nodes, tags = ctx.alloc(64 * n), ctx.alloc(2 * n)
    root, _ = ctx.merkle_build(2, 22, 0, d_leaves, n, nodes, tags)
    assert np.array_equal(nodes.download(np.uint8, (2 * n, 32))[1:], want_nodes[1:])
    assert root == bytes(want_nodes[1])


@pytest.mark.parametrize("kind", [1, 3])
def test_hash_rows_bitrev_order(ctx, be, oracle, kind):
    """ss_hash_rows_ex(SS_ORDER_BITREV): digest i is the digest of row bitrev(i) - the commitment order pinned by
    the reference's proof - computed from the natural-order matrix."""
    log_n, n = 11, 1 << 11
    cols = [random_column(n, 80 + c) for c in range(3)]
    m = be.Matrix.from_host(ctx, cols)
    got = m.hash_rows(kind, be.BITREV).download(np.uint8, (n, 32))
    nat = oracle.hash_rows(kind, cols)
    perm = [int(format(i, "0%db" % log_n)[::-1], 2) for i in range(n)]
    assert np.array_equal(got, nat[perm])


@pytest.mark.parametrize("tree", [1, 2])
def test_single_column_tree_bitrev_order(ctx, be, oracle, tree):
    """ss_merkle_build_ex(leaf_order = BITREV) on raw-element leaves == the natural build of the permuted column."""
    log_n, n = 9, 1 << 9
    col = random_column(n, 91)
    perm = [int(format(i, "0%db" % log_n)[::-1], 2) for i in range(n)]
    want_nodes, _ = oracle.merkle_build(tree, 22, 1, col[perm])
    nodes, tags = ctx.alloc(64 * n), ctx.alloc(2 * n)
    root, _ = ctx.merkle_build(tree, 22, 1, ctx.column(col), n, nodes, tags, be.BITREV)
    assert np.array_equal(nodes.download(np.uint8, (2 * n, 32))[1:], want_nodes[1:])
    assert root == bytes(want_nodes[1])


def test_merkle_openings_of_reference_proof(ctx, be, oracle, golden):
    """The reference's saved proof on the GPU: ss_hash_rows reproduces the leaf digests of the opened rows and
    ss_merkle_build (two leaves at a time, up each authentication path) reproduces the proof's roots."""
    g = golden("saved_proof_openings.json")
    roots = g["roots"]

    def rowhash(vals_hex):
        m = be.Matrix.from_host(ctx, [oracle.to_mont([int(v, 16)]) for v in vals_hex])
        return bytes(m.hash_rows(be.HASH_KECCAK_M20).download(np.uint8, (1, 32))[0])

    pair_leaves, nodes = ctx.alloc(64), ctx.alloc(128)

    def climb(cur, path, pos):
        for lvl, sib in enumerate(path):
            a, b = (cur, sib) if ((pos >> lvl) & 1) == 0 else (sib, cur)
            pair_leaves.upload(np.frombuffer(a + b, dtype=np.uint8))
            root, _ = ctx.merkle_build(be.TREE_KECCAK_M20, 0, 0, pair_leaves, 2, nodes)
            cur = root
        return cur

    for q in g["queries"][:2]:
        p = q["position"]
        for name in ("base", "composition"):
            leaf = rowhash(q[name]["row"])
            assert climb(leaf, [bytes.fromhex(d) for d in q[name]["path"]], p).hex() == roots[name]
        f = q["fri"][0]
        assert climb(rowhash(f["row"]), [bytes.fromhex(d) for d in f["path"]], f["position"]).hex() == roots["fri_layers"][0]
        # single-column tree: the leaf level hashes raw elements (UnhashedLeafConfig)
        pair = [q["extension"]["leaf"], q["extension"]["sibling"]]
        pair = pair if (p & 1) == 0 else pair[::-1]
        felts = ctx.column(oracle.to_mont([int(v, 16) for v in pair]))
        first, _ = ctx.merkle_build(be.TREE_KECCAK_M20, 0, 1, felts, 2, nodes)
        assert climb(first, [bytes.fromhex(d) for d in q["extension"]["path"]], p >> 1).hex() == roots["extension"]


def test_merkle_tree_classes(ctx, be, oracle):
    """from_matrix / root / prove on the reference's 8-row test matrices (merkle/mod.rs:455-634)."""
    col = oracle.to_mont(list(range(8)))
    for cls, tree, nf, kind in ((be.LeafVariantMerkleTreeUnmasked, 0, 0, 0), (be.LeafVariantMerkleTree, 1, 0, 1),
                                (be.FriendlyMerkleTree, 2, 22, 3)):
        for ncols in (1, 2):
            m = be.Matrix.from_host(ctx, [col] * ncols)
            t = cls.from_matrix(m)
            if ncols == 1:
                want, _ = oracle.merkle_build(tree, nf, 1, col)
            else:
                want, _ = oracle.merkle_build(tree, nf, 0, oracle.hash_rows(kind, [col] * ncols))
            assert t.root() == bytes(want[1])
            paths, _ = t.prove([3])
            assert bytes(paths[0, 0]) == bytes(want[(8 + 3) ^ 1])


# ------------------------------------------------------------------------ Pedersen
def test_pedersen_golden_and_oracle(ctx, be, oracle, golden):
    g = golden("pedersen.json")
    cases = g["hash_examples"] + g["extra"]
    a = oracle.to_mont([int(c["a"]) for c in cases])
    b = oracle.to_mont([int(c["b"]) for c in cases])
    ra, rb = random_column(200, 11), random_column(200, 12)
    A, B = np.concatenate([a, ra]), np.concatenate([b, rb])
    n = A.shape[0]
    out = ctx.alloc(32 * n)
    ctx.pedersen_hash(ctx.column(A), ctx.column(B), n, out)
    got = out.download(np.uint64, (n, 4))
    for i, c in enumerate(cases):
        assert oracle.from_mont(got[i]) == int(c["hash"]), c
    for i in range(len(cases), n):
        assert np.array_equal(got[i], oracle.pedersen_hash(A[i], B[i])), i


# ----------------------------------------------------------------------------- FRI
@pytest.mark.parametrize("fold", [2, 4, 8, 16])
@pytest.mark.parametrize("log_len", [4, 9, 13])
def test_fri_fold_vs_oracle(ctx, be, oracle, fold, log_len):
    n = 1 << log_len
    ev = random_column(n, 3 + fold)
    alpha = oracle.to_mont([0x1234567890ABCDEF ** 3 % P])[0]
    off = g3(oracle)
    out = ctx.alloc(32 * (n // fold))
    ctx.fri_fold(ctx.column(ev), log_len, fold, alpha, off, out)
    got = out.download(np.uint64, (n // fold, 4))
    assert np.array_equal(got, oracle.fri_fold(ev, fold, alpha, off))


@pytest.mark.parametrize("flags", [1, 2, 3])
@pytest.mark.parametrize("log_len", [3, 9, 13])
def test_fri_fold_conventions_vs_oracle(ctx, be, oracle, flags, log_len):
    n = 1 << log_len
    ev = random_column(n, 41 + flags)
    alpha = oracle.to_mont([0x1234567890ABCDEF ** 3 % P])[0]
    off = g3(oracle)
    out = ctx.alloc(32 * (n // 8))
    ctx.fri_fold(ctx.column(ev), log_len, 8, alpha, off, out, flags)
    assert np.array_equal(out.download(np.uint64, (n // 8, 4)), oracle.fri_fold(ev, 8, alpha, off, flags))


def test_fri_fold_matches_reference_proofs(ctx, be, oracle, golden):
    """The fold known-answer vectors recovered from the reference's shipped proof files, on the GPU."""
    g = golden("fri_saved_proofs.json")
    one = oracle.to_mont([1])[0]
    out = ctx.alloc(32)
    for v in g["vectors"]:
        conv = g["conventions"][v["convention"]]
        flags = (be.FRI_BITREV_ROWS if conv["bitrev_rows"] else 0) | (be.FRI_UNNORMALISED if conv["scale"] == 8 else 0)
        row = oracle.to_mont([int(x, 16) for x in v["values"]])
        beta = oracle.to_mont([int(v["beta"], 16)])[0]
        ctx.fri_fold(ctx.column(row), 3, 8, beta, one, out, flags)
        got = oracle.from_mont(out.download(np.uint64, (1, 4)))
        assert got[0] == int(v["next"], 16), (v["file"], v["layer"], v["row"])


def test_fri_fold_large_degree_property(ctx, be, oracle):
    """2^21 evaluations of a degree < 2^18 polynomial fold (x8) to evaluations of a degree < 2^15 one."""
    log_len, n = 21, 1 << 21
    coeffs = np.zeros((n, 4), dtype=np.uint64)
    coeffs[: 1 << 18] = random_column(1 << 18, 9)
    d = _up(ctx, [coeffs])
    ctx.ntt(d, log_len, be.FORWARD, g3(oracle))
    out = ctx.alloc(32 * (n // 8))
    alpha = oracle.to_mont([77])[0]
    ctx.fri_fold(d[0], log_len, 8, alpha, g3(oracle), out)
    ctx.ntt([out], log_len - 3, be.INVERSE, oracle.to_mont([pow(3, 8, P)])[0])
    c2 = out.download(np.uint64, (n // 8, 4))
    assert not np.any(c2[1 << 15:])
    cs = oracle.from_mont(coeffs[:64])
    for m in range(8):
        assert oracle.from_mont(c2[m]) == sum(pow(77, k, P) * int(cs[8 * m + k]) for k in range(8)) % P


# ----------------------------------------------------------------------------- PoW
@pytest.mark.parametrize("kind", [0, 1])
def test_pow_grind_vs_oracle(ctx, be, oracle, kind):
    digest = bytes((5 * i + kind) & 0xff for i in range(32))
    coin = oracle.Coin(kind, digest)
    for bits in (1, 8, 16):
        assert ctx.pow_grind(kind, digest, bits) == coin.grind(bits), bits


# ------------------------------------------------------------------------ D1 / Q1
@pytest.mark.parametrize("log_n", [1, 3, 6, 7, 12, 16])
def test_poly_eval_vs_oracle(ctx, be, oracle, log_n):
    n = 1 << log_n
    cols = [random_column(n, c + 30) for c in range(3)]
    x = oracle.to_mont([0xDEADBEEFCAFE ** 4 % P])[0]
    d = _up(ctx, [oracle.bitrev_permute(c) for c in cols])
    got = ctx.poly_eval(d, log_n, x)
    for c in range(3):
        assert np.array_equal(got[c], oracle.poly_eval(cols[c], x)), (log_n, c)


@pytest.mark.parametrize("log_n", [4, 11, 14])
def test_ood_eval_vs_oracle(ctx, be, oracle, log_n):
    n = 1 << log_n
    cols = [random_column(n, c + 40) for c in range(4)]
    m = be.Matrix.from_host(ctx, cols)
    _, co = m.lde(1, g3(oracle))
    coeffs = [oracle.lde(c, 1, g3(oracle))[1] for c in cols]
    mask = [(0, 0), (0, 1), (1, 0), (2, 3), (3, n - 1), (1, 7 % n), (3, 0)]
    z = 0xABCDEF0123456789 ** 3 % P
    got = ctx.ood_eval(co.cols, log_n, [c for c, _ in mask], [o for _, o in mask], oracle.to_mont([z])[0])
    w = pow(3, (P - 1) >> log_n, P)
    for j, (c, o) in enumerate(mask):
        want = oracle.poly_eval(coeffs[c], oracle.to_mont([z * pow(w, o, P) % P])[0])
        assert np.array_equal(got[j], want), (log_n, j)


@pytest.mark.parametrize("log_n,block_log", [(12, None), (14, None), (13, 0), (13, 1), (13, 2), (13, 3), (13, 4)])
def test_ood_eval_of_a_layout_sized_mask(ctx, be, oracle, log_n, block_log, monkeypatch):
    """ss_ood_eval of large columns evaluates point by point (deep.hip: the first S stages of the forward network on
    blocks of 2^S coefficients in registers, then one fused Horner tree per point): columns with 2, 7 and 40 distinct offsets take
    S = 0, 2 and 4; SS_OOD_BLOCK_LOG forces every S; the values are the oracle's and the transform path's (SS_OOD_TRANSFORM=1),
    duplicated cells and offsets beyond the trace length (taken mod n) included"""
    n = 1 << log_n
    rng = np.random.default_rng(log_n * 7 + (block_log or 0))
    cols = [random_column(n, c + 300) for c in range(4)]
    m = be.Matrix.from_host(ctx, cols)
    _, co = m.lde(1, g3(oracle))
    coeffs = [oracle.lde(c, 1, g3(oracle))[1] for c in cols]
    offs = [[0, 5], [0, 1, 2, 3, 16, 255, n - 1], sorted({int(v) for v in rng.integers(0, min(n, 40000), size=44)} | {0, 1, n - 1})[:40], [7]]
    mask = [(c, o) for c in range(4) for o in offs[c]] + [(1, 16), (2, offs[2][3] + n), (0, 5)]
    z = 0x1F2E3D4C5B6A7988 ** 3 % P
    zm = oracle.to_mont([z])[0]
    monkeypatch.setenv("SS_OOD_SPARSE_MIN_LOG", "12")           # the library takes this path from 2^20 coefficients on
    if block_log is not None:
        monkeypatch.setenv("SS_OOD_BLOCK_LOG", str(block_log))
    got = ctx.ood_eval(co.cols, log_n, [c for c, _ in mask], [o for _, o in mask], zm)
    monkeypatch.setenv("SS_OOD_TRANSFORM", "1")
    assert np.array_equal(got, ctx.ood_eval(co.cols, log_n, [c for c, _ in mask], [o for _, o in mask], zm))
    w = pow(3, (P - 1) >> log_n, P)
    for j, (c, o) in enumerate(mask):
        if j % 5 == 0 or j >= len(mask) - 3:
            want = oracle.poly_eval(coeffs[c], oracle.to_mont([z * pow(w, o % n, P) % P])[0])
            assert np.array_equal(got[j], want), (log_n, j)


@pytest.mark.parametrize("log_n", [4, 10, 13])
def test_deep_compose_vs_oracle(ctx, be, oracle, log_n):
    lb = 1
    n, N = 1 << log_n, 1 << (log_n + lb)
    g = g3(oracle)
    cols = [random_column(n, c + 70) for c in range(3)]
    m = be.Matrix.from_host(ctx, cols)
    ev, co = m.lde(lb, g)
    comp_coeffs = [random_column(n, 90 + k) for k in range(2)]
    cm = be.Matrix.from_host(ctx, [np.concatenate([c, np.zeros((N - n, 4), dtype=np.uint64)]) for c in comp_coeffs])
    cm.evaluate(g)
    z = 0x1357924680ACE ** 5 % P
    zm = oracle.to_mont([z])[0]
    mask = [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0), (2, 5 % n), (0, n - 1), (2, 1)]
    mc, mo = [c for c, _ in mask], [o for _, o in mask]
    ood_t = ctx.ood_eval(co.cols, log_n, mc, mo, zm)
    ood_c = np.stack([oracle.poly_eval(c, oracle.to_mont([z * z % P])[0]) for c in comp_coeffs])
    alpha = 987654321987654321
    ct = oracle.to_mont([pow(alpha, j, P) for j in range(len(mask))])
    cc = oracle.to_mont([pow(alpha, len(mask) + k, P) for k in range(2)])
    out = ctx.alloc(32 * N)
    ctx.deep_compose(ev.cols, cm.cols, log_n, lb, g, mc, mo, ood_t, ct, ood_c, cc, zm, out)
    got = out.download(np.uint64, (N, 4))
    want = oracle.deep_compose(ev.to_host(), cm.to_host(), log_n, lb, g, mc, mo, ood_t, ct, ood_c, cc, zm)
    assert np.array_equal(got, want)
    # ss_deep_prepare (ABI 12): the denominator tables queued ahead for this point are the ones the composer uses - the same values;
    # tables prepared for ANOTHER point (or size, or offset) are not
    for prepared_z, prepared_log in ((zm, log_n), (oracle.to_mont([z + 1])[0], log_n), (zm, log_n + 1)):
        ctx.deep_prepare(2, prepared_log, g, prepared_z)
        out2 = ctx.alloc(32 * N)
        ctx.deep_compose(ev.cols, cm.cols, log_n, lb, g, mc, mo, ood_t, ct, ood_c, cc, zm, out2)
        assert np.array_equal(out2.download(np.uint64, (N, 4)), want)
    # consistent OOD values => the DEEP quotient is a polynomial of degree < n
    ctx.ntt([out], log_n + lb, be.INVERSE, g)
    assert not np.any(out.download(np.uint64, (N, 4))[n:])


@pytest.mark.parametrize("log_n,log_blowup", [(5, 1), (3, 9), (6, 11), (5, 12), (4, 14), (1, 16), (8, 10)])
def test_evaluate_few_coefficients_on_many_points(ctx, be, oracle, log_n, log_blowup):
    """ss_evaluate_fp252 with a large blow-up (DEEP's rational functions: 2^8 coefficients on 2^24 points): the first log_blowup stages
    of the network only replicate and are skipped - the first pass whole when they are its 11 stages or more, and stages of the
    next (a strided pass that reads the source at the shifted index).  Equal to the full transform of the zero-padded column, and
    to the polynomial at a few points."""
    n, N = 1 << log_n, 1 << (log_n + log_blowup)
    g = g3(oracle)
    coeffs = [random_column(n, 900 + c) for c in range(2)]
    brev = [int(format(i, "0%db" % log_n)[::-1], 2) for i in range(n)]
    d_co = _up(ctx, [c[brev] for c in coeffs])
    outs = [ctx.alloc(32 * N) for _ in coeffs]
    ctx.evaluate(d_co, log_n, log_blowup, g, outs)
    full = be.Matrix.from_host(ctx, [np.concatenate([c, np.zeros((N - n, 4), dtype=np.uint64)]) for c in coeffs])
    full.evaluate(g)
    w = pow(3, (P - 1) >> (log_n + log_blowup), P)
    for c in range(2):
        got = outs[c].download(np.uint64, (N, 4))
        assert np.array_equal(got, full.to_host()[c])
        for i in (0, 1, N // 2 + 3, N - 1):
            assert np.array_equal(got[i], oracle.poly_eval(coeffs[c], oracle.to_mont([3 * pow(w, i, P) % P])[0])), (c, i)


@pytest.mark.parametrize("log_n,shape", [(10, "two large"), (12, "two large"), (12, "one large"), (13, "all large"), (11, "many offsets")])
def test_deep_compose_of_a_layout_sized_mask(ctx, be, oracle, log_n, shape, monkeypatch):
    """A layout's mask has columns with dozens of cells (starknet: 105, 60, 56) and one constant per distinct offset (191): from 2^20
    points on ss_deep_compose takes those as RATIONAL functions - A_c(x) / B(x), B over the mask's distinct offsets, evaluated by one
    pruned transform each (deep.hip) - instead of a tap per cell.  The same values as the taps (SS_DEEP_TAPS=1) and the oracle's:
    columns of 40 / 26 / 3 / 1 cells, one or all of them above the 24-cell bar, offsets up to n - 1 and beyond (taken mod n), a cell
    named twice, more distinct offsets than a 2^7-entry coefficient array holds."""
    lb = 1
    n, N = 1 << log_n, 1 << (log_n + lb)
    g = g3(oracle)
    rng = np.random.default_rng(log_n * 11 + len(shape))
    ncols = 4
    cols = [random_column(n, c + 700) for c in range(ncols)]
    m = be.Matrix.from_host(ctx, cols)
    ev, co = m.lde(lb, g)
    comp_coeffs = [random_column(n, 790 + k) for k in range(2)]
    cm = be.Matrix.from_host(ctx, [np.concatenate([c, np.zeros((N - n, 4), dtype=np.uint64)]) for c in comp_coeffs])
    cm.evaluate(g)
    pick = lambda k, hi: sorted({int(v) for v in rng.integers(0, hi, size=3 * k)} | {0, 1})[:k]
    if shape == "two large":
        offs = [pick(40, n), pick(26, min(n, 600)), [0, 1, n - 1], [5]]
    elif shape == "one large":
        offs = [pick(30, n), pick(20, n), [0, 1, 2], [0]]
    elif shape == "all large":
        offs = [pick(40, n), pick(30, n), pick(25, n), pick(24, 64)]
    else:
        offs = [pick(90, n), pick(80, n), [0], [1]]
    mask = [(c, o) for c in range(ncols) for o in offs[c]] + [(0, offs[0][3]), (1, offs[1][2] + n)]
    mc, mo = [c for c, _ in mask], [o for _, o in mask]
    z = 0x2468ACE13579B ** 5 % P
    zm = oracle.to_mont([z])[0]
    ood_t = ctx.ood_eval(co.cols, log_n, mc, mo, zm)
    ood_c = np.stack([oracle.poly_eval(c, oracle.to_mont([z * z % P])[0]) for c in comp_coeffs])
    alpha = 192837465564738291
    ct = oracle.to_mont([pow(alpha, j, P) for j in range(len(mask))])
    cc = oracle.to_mont([pow(alpha, len(mask) + k, P) for k in range(2)])
    out, out_taps = ctx.alloc(32 * N), ctx.alloc(32 * N)
    monkeypatch.setenv("SS_DEEP_RATIONAL_MIN_LOG", "8")           # the library takes this path from 2^20 points on
    ctx.deep_compose(ev.cols, cm.cols, log_n, lb, g, mc, mo, ood_t, ct, ood_c, cc, zm, out)
    monkeypatch.setenv("SS_DEEP_TAPS", "1")
    ctx.deep_compose(ev.cols, cm.cols, log_n, lb, g, mc, mo, ood_t, ct, ood_c, cc, zm, out_taps)
    got = out.download(np.uint64, (N, 4))
    assert np.array_equal(got, out_taps.download(np.uint64, (N, 4)))
    want = oracle.deep_compose(ev.to_host(), cm.to_host(), log_n, lb, g, mc, mo, ood_t, ct, ood_c, cc, zm)
    assert np.array_equal(got, want)


def test_production_paths_at_2p20_without_overrides(ctx, be, oracle, monkeypatch):
    """the paths `bench.py` times are chosen by thresholds (point-by-point out-of-domain values and DEEP's rational
    functions from 2^20 points on) that the parity tests above only reach by overriding them at small sizes.  Here at 2^20 rows
    with NO override: ss_ood_eval and ss_deep_compose as a proof calls them, against the other path forced explicitly
    (SS_OOD_TRANSFORM=1 / SS_DEEP_TAPS=1: the paths the oracle holds at small sizes) - bit for bit - and sampled cells / points
    against the oracle's polynomial evaluation."""
    for v in ("SS_OOD_SPARSE_MIN_LOG", "SS_OOD_TRANSFORM", "SS_OOD_BLOCK_LOG", "SS_DEEP_RATIONAL_MIN_LOG", "SS_DEEP_RATIONAL_MIN_CELLS", "SS_DEEP_TAPS"):
        monkeypatch.delenv(v, raising=False)
    log_n, lb = 20, 1
    n, N = 1 << log_n, 1 << (log_n + lb)
    g = g3(oracle)
    rng = np.random.default_rng(2020)
    cols = [random_column(n, c + 900) for c in range(3)]
    m = be.Matrix.from_host(ctx, cols)
    ev, co = m.lde(lb, g)
    comp_coeffs = [random_column(n, 990 + k) for k in range(2)]
    cm = be.Matrix.from_host(ctx, [np.concatenate([c, np.zeros((N - n, 4), dtype=np.uint64)]) for c in comp_coeffs])
    cm.evaluate(g)
    pick = lambda k, hi: sorted({int(v) for v in rng.integers(0, hi, size=3 * k)} | {0, 1})[:k]
    offs = [pick(60, 40000), pick(14, n), [0, 1, 33158]]          # a column above both bars (12 / 24 cells), one between, a small one
    mask = [(c, o) for c in range(3) for o in offs[c]]
    mc, mo = [c for c, _ in mask], [o for _, o in mask]
    z = 0x13579BDF02468ACE ** 3 % P
    zm = oracle.to_mont([z])[0]
    ood_t = ctx.ood_eval(co.cols, log_n, mc, mo, zm)                               # point by point (the default from 2^20 on)
    monkeypatch.setenv("SS_OOD_TRANSFORM", "1")
    assert np.array_equal(ood_t, ctx.ood_eval(co.cols, log_n, mc, mo, zm))          # one coset transform per column
    monkeypatch.delenv("SS_OOD_TRANSFORM")
    w = pow(3, (P - 1) >> log_n, P)
    coeff_h = co.to_host()
    for j in (0, 17, 59, 60, 73, len(mask) - 1):
        c, o = mask[j]
        want = oracle.poly_eval(oracle.bitrev_permute(coeff_h[c]), oracle.to_mont([z * pow(w, o, P) % P])[0])
        assert np.array_equal(ood_t[j], want), j
    ood_c = np.stack([oracle.poly_eval(c, oracle.to_mont([z * z % P])[0]) for c in comp_coeffs])
    alpha = 918273645546372819
    ct = oracle.to_mont([pow(alpha, j, P) for j in range(len(mask))])
    cc = oracle.to_mont([pow(alpha, len(mask) + k, P) for k in range(2)])
    out, out_taps = ctx.alloc(32 * N), ctx.alloc(32 * N)
    ctx.deep_compose(ev.cols, cm.cols, log_n, lb, g, mc, mo, ood_t, ct, ood_c, cc, zm, out)          # rational functions (default from 2^20 on)
    monkeypatch.setenv("SS_DEEP_TAPS", "1")
    ctx.deep_compose(ev.cols, cm.cols, log_n, lb, g, mc, mo, ood_t, ct, ood_c, cc, zm, out_taps)     # a tap per cell
    got = out.download(np.uint64, (N, 4))
    assert np.array_equal(got, out_taps.download(np.uint64, (N, 4)))


@pytest.mark.parametrize("seed,size,log_n", [(1, 30, 3), (2, 120, 8), (3, 400, 12)])
def test_eval_quotient_vs_oracle(ctx, be, oracle, seed, size, log_n):
    import random
    from sandstorm_amd import air_program as ap
    from tests.test_air_program import random_dag
    lb, ncols = 1, 3
    n, N = 1 << log_n, 1 << (log_n + lb)
    g = g3(oracle)
    m = be.Matrix.from_host(ctx, [random_column(n, c + seed) for c in range(ncols)])
    ev, _ = m.lde(lb, g, keep_coeffs=False)
    tabs = [random_column(4, 50 + seed), random_column(8, 60 + seed)]
    tables, desc = np.concatenate(tabs), [0, 2, 4, 3]
    root = random_dag(random.Random(seed), ncols, 2, 5, size)
    prog = ap.lower(root, P)
    out = ctx.alloc(32 * N)
    ctx.eval_quotient(prog, ctx.column(tables), desc, ev.cols, log_n, lb, g, out)
    got = out.download(np.uint64, (N, 4))
    want = oracle.eval_program(prog.code, oracle.to_mont(prog.consts), tables, desc, prog.n_slots, ev.to_host(),
                               log_n, lb, g)
    assert np.array_equal(got, want)


# ----------------------------------------------------------------------------- memory pool
def test_dev_pool_reuse_and_trim(ctx):
    """ss_dev_alloc/ss_dev_free are pooled: a freed block is handed out again for a request of
    similar size, data written through a recycled block is intact, and trim empties the cache."""
    import gc
    gc.collect()
    ctx.trim()                                # start from an empty cache (other tests share this context)
    a = ctx.alloc(1 << 20)
    pa = a.ptr
    a.upload(np.arange(1 << 17, dtype=np.uint64))
    a.free()
    b = ctx.alloc((1 << 20) - 4096)           # within the 25 % slack: same block
    assert b.ptr == pa
    src = np.arange(7, 7 + (1 << 17) - 512, dtype=np.uint64)
    b.upload(src)
    assert np.array_equal(b.download(np.uint64, src.shape), src)
    c = ctx.alloc(1 << 10)                    # far smaller: must not take a 1 MiB block
    assert c.ptr != pa
    b.free(); c.free()
    ctx.trim()
    d = ctx.alloc(1 << 20)
    d.free()
    # freeing a pointer the pool does not own is an error, not a crash
    import ctypes as C
    from sandstorm_amd._lib import SandstormHipError, check
    with pytest.raises(SandstormHipError):
        check(ctx.lib.ss_dev_free(ctx.handle, C.c_void_p(0x1000)))
