"""The row-block entry points of the single-proof multi-GPU path (host/sharded.cpp drives them): ss_eval_quotient_rows,
ss_deep_compose_rows + ss_deep_extend give, block by block, what the whole-domain entry points give - bit for bit, on the real
recursive program, wrap-around halo included."""
import os

import numpy as np
import pytest

from tests.test_gpu_real_quotient import _Prog, _rand
from tests.test_layout_recursive import load_run
from tests.test_layout_starknet import CHALLENGES, P

pytestmark = pytest.mark.gpu


def test_row_block_forms_are_the_whole_domain_forms(oracle):
    from sandstorm_amd import backend as be, hostlib
    from sandstorm_amd.layouts import recursive as lay
    _, _, pi = load_run()
    log_n, R = 15, 4
    n, N = 1 << log_n, 2 << log_n
    B = N // R
    cpp = hostlib.RecursiveHostAir(None, pi, log_n)
    code, consts, n_slots, specs = cpp.dump(n, [oracle.to_mont([c])[0] for c in CHALLENGES], oracle.to_mont([pow(5, 77, P)])[0])
    cpp.close()
    tables = lay.Tables(n)
    rng = np.random.default_rng(3)
    tabs, desc, off = [], [], 0
    for spec in specs:
        t = _rand(rng, tables.length(spec))
        desc += [off, len(t).bit_length() - 1]
        off += len(t)
        tabs.append(t)
    lde = [_rand(rng, N) for _ in range(10)]
    g = oracle.to_mont([3])[0]
    ctx = be.Context(0)
    m = be.Matrix.from_host(ctx, lde)
    d_tab = ctx.column(np.concatenate(tabs))
    prog = _Prog(code, [int(v) for v in oracle.from_mont(consts)], n_slots)
    halo = 2058 * 2                                              # the recursive layout's largest row offset, LDE rows
    for interpret in (False, True):                             # the compiled kernel and the interpreter
        if interpret:
            os.environ["SS_QUOTIENT_INTERPRET"] = "1"
        try:
            whole = ctx.alloc(32 * N)
            ctx.eval_quotient(prog, d_tab, desc, m.cols, log_n, 1, g, whole)
            want = whole.download(np.uint64, (N, 4))
            for r in range(R):
                idx = (r * B + np.arange(B + halo)) % N        # the block and the rows behind it, wrapping around the domain
                blocks = [ctx.column(c[idx]) for c in lde]
                out = ctx.alloc(32 * B)
                ctx.eval_quotient_rows(prog, d_tab, desc, blocks, log_n, 1, g, r * B, B, B + halo, out)
                assert np.array_equal(out.download(np.uint64, (B, 4)), want[r * B:(r + 1) * B]), (interpret, r)
        finally:
            os.environ.pop("SS_QUOTIENT_INTERPRET", None)
    # a block without the rows its constraints reach is refused, not read out of bounds
    from sandstorm_amd._lib import SandstormHipError
    with pytest.raises(SandstormHipError, match="reaches beyond the block"):
        ctx.eval_quotient_rows(prog, d_tab, desc, blocks, log_n, 1, g, 0, B, B + 16, out)
    # ---- DEEP: sub-coset blocks + extension == the whole composition
    mask = lay.mask()
    mc, mo = [c for c, _ in mask], [o for _, o in mask]
    comp = [_rand(rng, N) for _ in range(2)]
    cm = be.Matrix.from_host(ctx, comp)
    ood_t, ood_c = _rand(rng, len(mask)), _rand(rng, 2)
    ct, cc = _rand(rng, len(mask)), _rand(rng, 2)
    z = oracle.to_mont([pow(11, 1234567, P)])[0]
    whole = ctx.alloc(32 * N)
    ctx.deep_compose(m.cols, cm.cols, log_n, 1, g, mc, mo, ood_t, ct, ood_c, cc, z, whole)
    want = whole.download(np.uint64, (N, 4))
    cnt = n // R
    # a tap per cell (what this size takes by itself), then the large columns as rational functions (the library's choice from 2^20
    # points on: every rank evaluates the polynomials on the whole sub-coset and uses its range of them)
    for rational in (False, True):
        if rational:
            os.environ["SS_DEEP_RATIONAL_MIN_LOG"] = "8"
        try:
            if rational:
                ctx.deep_compose(m.cols, cm.cols, log_n, 1, g, mc, mo, ood_t, ct, ood_c, cc, z, whole)
                assert np.array_equal(whole.download(np.uint64, (N, 4)), want)
            sub = ctx.alloc(32 * n)
            for r in range(R):
                tb = [ctx.column(c[r * B:(r + 1) * B]) for c in lde]
                cb = [ctx.column(c[r * B:(r + 1) * B]) for c in comp]
                ctx.deep_compose_rows(tb, cb, log_n, 1, g, mc, mo, ood_t, ct, ood_c, cc, z, r * cnt, cnt, be.DeviceView(sub, 32 * r * cnt, 32 * cnt))
            out = ctx.alloc(32 * N)
            ctx.deep_extend(sub, log_n, 1, g, out)
            assert np.array_equal(out.download(np.uint64, (N, 4)), want), rational
        finally:
            os.environ.pop("SS_DEEP_RATIONAL_MIN_LOG", None)
    ctx.close()
