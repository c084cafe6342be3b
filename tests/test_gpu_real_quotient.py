"""Q1 on the REAL programs: the lowered composition constraints of the `starknet` (195 constraints,
269 mask cells, row offsets up to 33 158: layouts/src/starknet/air.rs:115-2406) and `recursive` (93 constraints, 133
cells: layouts/src/recursive/air.rs:61-1200) layouts, exactly as the C++ host hands them to ss_eval_quotient, run on
the device over a whole 2^17-point evaluation domain of random columns and random tables and compared, bit for bit,
with the oracle's constraint VM.  At 2^16 trace rows the largest offsets wrap around the domain several times less
than at 2^15, but still wrap (33 158 * 2 > 2^16 LDE rows only once the row is within the last half)."""
import os

import numpy as np
import pytest

from tests.test_layout_recursive import load_run
from tests.test_layout_starknet import CHALLENGES, P, starknet_example

pytestmark = pytest.mark.gpu


class _Prog:
    def __init__(self, code, consts, n_slots):
        self.code, self.consts, self.n_slots = code, consts, n_slots


def _rand(rng, count):                          # any limbs with the top one below 2^59 are felts below p
    v = rng.integers(0, 1 << 63, size=(count, 4), dtype=np.uint64)
    v[:, 3] &= np.uint64((1 << 59) - 1)
    return np.ascontiguousarray(v)


@pytest.mark.parametrize("layout,log_n", [("starknet", 16), ("recursive", 16), ("starknet", 15)])
def test_real_composition_program_vs_oracle(oracle, layout, log_n):
    from sandstorm_amd import backend as be, hostlib
    if layout == "starknet":
        from sandstorm_amd.layouts import starknet as lay
        _, _, pi = starknet_example(11)
        cpp, ncols = hostlib.StarknetHostAir(None, pi, log_n), 10
    else:
        from sandstorm_amd.layouts import recursive as lay
        _, _, pi = load_run()
        cpp, ncols = hostlib.RecursiveHostAir(None, pi, log_n), 10
    n, N = 1 << log_n, 2 << log_n
    alpha = pow(5, 77, P)
    code, consts, n_slots, specs = cpp.dump(n, [oracle.to_mont([c])[0] for c in CHALLENGES], oracle.to_mont([alpha])[0])
    cpp.close()
    assert len(code) // 2 > (1000 if layout == "starknet" else 500)
    tables = lay.Tables(n)
    rng = np.random.default_rng(17 + log_n)
    tabs, desc, off = [], [], 0
    for spec in specs:
        t = _rand(rng, tables.length(spec))
        desc += [off, len(t).bit_length() - 1]
        off += len(t)
        tabs.append(t)
    tab = np.concatenate(tabs)
    lde = [_rand(rng, N) for _ in range(ncols)]
    g = oracle.to_mont([3])[0]
    want = oracle.eval_program(code, consts, tab, desc, n_slots, lde, log_n, 1, g)
    assert want.any()
    ctx = be.Context(0)
    m = be.Matrix.from_host(ctx, lde)
    out = ctx.alloc(32 * N)
    prog = _Prog(code, [int(v) for v in oracle.from_mont(consts)], n_slots)
    ctx.eval_quotient(prog, ctx.column(tab), desc, m.cols, log_n, 1, g, out)
    got = out.download(np.uint64, (N, 4))
    assert np.array_equal(got, want)
    ctx.close()


@pytest.mark.parametrize("layout", ["starknet", "recursive"])
def test_compiled_kernel_is_the_interpreter(oracle, layout, monkeypatch):
    """the generated straight-line kernel (csrc/quotient_gen_<layout>.hip, picked by the program's code hash) and the
    interpreter (csrc/quotient.hip, forced with SS_QUOTIENT_INTERPRET) give the same composition at every point of a
    2^19-point domain; the program at this size IS the one the kernel was generated from (its hash is checked here
    against the generator's), so the first run really is the compiled path"""
    import sys
    from sandstorm_amd import backend as be, hostlib
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gen_quotient
    log_n = 18
    if layout == "starknet":
        from sandstorm_amd.layouts import starknet as lay
        _, _, pi = starknet_example(11)
        cpp = hostlib.StarknetHostAir(None, pi, log_n)
    else:
        from sandstorm_amd.layouts import recursive as lay
        _, _, pi = load_run()
        cpp = hostlib.RecursiveHostAir(None, pi, log_n)
    n, N = 1 << log_n, 2 << log_n
    code, consts, n_slots, specs = cpp.dump(n, [oracle.to_mont([c])[0] for c in CHALLENGES], oracle.to_mont([pow(3, 99, P)])[0])
    cpp.close()
    with open(os.path.join(os.path.dirname(gen_quotient.ROOT + "/x"), "sandstorm_amd", "csrc", "quotient_gen_%s.hip" % layout)) as f:
        assert ("0x%016x" % gen_quotient.code_hash(code)) in f.read(), "the committed kernel was generated from another program"
    tables = lay.Tables(n)
    rng = np.random.default_rng(5)
    tabs, desc, off = [], [], 0
    for spec in specs:
        t = _rand(rng, tables.length(spec))
        desc += [off, len(t).bit_length() - 1]
        off += len(t)
        tabs.append(t)
    tab = np.concatenate(tabs)
    lde = [_rand(rng, N) for _ in range(10)]
    g = oracle.to_mont([3])[0]
    ctx = be.Context(0)
    m = be.Matrix.from_host(ctx, lde)
    d_tab = ctx.column(tab)
    prog = _Prog(code, [int(v) for v in oracle.from_mont(consts)], n_slots)
    out = ctx.alloc(32 * N)
    ctx.eval_quotient(prog, d_tab, desc, m.cols, log_n, 1, g, out)
    compiled = out.download(np.uint64, (N, 4))
    monkeypatch.setenv("SS_QUOTIENT_INTERPRET", "1")
    ctx.zero(out)
    ctx.eval_quotient(prog, d_tab, desc, m.cols, log_n, 1, g, out)
    interpreted = out.download(np.uint64, (N, 4))
    assert compiled.any() and np.array_equal(compiled, interpreted)
    ctx.close()


def test_back_to_back_evaluations_are_ordered_on_the_stream(oracle, monkeypatch):
    """ss_eval_quotient with a compiled program queues its kernels (one per part) and returns without waiting for the stream
    (no host round trip inside the hot path): three evaluations of DIFFERENT constant tables issued back to back
    into three outputs - the per-launch constants travel through one pinned staging buffer and one device scratch - each
    equal to the interpreter's result for its own constants, at 2^20 points (above the size round 2's timing guard kicked in:
    there is no guard any more, the compiled kernels are what runs)"""
    from sandstorm_amd import backend as be, hostlib
    from sandstorm_amd.layouts import recursive as lay
    log_n = 19
    _, _, pi = load_run()
    cpp = hostlib.RecursiveHostAir(None, pi, log_n)
    n, N = 1 << log_n, 2 << log_n
    dumps = [cpp.dump(n, [oracle.to_mont([c])[0] for c in CHALLENGES], oracle.to_mont([pow(3, 99 + k, P)])[0]) for k in range(3)]
    cpp.close()
    code, _, n_slots, specs = dumps[0]
    assert all(np.array_equal(d[0], code) for d in dumps) and not np.array_equal(dumps[0][1], dumps[1][1])   # same code, other constants
    tables = lay.Tables(n)
    rng = np.random.default_rng(8)
    tabs, desc, off = [], [], 0
    for spec in specs:
        t = _rand(rng, tables.length(spec))
        desc += [off, len(t).bit_length() - 1]
        off += len(t)
        tabs.append(t)
    ctx = be.Context(0)
    m = be.Matrix.from_host(ctx, [_rand(rng, N) for _ in range(10)])
    d_tab = ctx.column(np.concatenate(tabs))
    progs = [_Prog(code, [int(v) for v in oracle.from_mont(d[1])], n_slots) for d in dumps]
    g = oracle.to_mont([3])[0]
    outs = [ctx.alloc(32 * N) for _ in progs]
    for prog, out in zip(progs, outs):                       # queued back to back, no synchronisation in between
        ctx.eval_quotient(prog, d_tab, desc, m.cols, log_n, 1, g, out)
    got = [out.download(np.uint64, (N, 4)) for out in outs]
    monkeypatch.setenv("SS_QUOTIENT_INTERPRET", "1")
    for prog, have in zip(progs, got):
        out = ctx.alloc(32 * N)
        ctx.eval_quotient(prog, d_tab, desc, m.cols, log_n, 1, g, out)
        assert have.any() and np.array_equal(have, out.download(np.uint64, (N, 4)))
    assert not np.array_equal(got[0], got[1])
    ctx.close()
