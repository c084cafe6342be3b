"""The `recursive` layout restatement (sandstorm_amd/layouts/recursive.py): the AIR constraints must vanish on the
base trace generated from the reference's own example run (tests/golden/example/: `cairo-run` output of array-sum,
2^14 steps) — constraints (layouts/src/recursive/air.rs) and trace generation (trace.rs) are restated independently
and validate each other; a corrupted cell must trip the constraint that reads it."""
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "tests", "golden", "example")


from sandstorm_amd.examples import load_run, recursive_example  # noqa: E402,F401  (moved: bench.py proves these statements too)


def with_extension(rec, cols, challenges):
    """base columns + the ORACLE's extension columns (the product's device version is compared with the oracle in
    tests/test_gpu_extension.py) -> (all 10 columns as ints, final products)"""
    import numpy as np
    from oracle import oracle_py as oracle
    n = len(cols[0])
    aux = {"npc": oracle.to_mont(cols[rec.COL_NPC]), "memory": oracle.to_mont(cols[rec.COL_MEMORY]),
           "range_check": oracle.to_mont(cols[rec.COL_RANGE_CHECK]),
           "diluted_unordered": oracle.to_mont(cols[rec.COL_DILUTED_UNORDERED]),
           "diluted_ordered": oracle.to_mont(cols[rec.COL_DILUTED_ORDERED])}
    ext, lasts = oracle.build_extension_columns("recursive", aux, [oracle.to_mont([c])[0] for c in challenges], n)
    return cols + [[int(v) for v in oracle.from_mont(e)] for e in ext], [int(v) for v in oracle.from_mont(np.stack(lasts))]


CHALLENGES = [pow(7, 11 + 3 * i, 2**251 + 17 * 2**192 + 1) for i in range(6)]


@pytest.fixture(scope="module")
def example():
    from sandstorm_amd.layouts import recursive as rec
    states, memory, pi = load_run()
    cols, lasts = with_extension(rec, rec.base_trace(states, memory, pi), CHALLENGES)
    hints = rec.Hints.from_public_input(pi, CHALLENGES, len(cols[0]))
    # the reference's own asserts (trace.rs:734, 757) and the two hints that are functions of the challenges
    assert lasts == [hints.memory_quotient, 1, 1]
    assert cols[rec.COL_DILUTED_AGGREGATE][-1] == hints.diluted_check_cumulative_value
    return rec, cols, rec.constraints(hints, CHALLENGES)


def sample(domain_rows, k=3000):
    rows = list(domain_rows)
    if len(rows) <= k:
        return rows
    rng = random.Random(7)
    return rows[:500] + rows[-500:] + rng.sample(rows, k - 1000)


def test_constraints_vanish_on_the_example_trace(example):
    rec, cols, constraints = example
    n = len(cols[0])
    assert n == 16 * 16384 and len(cols) == rec.NUM_BASE_COLUMNS + rec.NUM_EXTENSION_COLUMNS
    assert len(constraints) == 93                      # the reference's count (air.rs:1083-1180) and len({c.name for c in constraints}) == len(constraints)
    for c in constraints:
        assert rec.failing_rows(c, cols, sample(c.domain.rows(n))) == [], c.name


def test_a_corrupted_cell_trips_its_constraints(example):
    rec, cols, constraints = example
    by_name = {c.name: c for c in constraints}
    cycle_row = 16 * 1234
    cases = [
        (rec.COL_AUXILIARY, rec.Auxiliary.RES, ["cpu/operands/res"]),
        (rec.COL_AUXILIARY, rec.Auxiliary.AP, ["cpu/operands/mem_dst_addr", "cpu/update_registers/update_ap/ap_update"]),
        (rec.COL_NPC, rec.Npc.PC, ["cpu/update_registers/update_pc/pc_cond_negative"]),
        (rec.COL_RANGE_CHECK, rec.RangeCheck.OFF_OP0, ["cpu/decode/opcode_rc_input", "cpu/operands/mem0_addr"]),
        (rec.COL_FLAGS, 3, ["cpu/decode/opcode_rc/bit"]),
        (rec.COL_MEMORY, 1, ["memory/multi_column_perm/perm/step0", "memory/is_func"]),
        (rec.COL_RANGE_CHECK, rec.RangeCheck.ORDERED, ["rc16/perm/step0", "rc16/diff_is_bit"]),
        (rec.COL_MEM_RC_PERMUTATION, 0, ["memory/multi_column_perm/perm/step0"]),
        (rec.COL_NPC, rec.Npc.UNUSED_ADDR, ["memory/multi_column_perm/perm/step0"]),
    ]
    for col, cell, names in cases:
        old = cols[col][cycle_row + cell]
        cols[col][cycle_row + cell] = (old + 5) % rec.P
        try:
            rows = list(range(cycle_row - 16, cycle_row + 32))
            tripped = [nm for nm in names if rec.failing_rows(by_name[nm], cols, [r for r in rows if r in set(by_name[nm].domain.rows(len(cols[0])))] or rows)]
            assert tripped, (col, cell, names)          # which of them fire depends on the instruction at that cycle
        finally:
            cols[col][cycle_row + cell] = old


def test_bitwise_and_range_check_builtins_with_real_instances():
    """the example run uses no builtin, so its instances are all-zero dummies: here random bitwise and 128-bit
    range-check instances go through the same trace generation, and the builtin + diluted-check + range-check
    constraints must still vanish (the builtins' memory cells live in their own segments, so the rest is unaffected)"""
    from sandstorm_amd.layouts import recursive as rec
    states, memory, pi = load_run()
    rng = random.Random(11)
    private = {"bitwise": [(i, rng.getrandbits(251), rng.getrandbits(251)) for i in range(40)],
               # 16-bit parts close to the instruction offsets: the ordered range-check cells must hold every value
               # between the extremes (a full-range value would need a longer trace, as in the reference)
               "range_check": [(i, sum(rng.randrange(32700, 32800) << (16 * k) for k in range(8))) for i in range(25)]}
    import copy
    pi2 = copy.deepcopy(pi)
    cols = rec.base_trace(states, memory, pi2, private)
    # rc_min / rc_max of the public input must cover the builtin's 16-bit parts now
    parts = [(v >> (16 * k)) & 0xFFFF for _, v in private["range_check"] for k in range(8)]
    pi2.rc_min, pi2.rc_max = min(pi2.rc_min, min(parts)), max(pi2.rc_max, max(parts))
    cols = rec.base_trace(states, memory, pi2, private)
    cols, lasts = with_extension(rec, cols, CHALLENGES)
    hints = rec.Hints.from_public_input(pi2, CHALLENGES, len(cols[0]))
    assert lasts[1:] == [1, 1] and cols[rec.COL_DILUTED_AGGREGATE][-1] == hints.diluted_check_cumulative_value
    n = len(cols[0])
    x, y = private["bitwise"][7][1:]
    assert cols[rec.COL_NPC][7 * 128 + rec.Npc.BITWISE_X_OR_Y_ADDR + 1] == x | y
    for c in rec.constraints(hints, CHALLENGES):
        if c.name.split("/")[0] in ("bitwise", "diluted_check", "rc16", "rc_builtin"):
            assert rec.failing_rows(c, cols, sample(c.domain.rows(n))) == [], c.name


def test_pedersen_builtin_with_real_instances():
    """random inputs, the all-flags value p - 1 = 2^251 + 2^196 + 2^192, single-bit and zero inputs"""
    from sandstorm_amd.layouts import recursive as rec
    states, memory, pi = load_run()
    rng = random.Random(5)
    top = (1 << 251) | (1 << 196) | (1 << 192)
    assert top == rec.P - 1
    private = {"pedersen": [(0, rng.getrandbits(250), rng.getrandbits(250)), (1, top, (1 << 251) | (1 << 196)), (2, 0, 5), (3, 1 << 251, 1)]}
    cols = rec.base_trace(states, memory, pi, private)
    n = len(cols[0])
    assert cols[rec.COL_AUXILIARY][2048 + 7] == 1 and cols[rec.COL_AUXILIARY][2048 + 1022] == 1      # instance 1, input a: all three bits
    assert cols[rec.COL_AUXILIARY][2048 + 1024 + 7] == 0 and cols[rec.COL_AUXILIARY][2048 + 1024 + 1022] == 1
    for c in rec.constraints(rec.Hints.from_public_input(pi)):
        if c.name.startswith("pedersen"):
            rows = list(c.domain.rows(n))
            rows = rows if len(rows) <= 4000 else rows[:3500] + rows[-500:]
            assert rec.failing_rows(c, cols, rows) == [], c.name


def test_mask_is_the_reference_mask():
    """the trace cells the restated constraints read = the 133-cell mask extracted from the reference's source
    (its size is the length of the out-of-domain vector in the reference's shipped recursive proof)"""
    from tests import survey_masks
    from sandstorm_amd.layouts import recursive as rec
    _, _, pi = load_run()
    cells, seen = set(), set()

    def walk(e):
        if e._id in seen:
            return
        seen.add(e._id)
        if e.kind == "trace":
            cells.add(tuple(e.args))
        elif e.kind in ("add", "sub", "mul", "inv"):
            for a in e.args:
                walk(a)
    for c in rec.constraints(rec.Hints.from_public_input(pi, CHALLENGES, 1 << 18), CHALLENGES):
        walk(c.numerator)
    assert cells == {(c, o) for c, offs in survey_masks.RECURSIVE_MASK.items() for o in offs} and len(cells) == 133


def test_domain_rows_are_the_zeros_of_their_zerofiers():
    """both sides are restated from the reference's zerofier expressions: at a trace point g^r the multiplier
    prod(num) / prod(den) has a pole exactly on the rows where the constraint is enforced"""
    from sandstorm_amd.layouts import recursive as rec
    _, _, pi = load_run()
    n = 4096
    g = pow(3, (rec.P - 1) // n, rec.P)
    doms = {}
    for c in rec.constraints(rec.Hints.from_public_input(pi, CHALLENGES, n), CHALLENGES):
        doms[c.domain.name] = c.domain
    assert len(doms) >= 20
    xs = [pow(g, r, rec.P) for r in range(n)]
    for name, d in doms.items():
        want = set(d.rows(n))
        got = set()
        for r, x in enumerate(xs):
            den = 1
            for p_, e in d.den(n):
                den = den * (pow(x, p_, rec.P) - pow(g, e, rec.P)) % rec.P
            num = 1
            for p_, e in d.num(n):
                num = num * (pow(x, p_, rec.P) - pow(g, e, rec.P)) % rec.P
            if den == 0 and num != 0:
                got.add(r)
            assert not (den == 0 and num == 0) or r not in want, name       # a cancelled zero is a row that is NOT enforced
        assert got == want, name


def test_composition_of_the_example_is_a_polynomial(example, oracle):
    """The whole chain on the CPU with the oracle: LDE of the 10 real columns, the composition constraint
    sum alpha^i C_i (air.rs:1183-1199) lowered to the constraint VM with its periodic / zerofier tables, evaluated on
    the LDE coset — it interpolates to a polynomial of degree exactly 2n - 3 (the first-row constraints), i.e. every
    constraint is divisible by its zerofier; the VM agrees with the big-integer definition at sampled points; one
    corrupted trace cell destroys the divisibility."""
    import numpy as np
    from sandstorm_amd import air_program as ap
    rec, cols, _ = example
    _, _, pi = load_run()
    n = len(cols[0])
    N, log_n = 2 * n, n.bit_length() - 1
    hints = rec.Hints.from_public_input(pi, CHALLENGES, n)
    alpha = pow(5, 77, rec.P)
    tables = rec.Tables(n)
    expr = rec.composition(n, hints, CHALLENGES, alpha, tables)
    prog = ap.lower(expr, rec.P)
    assert rec.mask() == sorted(rec.mask()) and len(rec.mask()) == 133
    vals, desc, off, by_spec = [], [], 0, {}
    for spec in tables.specs:
        v = tables.host_values(spec)
        by_spec[spec] = v
        desc += [off, len(v).bit_length() - 1]
        off += len(v)
        vals += v
    tab, g = oracle.to_mont(vals), oracle.to_mont([3])[0]

    def composition_coefficients(columns):
        lde = [oracle.lde(oracle.to_mont(c), 1, g)[0] for c in columns]
        out = oracle.eval_program(prog.code, oracle.to_mont(prog.consts), tab, desc, prog.n_slots, lde, log_n, 1, g)
        return lde, out, oracle.ntt(out, inverse=True, offset=g)

    lde, out, coeffs = composition_coefficients(cols)
    nonzero = np.nonzero(coeffs.any(axis=1))[0]
    assert int(nonzero[-1]) == 2 * n - 3
    w = pow(3, (rec.P - 1) // N, rec.P)
    for i in (0, 1, 54321, N - 1):
        x = 3 * pow(w, i, rec.P) % rec.P
        want = ap.evaluate(expr, rec.P, x, lambda c, o: int(oracle.from_mont(lde[c][(i + 2 * o) % N][None])[0]),
                           lambda t: tables.value_at(tables.specs[t], x))
        assert int(oracle.from_mont(out[i][None])[0]) == want
    # the verifier's side: out-of-domain identity sum alpha^i C_i(z) = H0(z^2) + z H1(z^2) with the OOD values computed
    # from the polynomials themselves (trace coefficients at z w^k, even / odd halves of the composition at z^2)
    va = rec.verifier_air(pi)
    assert va.mask == rec.mask()
    z = pow(11, 1234567, rec.P)
    wn = pow(3, (rec.P - 1) // n, rec.P)
    trace_coeffs = [oracle.ntt(oracle.to_mont(c), inverse=True) for c in cols]
    ood = {(c, o): int(oracle.from_mont(oracle.poly_eval(trace_coeffs[c], oracle.to_mont([z * pow(wn, o, rec.P) % rec.P])[0])[None])[0])
           for c, o in va.mask}
    lhs = ap.evaluate(va.composition(n, CHALLENGES, alpha), rec.P, z, lambda c, o: ood[(c, o)], lambda t: va.table_at(n, z, t))
    h0, h1 = np.ascontiguousarray(coeffs[0::2]), np.ascontiguousarray(coeffs[1::2])
    z2 = oracle.to_mont([z * z % rec.P])[0]
    rhs = (int(oracle.from_mont(oracle.poly_eval(h0, z2)[None])[0]) + z * int(oracle.from_mont(oracle.poly_eval(h1, z2)[None])[0])) % rec.P
    assert lhs == rhs
    # the C++ host's mirror of the AIR (host/air_recursive.cpp): its program, with its own table numbering, evaluates to
    # the same composition on every point of the LDE coset
    from sandstorm_amd import hostlib
    cpp = hostlib.RecursiveHostAir(None, pi, log_n)
    assert cpp.mask_size == 133 and (cpp.num_base_columns, cpp.num_extension_columns) == (7, 3)
    code, consts, n_slots, specs = cpp.dump(n, [oracle.to_mont([c])[0] for c in CHALLENGES], oracle.to_mont([alpha])[0])
    assert sorted(map(repr, specs)) == sorted(map(repr, tables.specs))
    cvals, cdesc, coff = [], [], 0
    for spec in specs:
        v = by_spec[spec]
        cdesc += [coff, len(v).bit_length() - 1]
        coff += len(v)
        cvals += v
    out_cpp = oracle.eval_program(code, consts, oracle.to_mont(cvals), cdesc, n_slots, lde, log_n, 1, g)
    assert np.array_equal(out_cpp, out)
    # Air::prepare_program (what the prover calls while the device extends the extension trace): the program lowered ahead of the
    # composition coefficient, its powers patched in - word for word the program built in one go, for this coefficient and another;
    # a program prepared for OTHER challenges is not used
    mch = [oracle.to_mont([c])[0] for c in CHALLENGES]
    for a2 in (alpha, alpha * 7 + 3):
        want2 = cpp.dump(n, mch, oracle.to_mont([a2])[0])
        cpp.prepare(n, mch)
        got2 = cpp.dump(n, mch, oracle.to_mont([a2])[0])
        assert np.array_equal(got2[0], want2[0]) and np.array_equal(got2[1], want2[1]) and got2[2:] == want2[2:]
        cpp.prepare(n, [oracle.to_mont([c + 1])[0] for c in CHALLENGES])
        got3 = cpp.dump(n, mch, oracle.to_mont([a2])[0])
        assert np.array_equal(got3[0], want2[0]) and np.array_equal(got3[1], want2[1])
    cpp.close()
    bad = [list(c) for c in cols[:1]] + cols[1:]
    bad[0][16 * 777 + 3] = (bad[0][16 * 777 + 3] + 1) % rec.P          # one flag cell of one cycle
    _, _, coeffs_bad = composition_coefficients(bad)
    assert coeffs_bad[-1].any() and coeffs_bad[-2].any()


def test_cpp_base_trace_equals_the_python_one(oracle):
    """sandstorm_amd/host/trace_recursive.cpp against layouts/recursive.py::base_trace, cell for cell: the example run,
    and the same run with real Pedersen / bitwise / range-check instances"""
    import numpy as np
    from sandstorm_amd import hostlib
    from sandstorm_amd.layouts import recursive as rec
    states, memory, pi = load_run()
    with open(os.path.join(EX, "trace.bin"), "rb") as f:
        trace_bin = f.read()
    with open(os.path.join(EX, "memory.bin"), "rb") as f:
        memory_bin = f.read()
    rng = random.Random(21)
    top = (1 << 251) | (1 << 196) | (1 << 192)
    private = {"pedersen": [(0, rng.getrandbits(250), rng.getrandbits(250)), (1, top, (1 << 251) | (1 << 196)), (5, 0, 5)],
               "bitwise": [(i, rng.getrandbits(251), rng.getrandbits(251)) for i in range(9)],
               "range_check": [(i, sum(rng.randrange(32700, 32800) << (16 * k) for k in range(8))) for i in range(6)]}
    for priv in (None, private):
        want = rec.base_trace(states, memory, pi, priv)
        got = hostlib.recursive_base_trace(trace_bin, memory_bin, pi, priv)
        assert len(got) == len(want) == 7
        for c, (g, w) in enumerate(zip(got, want)):
            assert np.array_equal(g, oracle.to_mont(w)), "column %d" % c
        # the OpenMP path (normally only above 2^16 cycles) must give the same cells
        os.environ["SSH_TRACE_PARALLEL_MIN"] = "1"
        try:
            par = hostlib.recursive_base_trace(trace_bin, memory_bin, pi, priv)
        finally:
            del os.environ["SSH_TRACE_PARALLEL_MIN"]
        for c, (g, w) in enumerate(zip(got, par)):
            assert np.array_equal(g, w), "parallel path, column %d" % c
    from sandstorm_amd._lib import SandstormHipError
    with pytest.raises(SandstormHipError, match="power of two"):
        hostlib.recursive_base_trace(trace_bin[:24 * 1000], memory_bin, pi)


def test_cpp_trace_refuses_a_far_away_address_without_sizing_anything_by_it():
    """one corrupt high address (here a public-memory entry at 2^32 - 16: addresses cross the C boundary as 32-bit) must end in the generator's own message - more gaps
    than there are cycles to hold them - and not in count / first-access / seen arrays of 20 B per address up to it (86 GB).  The
    address space limit of the child process is what makes a regression visible here."""
    import dataclasses
    import subprocess
    import sys
    code = """
import os, resource, sys
sys.path.insert(0, %r)
from sandstorm_amd import hostlib
from sandstorm_amd._lib import SandstormHipError
from sandstorm_amd.examples import load_run
import dataclasses
states, memory, pi = load_run()
ex = %r
trace_bin, memory_bin = open(os.path.join(ex, "trace.bin"), "rb").read(), open(os.path.join(ex, "memory.bin"), "rb").read()
hostlib.load()
resource.setrlimit(resource.RLIMIT_AS, (24 << 30, 24 << 30))
for addr in (0xFFFFFFF0, 1 << 24):
    bad = dataclasses.replace(pi, public_memory=pi.public_memory + [(addr, 7)])
    try:
        hostlib.recursive_base_trace(trace_bin, memory_bin, bad)
    except SandstormHipError as e:
        assert "more memory gaps than cycles" in str(e), e
        print("REFUSED", addr)
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), EX)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.count("REFUSED") == 2, out.stdout[-1000:] + out.stderr[-3000:]


@pytest.mark.parametrize("layout", ["recursive", "starknet"])
def test_column_done_is_called_when_a_column_is_final(layout):
    """the generators' column_done callbacks (host/trace_*.hpp; ssh_prove_files uploads a column the moment its callback comes): at
    callback time the column already holds what it holds at the end, every column is announced once, in the announced order"""
    import numpy as np
    from sandstorm_amd import binary, examples, hostlib
    if layout == "recursive":
        states, memory, pi = examples.recursive_example(14)
        plain = hostlib.recursive_base_trace
    else:
        states, memory, pi = examples.starknet_example(17)             # (the reference's own run: its diluted values need 2^17 steps)
        plain = hostlib.starknet_base_trace
    trace_bin, memory_bin = binary.write_register_states(states), binary.write_memory(memory)
    ncols, n = (7, 16 << 14) if layout == "recursive" else (9, 16 << 17)
    out = [np.full((n, 4), 0xA5A5A5A5A5A5A5A5, dtype=np.uint64) for _ in range(ncols)]          # (not zero: a cell nobody wrote would show)
    seen, snap = [], {}

    def done(c):
        seen.append(c)
        snap[c] = out[c].copy()
    cols, order = hostlib.base_trace_with_callback(layout, trace_bin, memory_bin, pi, None, out, done)
    assert seen == order and sorted(seen) == list(range(ncols))
    want = plain(trace_bin, memory_bin, pi)
    for c in range(ncols):
        assert np.array_equal(cols[c], want[c]), "column %d" % c
        assert np.array_equal(snap[c], want[c]), "column %d was announced before it was final" % c


_PEDERSEN_DIGEST_SCRIPT = r"""
import hashlib, os, random, sys
sys.path.insert(0, %(root)r)
from sandstorm_amd import hostlib
ex = os.path.join(%(root)r, "tests", "golden", "example")
trace_bin, memory_bin = open(os.path.join(ex, "trace.bin"), "rb").read(), open(os.path.join(ex, "memory.bin"), "rb").read()
sys.path.insert(0, os.path.join(%(root)r, "tests"))
from test_layout_recursive import load_run, many_pedersen_instances
_, _, pi = load_run()
cols = hostlib.recursive_base_trace(trace_bin, memory_bin, pi, {"pedersen": many_pedersen_instances()})
print(" ".join(hashlib.sha256(c.tobytes()).hexdigest() for c in cols))
"""


def many_pedersen_instances():
    """inputs of the Pedersen builtin that meet every branch of its curve steps: no bit set, one bit, every low bit, the top bits the
    AIR's flags read (251, 196, 192), the largest field element, and random ones - as many as the example run has room for"""
    p = (1 << 251) + 17 * (1 << 192) + 1
    rng = random.Random(1906)
    edge = [0, 1, 2, (1 << 248) - 1, 1 << 247, 1 << 248, (1 << 251) | (1 << 196) | (1 << 192), (1 << 251) | (1 << 196), 1 << 251, p - 1, p - 2, (1 << 251) - 1]
    pairs = [(a, b) for a in edge[:6] for b in (0, edge[6])] + [(edge[i], edge[-1 - i]) for i in range(len(edge))]
    pairs += [(rng.randrange(p), rng.randrange(p)) for _ in range(20)]
    pairs += [pairs[3], pairs[-1]]                                   # (instances that repeat: traced once)
    return [(i, a, b) for i, (a, b) in enumerate(pairs)]


def test_pedersen_steps_from_one_inversion_equal_the_affine_ones(oracle):
    """host/trace_common.hpp element_steps: the partial sums of an instance in Jacobian coordinates, ONE batched inversion per input for
    every affine sum and every slope, the distinct instances traced by all threads - against the reference's shape (affine, an inversion
    per set bit: SSH_TRACE_AFFINE_STEPS=1 in a process of its own; the switch is read once) on inputs that meet every branch, column
    digest for column digest; and both against layouts/recursive.py on a few of them"""
    import hashlib
    import subprocess
    import numpy as np
    from sandstorm_amd import hostlib
    from sandstorm_amd.layouts import recursive as rec
    inst = many_pedersen_instances()
    assert len(inst) <= 128                                          # the example run's 2^13... steps hold 2^18 / 2048 instances
    script = _PEDERSEN_DIGEST_SCRIPT % {"root": ROOT}
    digests = {}
    for mode in ("jacobian", "affine"):
        env = dict(os.environ)
        env.pop("SSH_TRACE_AFFINE_STEPS", None)
        if mode == "affine":
            env["SSH_TRACE_AFFINE_STEPS"] = "1"
        out = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        digests[mode] = out.stdout.split()
    assert len(digests["jacobian"]) == 7 and digests["jacobian"] == digests["affine"]
    # the Python mirror on a handful (pure-Python curve arithmetic: 0.2 s per instance)
    states, memory, pi = load_run()
    with open(os.path.join(EX, "trace.bin"), "rb") as f:
        trace_bin = f.read()
    with open(os.path.join(EX, "memory.bin"), "rb") as f:
        memory_bin = f.read()
    few = {"pedersen": [(k, a, b) for k, (_, a, b) in enumerate(inst[9:14] + inst[-4:])]}
    want = rec.base_trace(states, memory, pi, few)
    got = hostlib.recursive_base_trace(trace_bin, memory_bin, pi, few)
    for c, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g, oracle.to_mont(w)), "column %d" % c
