"""The `starknet` layout restatement (sandstorm_amd/layouts/starknet.py) on the CPU.

What pins it to the reference (layouts/src/starknet/{air,trace}.rs, builtins/src/*):
  * the constraint count (195), the 269-cell mask with the per-column counts and largest offsets read off the
    reference's source - 269 is the length of the out-of-domain vector of its shipped starknet proofs;
  * the nine periodic-column polynomials, DERIVED here (curve-point doublings, Hades round constants) and equal,
    coefficient for coefficient, to the reference's tables (tests/golden/starknet_periodic_fingerprints.json);
  * the Poseidon margin keys derived here equal to the literals of air.rs:2052-2160, and StarkWare's zero-input
    example of the permutation (builtins/src/poseidon/mod.rs tests);
  * the trace generation and the constraints validate each other on the reference's own starknet-layout run
    (example/bootloader: 2^17 steps, its public and private input, two real Pedersen instances; committed compressed
    under tests/golden/bootloader): every constraint vanishes on its domain, the memory is continuous over the real
    builtin segments, the three permutation products close and the diluted aggregate ends at its closed form.  Real
    range-check, ECDSA, bitwise, EC-op and Poseidon instances are added on top (the run itself uses none).
The starknet layout needs 2^17 steps before its diluted check fits (60 free cells per 1024 rows for the 65535 padding
values): the array-sum example of the other tests only serves the size checks and the C++ comparison here."""
import gzip
import hashlib
import json
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_layout_recursive import load_run, sample  # noqa: E402

P = 2**251 + 17 * 2**192 + 1
CHALLENGES = [pow(7, 11 + 3 * i, P) for i in range(6)]
LOG_STEPS = 17


from sandstorm_amd.examples import starknet_example  # noqa: E402,F401  (moved: bench.py proves this statement too)


def bootloader_run():
    """example/bootloader of the reference: register states, memory, public input, private input (its Pedersen instances)"""
    from sandstorm_amd import binary, public_input
    g = os.path.join(ROOT, "tests", "golden")
    with gzip.open(os.path.join(g, "bootloader", "trace.bin.gz")) as f:
        states = binary.read_register_states(f.read())
    with gzip.open(os.path.join(g, "bootloader", "memory.bin.gz")) as f:
        memory = binary.read_memory(f.read())
    pi = public_input.AirPublicInput.from_json(os.path.join(g, "air_public_input_bootloader.json"))
    priv = binary.AirPrivateInput.from_json(os.path.join(g, "bootloader", "air-private-input.json")).instances
    assert all(priv[k] == [] for k in ("range_check", "ecdsa", "bitwise", "ec_op", "poseidon")) and len(priv["pedersen"]) == 2
    return states, memory, pi, {"pedersen": priv["pedersen"]}


def real_instances():
    from sandstorm_amd.layouts import starknet as sk
    rng = random.Random(2024)
    priv, msg = 0x1234567, rng.getrandbits(250)
    k = 1000
    while True:                                              # an honest signature (w = s^-1 as the builtin takes it)
        k += 1
        r = sk._ec_mul(k, sk.GENERATOR)[0]
        w = k * pow(msg + r * priv, -1, sk.CURVE_ORDER) % sk.CURVE_ORDER
        if 0 < r < 1 << 251 and 0 < w < 1 << 251:
            break
    pub = sk._ec_mul(priv, sk.GENERATOR)
    # a second signer over a message of one bit (a multiply-add whose chain has a single addition)
    priv2, msg2, k2 = rng.getrandbits(240) + 2, 1 << 7, 2000
    while True:
        k2 += 1
        r2 = sk._ec_mul(k2, sk.GENERATOR)[0]
        w2 = k2 * pow(msg2 + r2 * priv2, -1, sk.CURVE_ORDER) % sk.CURVE_ORDER
        if 0 < r2 < 1 << 251 and 0 < w2 < 1 << 251:
            break
    pub2 = sk._ec_mul(priv2, sk.GENERATOR)
    p5, q7, q9 = sk._ec_mul(5, sk.GENERATOR), sk._ec_mul(7, sk.GENERATOR), sk._ec_mul(9, sk.GENERATOR)
    top = (1 << 251) | (1 << 196) | (1 << 192)
    return {
        "pedersen": [(2, rng.getrandbits(250), rng.getrandbits(250)), (3, top, (1 << 251) | (1 << 196)), (7, 0, 5)],
        "range_check": [(i, sum(rng.randrange(32758, 32794) << (16 * j) for j in range(8))) for i in range(5)],
        "ecdsa": [(1, pub[0], msg, r, w), (4, pub2[0], msg2, r2, w2)],
        "bitwise": [(i, rng.getrandbits(251), rng.getrandbits(251)) for i in range(6)],
        "ec_op": [(0, p5[0], p5[1], q7[0], q7[1], rng.getrandbits(250)), (2, p5[0], p5[1], q9[0], q9[1], top),
                  (3, q7[0], q7[1], p5[0], p5[1], (1 << 251) | (1 << 196) | rng.getrandbits(190)),
                  # scalars of one and two bits, and of every low bit (the chains of host/trace_starknet.cpp mad_chain: sums that never,
                  # once, always change)
                  (6, p5[0], p5[1], q9[0], q9[1], 1), (7, q9[0], q9[1], q7[0], q7[1], 3), (9, q7[0], q7[1], q9[0], q9[1], (1 << 250) - 1)],
        "poseidon": [(0, 1, 2, 3), (5, rng.getrandbits(251), rng.getrandbits(251), rng.getrandbits(251))],
    }


_TRACES = {}


def bootloader_trace_with_real_instances():
    """-> (public input, private input, the 9 base columns): the Python generator on the bootloader run with real instances of every
    builtin added - two million rows take it half a minute, so the tests that need it share one (none of them writes to it)"""
    if "bootloader" not in _TRACES:
        from sandstorm_amd.layouts import starknet as sk
        states, memory, spi, private = bootloader_run()
        assert spi.layout == "starknet" and len(states) == spi.n_steps == 1 << LOG_STEPS and len(private["pedersen"]) == 2
        extra = real_instances()
        private["pedersen"] += extra.pop("pedersen")
        private.update(extra)
        _TRACES["bootloader"] = (spi, private, sk.base_trace(states, memory, spi, private))
    return _TRACES["bootloader"]


def array_sum_trace():
    """-> (public input, the 9 base columns) of starknet_example(17), shared likewise"""
    if "array_sum" not in _TRACES:
        from sandstorm_amd.layouts import starknet as sk
        states, memory, spi = starknet_example(17)
        _TRACES["array_sum"] = (spi, sk.base_trace(states, memory, spi))
    return _TRACES["array_sum"]


@pytest.fixture(scope="module")
def example():
    import numpy as np
    from oracle import oracle_py as oracle
    from sandstorm_amd.layouts import starknet as sk
    spi, private, base = bootloader_trace_with_real_instances()
    cols = list(base)
    n = len(cols[0])
    aux = {"npc": oracle.to_mont(cols[sk.COL_NPC]), "memory": oracle.to_mont(cols[sk.COL_MEMORY]), "range_check": oracle.to_mont(cols[sk.COL_RANGE_CHECK])}
    ext, lasts = oracle.build_extension_columns("starknet", aux, [oracle.to_mont([c])[0] for c in CHALLENGES], n)
    cols.append([int(v) for v in oracle.from_mont(ext[0])])
    hints = sk.Hints.from_public_input(spi, CHALLENGES, n)
    assert [int(v) for v in oracle.from_mont(np.stack(lasts))] == [hints.memory_quotient, 1, 1]      # trace.rs:1017, 1035, 1055
    assert cols[sk.COL_PERMUTATION][n - 8 + sk.DilutedCheck.AGGREGATE] == hints.diluted_check_cumulative_value
    return sk, cols, sk.constraints(hints, CHALLENGES)


def test_constraint_set_and_mask_have_the_reference_shape(golden):
    from tests import survey_masks
    from sandstorm_amd.layouts import starknet as sk
    cs = sk.constraints(sk.Hints(0, 0, 0, 0), CHALLENGES)
    assert len(cs) == 195 and len({c.name for c in cs}) == 195
    assert [c.name for c in cs[:2]] == ["cpu/decode/opcode_rc/bit", "cpu/decode/opcode_rc/zero"]
    assert cs[-1].name == "poseidon/poseidon/margin_partial_to_full2" and cs[33].name == "memory/multi_column_perm/perm/init0"
    mask = sk.mask()
    assert mask == sorted(set(mask)) and len(mask) == 269
    per_col = [sum(1 for c, _ in mask if c == k) for k in range(10)]
    assert per_col == survey_masks.STARKNET_CELLS_PER_COLUMN
    assert [max(o for c, o in mask if c == k) for k in range(10)] == survey_masks.STARKNET_MAX_OFFSET
    assert [len(p["ood_trace"]) for p in golden("saved_proofs.json")][:2] == [269, 269]


def test_periodic_columns_are_the_references_polynomials():
    from sandstorm_amd.layouts import starknet as sk
    with open(os.path.join(ROOT, "tests", "golden", "starknet_periodic_fingerprints.json")) as f:
        want = json.load(f)
    assert len(want) == sk.NUM_PERIODIC
    for table, key in enumerate(want):
        coeffs = sk.periodic_coefficients(table)
        assert len(coeffs) == want[key]["count"], key
        assert hashlib.sha256(",".join(str(v) for v in coeffs).encode()).hexdigest() == want[key]["sha256"], key
        values, period = sk.periodic_columns()[table]
        w = pow(3, (P - 1) // len(values), P)
        for j in (0, 1, len(values) - 1):
            acc = 0
            for c in reversed(coeffs):
                acc = (acc * pow(w, j, P) + c) % P
            assert acc == values[j]


def test_poseidon_keys():
    from sandstorm_amd.layouts import starknet as sk
    # StarkWare's example (builtins/src/poseidon/mod.rs zero_hash_matches_starkware_example)
    assert sk.poseidon_states((0, 0, 0))[2] == [
        3446325744004048536138401612021367625846492093718951375866996507163446763827,
        1590252087433376791875644726012779423683501236913937337746052470473806035332,
        867921192302518434283879514999422690776342565400001269945778456016268852423]
    keys = sk.poseidon_air_keys()
    # the literals of air.rs:2052, 2065, 2123, 2137, 2150
    assert keys["margin_full_to_partial"][1:] == [
        2006642341318481906727563724340978325665491359415674592697055778067937914672,
        427751140904099001132521606468025610873158555767197326325930641757709538586]
    assert keys["margin_partial_to_full"] == [
        560279373700919169769089400651532183647886248799764942664266404650165812023,
        1401754474293352309994371631695783042590401941592571735921592823982231996415,
        1246177936547655338400308396717835700699368047388302793172818304164989556526]
    # the keys are identities of the permutation: the same constants come out on any other input
    full, s, _ = sk.poseidon_states((5, 7, 11))
    c = [pow(v, 3, P) for v in s]
    assert [(s[k + 3] - (8 * c[k] + 4 * s[k + 1] + 6 * c[k + 1] + 2 * s[k + 2] - 2 * c[k + 2])) % P for k in range(80)] == keys["partial"]
    assert (full[4][1] - (4 * c[81] + 2 * s[82] + c[82])) % P == keys["margin_partial_to_full"][1]


def test_ecdsa_dummy_instance_and_signature_checks():
    from sandstorm_amd.layouts import starknet as sk
    pub_x, msg, r, w = sk.ecdsa_dummy_instance()
    assert pub_x == sk.GENERATOR[0] and 0 < r < 1 << 251 and 0 < w < 1 << 251
    t = sk.EcdsaInstanceTrace(pub_x, msg, r, w)
    assert sk._ec_add(t.wb, sk._ec_neg(sk.SHIFT_POINT))[0] == r and t.pubkey[0] == pub_x
    with pytest.raises(ValueError, match="signature is invalid"):
        sk.EcdsaInstanceTrace(pub_x, msg + 1, r, w)
    x = 2
    while sk._sqrt((pow(x, 3, P) + x + sk.CURVE_BETA) % P) is not None:
        x += 1
    with pytest.raises(ValueError, match="not on the curve"):
        sk.EcdsaInstanceTrace(x, msg, r, w)
    assert sk._sqrt(49) in (7, P - 7) and (sk.GENERATOR[1] ** 2 - (sk.GENERATOR[0] ** 3 + sk.GENERATOR[0] + sk.CURVE_BETA)) % P == 0
    assert sk._ec_mul(sk.CURVE_ORDER - 1, sk.GENERATOR) == sk._ec_neg(sk.GENERATOR)


def test_domain_rows_are_the_zeros_of_their_zerofiers():
    """at a trace point g^r the multiplier prod(num) / prod(den) has a pole exactly on the rows the constraint is
    enforced on; the compound domains are also stated explicitly"""
    from sandstorm_amd.layouts import starknet as sk
    n = 1 << 15
    g = pow(3, (P - 1) // n, P)
    doms = {}
    for c in sk.constraints(sk.Hints(0, 0, 0, 0), CHALLENGES):
        doms[c.domain.name] = c.domain
    assert len(doms) == 42
    explicit = {
        sk.PEDERSEN_TRANSITION.name: lambda r: r % 256 != 255,
        sk.PEDERSEN_HASH_START.name: lambda r: r % 512 == 0,
        sk.EC_OP_TRANSITION.name: lambda r: r % 64 == 0 and (r % 16384) // 64 != 255,
        sk.ECDSA_TRANSITION.name: lambda r: r % 128 == 0 and (r % 32768) // 128 != 255,
        sk.ECDSA_STEP_251.name: lambda r: r % 32768 == 128 * 251,
        sk.EC_OP_STEP_252.name: lambda r: r % 16384 == 64 * 252,
        sk.BITWISE_TRANSITION.name: lambda r: r % 256 == 0 and r % 1024 != 768,
        sk.EVERY_16_BIT_SEGMENT.name: lambda r: r % 16 == 0 and r % 1024 < 256,
        sk.POSEIDON_ADDR_STEP.name: lambda r: r % 64 == 0 and (r % 512) // 64 <= 4,
        sk.POSEIDON_PARTIAL1_SQUARING.name: lambda r: r % 16 == 0 and (r % 512) // 16 <= 21,
        sk.POSEIDON_HALF_FULL_ROUND_TRANSITION.name: lambda r: r % 64 == 0 and (r % 256) // 64 != 3,
        sk.POSEIDON_PARTIAL_ROUND0.name: lambda r: r % 8 == 0 and (r % 512) // 8 <= 60,
        sk.POSEIDON_PARTIAL_ROUND1.name: lambda r: r % 16 == 0 and (r % 512) // 16 <= 18,
    }
    assert set(explicit) <= set(doms)
    gp = {}
    for name, d in doms.items():
        want = set(d.rows(n))
        if name in explicit:
            assert want == {r for r in range(n) if explicit[name](r)}, name
        got = set()
        factors = [(p_, pow(g, e, P)) for p_, e in d.den(n)], [(p_, pow(g, e, P)) for p_, e in d.num(n)]
        for p_ in {p_ for fs in factors for p_, _ in fs}:
            if p_ not in gp:
                step, acc, vals = pow(g, p_, P), 1, []
                for _ in range(n):
                    vals.append(acc)
                    acc = acc * step % P
                gp[p_] = vals
        for r in range(n):
            den_zero = any(gp[p_][r] == c for p_, c in factors[0])
            num_zero = any(gp[p_][r] == c for p_, c in factors[1])
            if den_zero and not num_zero:
                got.add(r)
        assert got == want, name


def test_multiplier_tables_are_the_zerofier_quotients():
    """the composition's tables (periodic multipliers, full-length inverses, periodic columns) against the definitions:
    at a random point, and entry by entry on the LDE coset"""
    from sandstorm_amd import air_program as ap
    from sandstorm_amd.layouts import starknet as sk
    n = 1 << 15
    tables = sk.Tables(n)
    rng = random.Random(5)
    x = rng.randrange(P)
    for c in sk.constraints(sk.Hints(0, 0, 0, 0), CHALLENGES):
        got = ap.evaluate(tables.multiplier(c.domain), P, x, None, lambda t: tables.value_at(tables.specs[t], x))
        assert got == c.domain.multiplier_at(n, x), c.domain.name
    w = pow(3, (P - 1) // (2 * n), P)
    for spec in tables.specs:
        if spec[0] == "inverse" or tables.length(spec) > 4096:
            continue
        vals = tables.host_values(spec)
        assert len(vals) == tables.length(spec)
        for i in (0, 1, len(vals) - 1):
            assert vals[i] == tables.value_at(spec, 3 * pow(w, i, P) % P), spec
        if spec[0] == "column":                                     # on the trace domain a periodic column takes its values
            values, period = sk.periodic_columns()[spec[1]]
            gn = pow(3, (P - 1) // n, P)
            for row in (0, period // len(values), period + 3 * (period // len(values))):
                assert tables.value_at(spec, pow(gn, row, P)) == sk.periodic_value(spec[1], row)


def test_constraints_vanish_on_the_example_trace(example):
    sk, cols, constraints = example
    n = len(cols[0])
    assert n == 1 << 21 and len(cols) == 10
    for c in constraints:
        assert sk.failing_rows(c, cols, sample(c.domain.rows(n))) == [], c.name


def test_a_corrupted_cell_trips_its_constraints(example):
    sk, cols, constraints = example
    n = len(cols[0])
    by_name = {c.name: c for c in constraints}
    probes = [  # (column, row, a constraint that must notice, a row of its domain that reads the cell)
        (sk.COL_PEDERSEN_X, 512 * 2 + 300, ("pedersen/hash0/ec_subset_sum/add_points/x", "pedersen/hash0/ec_subset_sum/copy_point/x"), 512 * 2 + 299),
        (sk.COL_AUXILIARY, 32768 + 64 * 10 + sk.Ecdsa.PUBKEY_DOUBLING_Y, "ecdsa/signature0/doubling_key/y", 32768 + 64 * 10),
        (sk.COL_AUXILIARY, 32768 + 128 * 3 + sk.Ecdsa.GENERATOR_PARTIAL_SUM_X, "ecdsa/signature0/exponentiate_generator/add_points/x_diff_inv", 32768 + 128 * 3),
        (sk.COL_AUXILIARY, 32768 + sk.Ecdsa.R_POINT_SLOPE, "ecdsa/signature0/extract_r/x", 32768),
        (sk.COL_AUXILIARY, 16384 * 2 + 64 * 200 + sk.EcOp.R_PARTIAL_SUM_Y, ("ec_op/ec_subset_sum/add_points/y", "ec_op/ec_subset_sum/copy_point/y"), 16384 * 2 + 64 * 200),
        (sk.COL_AUXILIARY, 16384 * 2 + sk.EcOp.M_BIT251_AND_BIT196, "ec_op/ec_subset_sum/bit_unpacking/cumulative_bit196", 16384 * 2),
        (sk.COL_AUXILIARY, 512 * 5 + 64 * 2 + 53, "poseidon/poseidon/full_rounds_state0_squaring", 512 * 5 + 128),
        (sk.COL_RANGE_CHECK, 512 * 5 + 8 * 30 + 3, "poseidon/poseidon/partial_round0", 512 * 5 + 8 * 27),
        (sk.COL_AUXILIARY, 512 * 5 + 16 * 21 + 6, "poseidon/poseidon/margin_partial_to_full1", 512 * 5),
        (sk.COL_NPC, 512 * 5 + 231, "poseidon/poseidon/last_full_round0", 512 * 5),
        (sk.COL_RANGE_CHECK, 1024 * 2 + 256 + 17, "bitwise/partition", 1024 * 2 + 256),
        (sk.COL_RANGE_CHECK, 256 * 3 + 32 * 2 + 12, "rc_builtin/value", 256 * 3),
        (sk.COL_PERMUTATION, 8 * 100 + 3, "diluted_check/step", 8 * 100),
        (sk.COL_FLAGS, 16 * 9 + 3, "cpu/decode/opcode_rc/bit", 16 * 9 + 3),
    ]
    for col, row, names, at in probes:
        names = (names,) if isinstance(names, str) else names           # which of two fires depends on the bit of the step
        assert all(sk.failing_rows(by_name[name], cols, [at]) == [] for name in names)
        old = cols[col][row]
        cols[col][row] = (old + 2) % P
        try:
            assert any(sk.failing_rows(by_name[name], cols, [at]) == [at] for name in names), names
        finally:
            cols[col][row] = old


def test_trace_generation_refuses_what_the_reference_refuses():
    from sandstorm_amd.layouts import starknet as sk
    states, memory, spi = starknet_example(11)
    with pytest.raises(ValueError, match="do not fit the trace"):
        sk.base_trace(states, memory, spi)                        # 2^15 rows: the diluted check cannot hold its 65536 values
    with pytest.raises(ValueError, match="at least"):
        sk.base_trace(states[:1024], memory, spi)
    with pytest.raises(ValueError, match="power of two"):
        sk.base_trace(states[:1000], memory, spi)


def test_cpp_air_is_the_python_one(oracle):
    """sandstorm_amd/host/air_starknet.cpp against layouts/starknet.py: the same mask, the same table set, and - both
    lowered programs run by the oracle's constraint VM on one random evaluation domain (2^15-row trace, blowup 2) with
    one random table per table description - the same composition at every point"""
    import numpy as np
    from sandstorm_amd import air_program as ap
    from sandstorm_amd import hostlib
    from sandstorm_amd.layouts import starknet as sk
    _, _, spi = starknet_example(11)
    log_n = 15
    n, N = 1 << log_n, 2 << log_n
    alpha = pow(5, 77, P)
    hints = sk.Hints.from_public_input(spi, CHALLENGES, n)
    tables = sk.Tables(n)
    prog = ap.lower(sk.composition(n, hints, CHALLENGES, alpha, tables), P)
    cpp = hostlib.StarknetHostAir(None, spi, log_n)
    assert cpp.mask_size == 269 and (cpp.num_base_columns, cpp.num_extension_columns) == (9, 1)
    code, consts, n_slots, specs = cpp.dump(n, [oracle.to_mont([c])[0] for c in CHALLENGES], oracle.to_mont([alpha])[0])
    # Air::prepare_program: lowered ahead of the composition coefficient, its 195 powers patched in - the same words
    cpp.prepare(n, [oracle.to_mont([c])[0] for c in CHALLENGES])
    for a2 in (alpha, alpha + 12345):
        got2 = cpp.dump(n, [oracle.to_mont([c])[0] for c in CHALLENGES], oracle.to_mont([a2])[0])
        if a2 == alpha:
            assert np.array_equal(got2[0], code) and np.array_equal(got2[1], consts) and got2[2] == n_slots
        else:
            assert np.array_equal(got2[0], code) and not np.array_equal(got2[1], consts)
    cpp.close()
    assert sorted(map(repr, specs)) == sorted(map(repr, tables.specs)) and specs[:9] == tables.specs[:9]
    rng = np.random.default_rng(11)
    def rand(count):                                              # any limbs with the top one below 2^59 are felts below p
        v = rng.integers(0, 1 << 63, size=(count, 4), dtype=np.uint64)
        v[:, 3] &= np.uint64((1 << 59) - 1)
        return np.ascontiguousarray(v)
    by_spec = {spec: rand(tables.length(spec)) for spec in tables.specs}

    def packed(order):
        desc, off = [], 0
        for spec in order:
            desc += [off, len(by_spec[spec]).bit_length() - 1]
            off += len(by_spec[spec])
        return np.concatenate([by_spec[s] for s in order]), desc
    lde = [rand(N) for _ in range(10)]
    g = oracle.to_mont([3])[0]
    tab, desc = packed(tables.specs)
    out_py = oracle.eval_program(prog.code, oracle.to_mont(prog.consts), tab, desc, prog.n_slots, lde, log_n, 1, g)
    tab, desc = packed(specs)
    out_cpp = oracle.eval_program(code, consts, tab, desc, n_slots, lde, log_n, 1, g)
    assert out_py.any() and np.array_equal(out_cpp, out_py)
    with pytest.raises(Exception, match="starknet layout"):
        hostlib.StarknetHostAir(None, load_run()[2], log_n)


def test_base_trace_is_the_one_the_references_proof_opens(oracle, golden):
    """`example/bootloader/bootloader-proof.bin` is the reference's own proof of the run shipped beside it.  The 100 rows of the
    base-trace LDE it opens (tests/golden/make_starknet_rows_golden.py) equal, in all 9 columns, the rows of the trace
    regenerated here from trace.bin / memory.bin / the public and private input: every value of a column's extension
    depends on all 2^21 cells of that column, so this pins the whole base-trace generation - CPU cells, the memory pool
    with gap fillers, sorted memory, the range-check and diluted pools with their padding, the real Pedersen instances
    and the DUMMY instances of all six builtins - together with the LDE coset (offset 3, blowup 2), against real
    prover output."""
    from sandstorm_amd.layouts import starknet as sk
    g = golden("starknet_opened_rows.json")
    states, memory, pi, private = bootloader_run()
    cols = sk.base_trace(states, memory, pi, private)
    assert len(cols[0]) == g["trace_len"] and len(g["positions"]) == 100 == len(set(g["positions"]))
    offset = oracle.to_mont([g["lde_offset"]])[0]
    for c, col in enumerate(cols):
        lde = oracle.lde(oracle.to_mont(col), 1, offset)[0]
        got = [int(v) for v in oracle.from_mont(lde[g["positions"]])]
        assert got == [int(row[c], 16) for row in g["rows"]], "column %d" % c
    # the pin discriminates: one changed cell anywhere in a column changes (almost) every opened value of it
    col = list(cols[sk.COL_AUXILIARY])
    col[123456] = (col[123456] + 1) % P
    lde = oracle.lde(oracle.to_mont(col), 1, offset)[0]
    got = [int(v) for v in oracle.from_mont(lde[g["positions"]])]
    assert sum(a == int(row[sk.COL_AUXILIARY], 16) for a, row in zip(got, g["rows"])) == 0


def test_cpp_base_trace_equals_the_python_one(oracle):
    """sandstorm_amd/host/trace_starknet.cpp against layouts/starknet.py::base_trace, cell for cell, on the reference's
    bootloader run with real instances of every builtin added (the other slots hold the dummies of the trace the
    reference's proof opens)"""
    import numpy as np
    from sandstorm_amd import hostlib
    from sandstorm_amd.layouts import starknet as sk
    g = os.path.join(ROOT, "tests", "golden")
    with gzip.open(os.path.join(g, "bootloader", "trace.bin.gz")) as f:
        trace_bin = f.read()
    with gzip.open(os.path.join(g, "bootloader", "memory.bin.gz")) as f:
        memory_bin = f.read()
    pi, both, python_cols = bootloader_trace_with_real_instances()
    for priv in (both,):
        want = python_cols
        got = hostlib.starknet_base_trace(trace_bin, memory_bin, pi, priv)
        assert len(got) == len(want) == 9
        for c, (a, b) in enumerate(zip(got, want)):
            assert np.array_equal(a, oracle.to_mont(b)), "column %d" % c
    from sandstorm_amd._lib import SandstormHipError
    with pytest.raises(SandstormHipError, match="signature is invalid"):
        bad = dict(both)
        bad["ecdsa"] = [(1,) + tuple(both["ecdsa"][0][1:4]) + (both["ecdsa"][0][4] + 1,)]
        hostlib.starknet_base_trace(trace_bin, memory_bin, pi, bad)
    with pytest.raises(SandstormHipError, match="at least 2048"):
        hostlib.starknet_base_trace(trace_bin[:24 * 1024], memory_bin, pi)


def test_references_own_starknet_proof_verifies(golden):
    """`example/array-sum.proof.saved` (committed as tests/golden/reference_array_sum_starknet.proof) is the reference's own
    proof of its array-sum example under the starknet layout (EthVerifierClaim: masked Keccak trees, Solidity coin), 2^17
    steps.  This repo's verifier accepts it from nothing but the public input: the seed (public_input.py), the whole
    Fiat-Shamir transcript in order (M10) down to the proof of work and the 16 query positions, the out-of-domain
    identity of the restated 195-constraint AIR under the replayed challenges, every Merkle opening, the DEEP value of
    every query, the six FRI layers and the remainder.  The public input is the array-sum run re-declared for the
    layout (starknet_example) - the run's trace is also the one the proof opens."""
    import functools
    from sandstorm_amd import backend as be, hostlib, public_input, verifier
    from sandstorm_amd.layouts import starknet as sk
    from sandstorm_amd.prover import Conventions
    with open(os.path.join(ROOT, "tests", "golden", "reference_array_sum_starknet.proof"), "rb") as f:
        raw = f.read()
    _, _, spi = starknet_example(17)
    seed = public_input.public_coin_seed(spi, be.COIN_SOLIDITY)
    # the proof's own options are 16 queries + 16 grinding bits at blowup 2: 32 conjectured bits (the CLI's default
    # requirement of 80 rejects it, as the reference's `verify` would)
    verify = functools.partial(verifier.verify, required_security_bits=32)
    cverify = functools.partial(hostlib.verify, required_security_bits=32)
    args = (sk.verifier_air(spi), be.TREE_KECCAK_M20, be.COIN_SOLIDITY)
    positions = verify(raw, *args, seed)                                                   # default conventions = the reference's
    assert len(positions) == 16 and positions == sorted(positions)
    assert positions[:4] == golden("saved_proof_openings.json")["positions"][:4]        # the positions the openings golden was cut at
    assert verify(raw, *args, seed, expected_options=[16, 2, 16, 8, 16]) == positions
    # what the acceptance rests on
    with pytest.raises(verifier.VerificationError, match="32 bits of conjectured security, 80 required"):
        verifier.verify(raw, *args, seed)
    with pytest.raises(verifier.VerificationError, match="options differ"):
        verify(raw, *args, seed, expected_options=[65, 2, 16, 8, 16])
    with pytest.raises(verifier.VerificationError, match="proof of work"):
        verify(raw, *args, bytes(32))                                                      # another seed: another statement
    with pytest.raises(verifier.VerificationError, match="does not fold"):
        verify(raw, *args, seed, Conventions(fri_alpha_times_offset=False))               # the bare draw as FRI challenge
    import copy
    other = copy.deepcopy(spi)
    other.rc_max += 1
    with pytest.raises(verifier.VerificationError, match="proof of work"):
        verify(raw, sk.verifier_air(other), be.TREE_KECCAK_M20, be.COIN_SOLIDITY, public_input.public_coin_seed(other, be.COIN_SOLIDITY))
    with pytest.raises(verifier.VerificationError, match="out-of-domain identity"):       # same seed, one hint of the AIR off by one
        verify(raw, sk.verifier_air(other), be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed)
    bad = bytearray(raw)
    bad[len(raw) - 40] ^= 1                                                                # inside the out-of-domain vector
    with pytest.raises(verifier.VerificationError):
        verify(bytes(bad), *args, seed)
    # the C++ host: its verifier (host/verifier.cpp) with its own starknet AIR (host/air_starknet.cpp) and seed accepts it too
    from sandstorm_amd._lib import SandstormHipError
    assert hostlib.public_coin_seed(spi, be.COIN_SOLIDITY)[0] == seed
    air = hostlib.StarknetHostAir(None, spi, 21)
    assert cverify(air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, raw) == positions
    assert cverify(air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, raw, expected_options=[16, 2, 16, 8, 16]) == positions
    with pytest.raises(SandstormHipError, match="32 bits of conjectured security, 80 required"):
        hostlib.verify(air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, raw)
    with pytest.raises(SandstormHipError, match="options differ"):
        cverify(air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, raw, expected_options=[16, 2, 16, 8, 8])
    with pytest.raises(SandstormHipError, match="does not fold"):
        cverify(air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, raw, fri_alpha_times_offset=False)
    with pytest.raises(SandstormHipError):
        cverify(air, be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, bytes(bad))
    air.close()


def test_array_sum_trace_is_the_one_the_references_proof_opens(oracle, golden):
    """the 16 base-trace rows that proof opens are rows of the LDE of the trace regenerated from example/trace.bin /
    memory.bin re-declared for the starknet layout (starknet_example): all dummy builtin instances, no Pedersen"""
    from sandstorm_amd import wire
    from sandstorm_amd.layouts import starknet as sk
    from sandstorm_amd.prover import bitrev
    with open(os.path.join(ROOT, "tests", "golden", "reference_array_sum_starknet.proof"), "rb") as f:
        w = wire.parse(f.read())
    positions = golden("saved_proof_openings.json")["positions"]
    spi, cols = array_sum_trace()
    offset = oracle.to_mont([3])[0]
    natural = [bitrev(p, 22) for p in positions]                       # committed index -> exponent of w (M3)
    for c in (0, 5, 6, 7, 8):                                           # flags, memory pool, sorted memory, range check, auxiliary
        lde = oracle.lde(oracle.to_mont(cols[c]), 1, offset)[0]
        got = [int(v) for v in oracle.from_mont(lde[natural])]
        assert got == [int(w.base_rows[9 * q + c]) for q in range(len(positions))], "column %d" % c


def test_extension_column_is_the_one_the_references_proof_opens(oracle):
    """A2 pinned by reference OUTPUT: replaying the transcript of `example/array-sum.proof.saved` gives
    the six challenges its prover drew after the base commitment; the extension column rebuilt from them with the
    oracle's `build_extension_columns` (memory, range-check and diluted-check products + the diluted aggregate, all in
    the one permutation column: layouts/src/starknet/trace.rs:997-1100), extended over 3<w>, equals the 16 extension
    leaves the proof opens.  One wrong challenge, or one changed cell, changes every opened value.  The GPU twin is
    tests/test_gpu_extension.py::test_extension_column_of_the_reference_proof (and the byte-for-byte proof of
    tests/test_gpu_reference_proof.py)."""
    import numpy as np
    from sandstorm_amd import backend as be, public_input, wire
    from sandstorm_amd.coin import PublicCoin
    from sandstorm_amd.layouts import starknet as sk
    from sandstorm_amd.prover import bitrev
    with open(os.path.join(ROOT, "tests", "golden", "reference_array_sum_starknet.proof"), "rb") as f:
        w = wire.parse(f.read())
    spi, cols = array_sum_trace()
    coin = PublicCoin(be.COIN_SOLIDITY, public_input.public_coin_seed(spi, be.COIN_SOLIDITY))
    coin.reseed_with_digest(w.base_root)
    challenges = [coin.draw() for _ in range(6)]
    positions = reference_query_positions(w, spi)
    assert len(positions) == 16 == len(w.extension_rows)
    n = len(cols[0])
    aux = {"npc": oracle.to_mont(cols[sk.COL_NPC]), "memory": oracle.to_mont(cols[sk.COL_MEMORY]), "range_check": oracle.to_mont(cols[sk.COL_RANGE_CHECK])}
    natural = [bitrev(p, 22) for p in positions]
    offset = oracle.to_mont([3])[0]

    def opened(ch):
        ext, lasts = oracle.build_extension_columns("starknet", aux, ch, n)
        lde = oracle.lde(ext[0], 1, offset)[0]
        return [int(v) for v in oracle.from_mont(lde[natural])], lasts
    got, lasts = opened(challenges)
    assert got == [int(v) for v in w.extension_rows]
    hints = sk.Hints.from_public_input(spi, [wire._canon(c) for c in challenges], n)
    assert [int(v) for v in oracle.from_mont(np.stack(lasts))] == [hints.memory_quotient, 1, 1]
    # the pin discriminates
    wrong = list(challenges)
    wrong[3] = oracle.to_mont([wire._canon(challenges[3]) + 1])[0]         # the diluted-check permutation challenge
    bad, _ = opened(wrong)
    assert sum(a == int(b) for a, b in zip(bad, w.extension_rows)) == 0


def reference_query_positions(w, spi):
    """the query positions of the reference's proof, by replaying its transcript (the verifier's step 1)"""
    from sandstorm_amd import backend as be, public_input, verifier
    from sandstorm_amd.layouts import starknet as sk
    seed = public_input.public_coin_seed(spi, be.COIN_SOLIDITY)
    return verifier.verify(w, sk.verifier_air(spi), be.TREE_KECCAK_M20, be.COIN_SOLIDITY, seed, required_security_bits=32)
