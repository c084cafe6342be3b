"""The `recursive` layout restatement (sandstorm_amd/layouts/recursive.py): the AIR constraints must vanish on the
base trace generated from the reference's own example run (tests/golden/example/: `cairo-run` output of array-sum,
2^14 steps) — constraints (layouts/src/recursive/air.rs) and trace generation (trace.rs) are restated independently
and validate each other; a corrupted cell must trip the constraint that reads it."""
import os
import random

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "tests", "golden", "example")


@pytest.fixture(scope="module")
def example():
    from sandstorm_amd import binary, public_input
    from sandstorm_amd.layouts import recursive as rec
    with open(os.path.join(EX, "trace.bin"), "rb") as f:
        states = binary.read_register_states(f.read())
    with open(os.path.join(EX, "memory.bin"), "rb") as f:
        memory = binary.read_memory(f.read())
    pi = public_input.AirPublicInput.from_json(os.path.join(ROOT, "tests", "golden", "air_public_input_array_sum.json"))
    import numpy as np
    from oracle import oracle_py as oracle
    cols = rec.base_trace(states, memory, pi)
    n = len(cols[0])
    challenges = [pow(7, 11 + 3 * i, rec.P) for i in range(6)]
    # the extension columns from the ORACLE's build_extension_columns on the generated columns (the product's device
    # version is compared with it in tests/test_gpu_extension.py)
    aux = {"npc": oracle.to_mont(cols[rec.COL_NPC]), "memory": oracle.to_mont(cols[rec.COL_MEMORY]),
           "range_check": oracle.to_mont(cols[rec.COL_RANGE_CHECK]),
           "diluted_unordered": oracle.to_mont(cols[rec.COL_DILUTED_UNORDERED]),
           "diluted_ordered": oracle.to_mont(cols[rec.COL_DILUTED_ORDERED])}
    ext, lasts = oracle.build_extension_columns("recursive", aux, [oracle.to_mont([c])[0] for c in challenges], n)
    cols = cols + [[int(v) for v in oracle.from_mont(e)] for e in ext]
    hints = rec.Hints.from_public_input(pi, challenges, n)
    last_mem, last_rc, last_dc = (int(v) for v in oracle.from_mont(np.stack(lasts)))
    assert last_mem == hints.memory_quotient and last_rc == 1 and last_dc == 1     # the reference's own asserts (trace.rs:734, 757)
    return rec, cols, rec.constraints(hints, challenges)


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
    assert len(constraints) >= 50 and len({c.name for c in constraints}) == len(constraints)
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
