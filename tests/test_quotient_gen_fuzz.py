"""tools/gen_quotient.py on programs it has never seen: random constraint programs in the shapes the layouts' compositions have
(x y - z, L - (P1 + P2 + P3 + T), sums of constant x cell, x^2 - x, products of products, sign flips through RSUB, products that
wait in scratch slots or are used twice, single-constraint groups scaled by a constant and a table), generated with every
combination of the generator's knobs (wide sums per constraint, lazy subtrahends, constants as factors, minimum products per wide
sum, fused dot products, prefetch depth), compiled for the host (tests/cpp/quotient_gen_host_test.cpp) and held to the oracle's
constraint VM on columns drawn from the edges of the limb forms.  The two real programs exercise one path through the
generator's state machine each; these exercise its corners - and its fallback (a constraint whose half-summed value is used in
a way the emission cannot express goes back to plain products)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.test_quotient_gen_host import CPP, ROOT, _extreme, run_host
from tests.test_layout_starknet import P

sys.path.insert(0, os.path.join(ROOT, "tools"))

OP_MOV, OP_ADD, OP_SUB, OP_RSUB, OP_MUL, OP_INV, OP_ST, OP_OUT = range(8)
ACC, SLOT, CONST, TRACE, TABLE, X = range(6)
NCOLS, NCONSTS, NTABLES, NSLOTS = 6, 24, 5, 6
LOG_N = 6


class Builder:
    """one random program: out = sum over groups of (sum over constraints of C_k alpha_k) x table [x table]"""

    def __init__(self, rng):
        self.rng, self.ins = rng, []

    def emit(self, op, d, kind=0, w1=0):
        self.ins.append((op, d, kind, int(w1)))

    def cell(self):
        return (TRACE, (int(self.rng.integers(NCOLS)) << 24) | int(self.rng.integers(0, 5)))

    def leaf(self):
        k = self.rng.integers(10)
        if k < 6:
            return self.cell()
        if k < 8:
            return (CONST, int(self.rng.integers(NCONSTS)))
        if k < 9:
            return (TABLE, int(self.rng.integers(NTABLES)))
        return (X, 0)

    def linear(self, d, terms=None):
        """acc d = a random signed sum of leaves"""
        terms = int(self.rng.integers(1, 4)) if terms is None else terms
        self.emit(OP_MOV, d, *self.leaf())
        for _ in range(terms - 1):
            self.emit([OP_ADD, OP_SUB, OP_RSUB][int(self.rng.integers(3))], d, *self.leaf())

    def constraint(self, d, t):
        """the value of one constraint in acc d (t: a free accumulator), in one of the shapes of the layouts' constraints"""
        r, e = self.rng, self.emit
        shape = int(r.integers(9))
        if shape == 0:                                           # x y - z ...
            self.linear(d)
            e(OP_MUL, d, *self.cell())
            for _ in range(int(r.integers(0, 3))):
                e([OP_ADD, OP_SUB, OP_RSUB][int(r.integers(3))], d, *self.leaf())
        elif shape == 1:                                         # L - (P1 + P2 + ... + T): products parked in scratch slots
            self.linear(d)
            nprod = int(r.integers(2, 5))
            for j in range(nprod):
                self.linear(t)
                e(OP_MUL, t, *self.cell())
                if j < nprod - 1:
                    e(OP_ST, t, 0, j)
            for j in range(nprod - 1):
                e(OP_ADD if r.integers(2) else OP_SUB, t, SLOT, j)
            e(OP_ADD, t, *self.cell())
            e(OP_SUB if r.integers(2) else OP_ADD, d, ACC, t)
        elif shape == 2:                                         # sum of constant x cell
            e(OP_MOV, d, *self.cell())
            for _ in range(int(r.integers(2, 15))):
                e(OP_MOV, t, CONST, int(r.integers(NCONSTS)))
                e(OP_MUL, t, *self.cell())
                e(OP_ADD if r.integers(3) else OP_SUB, d, ACC, t)
        elif shape == 3:                                         # x^2 - x with x = cell - 2 cell'
            c = self.cell()
            e(OP_MOV, d, *c)
            e(OP_ADD, d, *c)
            e(OP_RSUB, d, *self.cell())
            e(OP_ST, d, 0, 5)
            e(OP_MUL, d, ACC, d)
            e(OP_SUB, d, SLOT, 5)
        elif shape == 4:                                         # (x y) z - w: a product of a product
            self.linear(d, 1)
            e(OP_MUL, d, *self.cell())
            e(OP_MUL, d, *self.cell())
            e(OP_SUB, d, *self.leaf())
        elif shape == 5:                                         # a product behind a sign flip
            self.linear(d)
            e(OP_MUL, d, *self.cell())
            e(OP_RSUB, d, *self.cell())
            e(OP_ADD, d, *self.leaf())
        elif shape == 6:                                         # a product used twice (never a wide term)
            self.linear(t)
            e(OP_MUL, t, *self.cell())
            e(OP_ST, t, 0, 4)
            e(OP_MOV, d, SLOT, 4)
            e(OP_MUL, d, *self.cell())
            e(OP_ADD, d, SLOT, 4)
        elif shape == 7:                                         # (a - b)(c - d) + (e - f)(g - h): lazy factors on both sides
            self.linear(d, 2)
            self.linear(t, 2)
            e(OP_MUL, d, ACC, t)
            self.linear(t, 2)
            e(OP_ST, t, 0, 3)
            self.linear(t, 2)
            e(OP_MUL, t, SLOT, 3)
            e(OP_ADD if r.integers(2) else OP_SUB, d, ACC, t)
        else:                                                    # linear only
            self.linear(d, int(r.integers(1, 5)))

    def program(self, first):
        r, e = self.rng, self.emit
        out_started = False
        for _ in range(int(r.integers(1, 4))):
            if r.integers(4) == 0:                               # a single-constraint group: C x alpha x table
                self.constraint(1, 2)
                e(OP_MUL, 1, CONST, int(r.integers(NCONSTS)))
            else:
                started = False
                for _ in range(int(r.integers(1, 6))):
                    self.constraint(2, 3)
                    e(OP_MUL, 2, CONST, int(r.integers(NCONSTS)))
                    if started:
                        e(OP_ADD, 1, ACC, 2)
                    else:
                        e(OP_MOV, 1, ACC, 2)
                        started = True
            for _ in range(int(r.integers(1, 3))):
                e(OP_MUL, 1, TABLE, int(r.integers(NTABLES)))
            if out_started:
                e(OP_ADD, 0, ACC, 1)
            else:
                e(OP_MOV, 0, ACC, 1)
                out_started = True
        e(OP_OUT, 0)
        return self.ins


def _add_mod_p(a, b):
    to_int = lambda v: [sum(int(x[k]) << (64 * k) for k in range(4)) for x in v]
    s = [(x + y) % P for x, y in zip(to_int(a), to_int(b))]
    return np.array([[(v >> (64 * k)) & ((1 << 64) - 1) for k in range(4)] for v in s], dtype=np.uint64)


KNOBS = [  # (fused dot products, wide sums, lazy subtrahends, constants as factors, minimum products, prefetch depth)
    (True, True, True, True, 1, 3), (True, True, False, False, 1, 2), (True, True, True, False, 2, 1), (True, False, True, True, 1, 3),
    (False, False, False, False, 1, 2), (True, True, False, True, 1, 4)]


@pytest.mark.parametrize("knobs", KNOBS, ids=lambda k: "fuse%d-wide%d-lazy%d-cf%d-min%d-d%d" % tuple(int(x) for x in k))
def test_random_programs_through_the_generator(oracle, knobs, tmp_path, monkeypatch):
    import gen_quotient
    fuse, wide, lazy_sub, const_factor, min_terms, depth = knobs
    tmp = str(tmp_path)
    monkeypatch.setattr(gen_quotient, "OUT_DIR", tmp)
    rng = np.random.default_rng(1000 + int(os.environ.get("SS_FUZZ_SEED", "0")) * 977 + sum(int(x) << i for i, x in enumerate(knobs)))
    programs = [Builder(rng).program(j == 0) for j in range(14)]
    # the tables that are only ever multipliers, read from their 2^24-fold copies (r280 products: tools/gen_quotient.py generate) - in
    # every second knob set, so that both forms of "sum x table" are in the sample
    scaled = []
    if depth % 2 == 1:
        uses = {}
        for ins in programs:
            for op, d, kind, w1 in ins:
                if op <= OP_MUL and kind == TABLE:
                    uses.setdefault(w1, set()).add(op)
        scaled = sorted(t for t, ops in uses.items() if ops == {OP_MUL})
    with open(os.path.join(tmp, "qg_scaled.h"), "w") as f:
        f.write("static const uint32_t QG_N_TABLES = %du, QG_N_SCALED = %du;\nstatic const uint32_t QG_SCALED_TABLES[] = {%s};\n"
                % (NTABLES, len(scaled), ", ".join("%du" % t for t in scaled) if scaled else "0u"))
    wide_terms = 0
    with open(os.path.join(tmp, "qg_parts.h"), "w") as f:
        for j, ins in enumerate(programs):
            stats = gen_quotient.generate_body("fuzz", ins, NCONSTS, NSLOTS, NTABLES, NCOLS, depth, "fuzz_p%d.inc" % j, fuse,
                                               "QG_OUT" if j == 0 else "QG_OUT_ACC", 0, wide, lazy_sub, const_factor, min_terms, scaled)
            wide_terms += stats["wide_terms"]
            f.write("static void run_lane_p%d(HostArgs &a, uint64_t lane, uint64_t lanes) {\n    QG_LANE_PRELUDE\n#include \"%s\"\n}\n"
                    % (j, os.path.join(tmp, "fuzz_p%d.inc" % j)))
        f.write("static const part_fn PARTS[] = {%s};\n" % ", ".join("run_lane_p%d" % j for j in range(len(programs))))
    if wide and fuse:
        assert wide_terms > 0, "no program of the sample got a wide sum: the sample does not test what it is meant to"
    exe = os.path.join(tmp, "qg_fuzz")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fopenmp", "-I", tmp, "-DQG_PARTS_H=\"qg_parts.h\"", "-DQG_SCALED_H=\"qg_scaled.h\"", "-o", exe, CPP])
    n, N = 1 << LOG_N, 2 << LOG_N
    tabs, desc, off = [], [], 0
    for t in range(NTABLES):
        length = 1 << int(rng.integers(1, LOG_N + 2))
        desc += [off, length.bit_length() - 1]
        off += length
        tabs.append(_extreme(rng, length))
    tab = np.concatenate(tabs)
    lde = [_extreme(rng, N) for _ in range(NCOLS)]
    consts = _extreme(rng, NCONSTS)
    g = oracle.to_mont([3])[0]
    w = oracle.to_mont([pow(3, (P - 1) // N, P)])[0]
    want = None
    for ins in programs:
        code = gen_quotient.encode(ins)
        v = oracle.eval_program(code, consts, tab, desc, NSLOTS, lde, LOG_N, 1, g)
        want = v if want is None else _add_mod_p(want, v)
    got = run_host(exe, tmp, lde, tab, desc, consts, N, 0, N - 1, 1, 24, g, w)
    assert np.array_equal(got, want)
