"""Host logic: the Expr -> program lowering (sandstorm_amd/air_program.py) against
direct big-integer evaluation of the same DAG, through the CPU oracle's
interpreter.  No GPU."""
import random

import numpy as np

from sandstorm_amd import air_program as ap
from tests import pyref
from tests.pyref import P
from tests.util import random_column


def random_dag(rng, ncols, ntables, nconsts, size):
    consts = [rng.randrange(P) for _ in range(nconsts)]
    pool = [ap.X] + [ap.Const(c) for c in consts]
    pool += [ap.Trace(c, o) for c in range(ncols) for o in (0, 1, 3)]
    pool += [ap.Table(t) for t in range(ntables)]
    for _ in range(size):
        k = rng.random()
        a, b = rng.choice(pool), rng.choice(pool)
        if k < 0.35:
            e = a * b
        elif k < 0.6:
            e = a + b
        elif k < 0.85:
            e = a - b
        elif k < 0.93:
            e = a * a
        else:
            e = (a * b + 1).inverse() if rng.random() < 0.5 else a ** rng.choice([2, 3, 5, 8])
        pool.append(e)
    root = pool[-1]
    for e in pool[-8:-1]:            # make sure late nodes are shared and all used
        root = root * e + e
    return root


def run_case(oracle, seed, size):
    rng = random.Random(seed)
    log_n, lb, ncols, ntables = 3, 1, 3, 2
    n, N = 1 << log_n, 1 << (log_n + lb)
    g = oracle.to_mont([3])[0]
    cols = [oracle.lde(random_column(n, c + seed), lb, g)[0] for c in range(ncols)]
    tabs = [random_column(4, 50 + seed), random_column(8, 60 + seed)]
    tables = np.concatenate(tabs)
    desc = [0, 2, 4, 3]
    root = random_dag(rng, ncols, ntables, 5, size)
    prog = ap.lower(root, P)
    out = oracle.eval_program(prog.code, oracle.to_mont(prog.consts) if prog.consts else [], tables, desc,
                              prog.n_slots, cols, log_n, lb, g)
    colv = [list(oracle.from_mont(c)) for c in cols]
    tabv = [list(oracle.from_mont(t)) for t in tabs]
    wN = pyref.root_of_unity(N)
    for i in range(N):
        x = 3 * pow(wN, i, P) % P
        want = ap.evaluate(root, P, x, lambda c, o: colv[c][(i + (o << lb)) % N], lambda t: tabv[t][i % len(tabv[t])])
        assert oracle.from_mont(out[i]) == want, (seed, i)
    return prog


def test_lowering_random_dags(oracle):
    for seed in range(12):
        prog = run_case(oracle, seed, 40 + 15 * seed)
        assert prog.n_instr > 10


def test_lowering_shares_nodes():
    t = ap.Trace(0, 0) * ap.Trace(1, 0) + ap.Const(7)
    e = t * t + t * ap.X + (t - ap.X) * t
    prog = ap.lower(e, P)
    muls = sum(1 for i in range(prog.n_instr) if (prog.code[2 * i] & 0xff) == ap.OP.MUL)
    assert muls == 4          # T0*T1 once, then three products with the shared node


def test_deep_accumulator_chain_spills():
    """an expression needing more than four live partial results uses slots"""
    def tree(d, k):
        if d == 0:
            return ap.Trace(k % 3, k % 2) * ap.Const(k + 2) + ap.X
        return tree(d - 1, 2 * k) * tree(d - 1, 2 * k + 1)
    prog = ap.lower(tree(5, 1), P)
    assert prog.n_slots >= 1
