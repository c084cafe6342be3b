"""A small valid AIR used to exercise the whole proving pipeline end to end
(test infrastructure).  2 base columns + 1 extension column, 5 constraints:

    c0' = c1                    c1' = c0*c1 + c0            (all rows but the last)
    c0[0] = 1
    e0' = e0 * (gamma + c0')    e0[0] = gamma + c0[0]       (gamma = challenge 0)

Zerofiers sit inside the constraints as in the reference's AIRs
(layouts/src/recursive/air.rs:146-151): 1/(X^n - 1) is periodic on the LDE coset
(a 2-entry table for blowup 2), 1/(X - 1) is evaluated per point.
"""
import numpy as np

from sandstorm_amd import air_program as ap
from sandstorm_amd import backend as be
from sandstorm_amd.coin import canonical
from sandstorm_amd.prover import Air

P = be.P
MASK = [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0), (2, 1)]


def base_trace(n):
    c0, c1 = [1], [2]
    for _ in range(n - 1):
        a, b = c0[-1], c1[-1]
        c0.append(b)
        c1.append((a * b + a) % P)
    return c0, c1


def extension_trace(c0, gamma):
    e = [(gamma + c0[0]) % P]
    for i in range(1, len(c0)):
        e.append(e[-1] * (gamma + c0[i]) % P)
    return e


def composition(n, gamma, alpha, table0=None):
    """the composition constraint as an Expr DAG; table 0 = 1/(X^n - 1)"""
    w = pow(3, (P - 1) // n, P)
    last = ap.X - ap.Const(pow(w, n - 1, P))
    inv_all = ap.Table(0) if table0 is None else table0
    inv_first = (ap.X - 1).inverse()
    c0, c0n, c1, c1n, e0, e0n = (ap.Trace(c, o) for c, o in MASK)
    g = ap.Const(gamma)
    ks = [(c0n - c1) * last * inv_all,
          (c1n - c0 * c1 - c0) * last * inv_all,
          (c0 - 1) * inv_first,
          (e0n - e0 * (g + c0n)) * last * inv_all,
          (e0 - (g + c0)) * inv_first]
    total = None
    for k, c in enumerate(ks):
        term = c * ap.Const(pow(alpha, k, P))
        total = term if total is None else total + term
    return total


def make_air(oracle_to_mont):
    def build_program(n, challenges, comp_coeff):
        gamma, alpha = canonical(challenges[0]), canonical(comp_coeff)
        prog = ap.lower(composition(n, gamma, alpha), P)
        gn = pow(3, n, P)
        tab = [pow(gn - 1, -1, P), pow(-gn - 1, -1, P)]          # x_i^n = g^n * (-1)^i
        return prog, oracle_to_mont(tab), [0, 1]
    return Air("mini", 2, 1, 1, MASK, build_program)


def table_at(n, x):
    """1/(X^n - 1) at an arbitrary point (verifier side)"""
    return pow(pow(x, n, P) - 1, -1, P)
