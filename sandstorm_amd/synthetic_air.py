"""Layout-SHAPED synthetic AIRs for bench.py.

The reference's real constraint sets (93 constraints for `recursive`, 195 for
`starknet`; layouts/src/{recursive,starknet}/air.rs) are lowered by the host from
its `Expr` DAG.  The `recursive` set is restated (sandstorm_amd/layouts/recursive.py) but
needs a VALID trace, which exists only at the size of the reference's example run; the
`starknet` set is not restated yet.  For the benchmark sizes (2^16 .. 2^22 steps of
random columns) the bench therefore drives the same kernels with a synthetic
composition constraint that has the layout's *shape*: the same
column counts, the same mask (SURVEY.md §8a "Mask / zerofier note": 133 cells for
recursive — the exact list — and 269 for starknet with the per-column counts and
maximum offsets), degree-2 constraints, one alpha power per constraint (constraints
that share a zerofier are summed before the single multiplication by its inverse), periodic
zerofier-inverse tables with the layout's periods and full-length tables for the
single-point boundary zerofiers.  Timing is data-independent, so the cost profile
is representative; the VALUES prove nothing about Cairo.
"""
import random

import numpy as np

from . import air_program as ap
from . import backend as be
from .coin import canonical
from .prover import Air

P = be.P

# full recursive mask (column: row offsets)
RECURSIVE_MASK = {
    0: list(range(16)),
    1: [0, 1] + list(range(2, 33, 2)) + [33, 64, 65, 88, 90, 92, 94, 96, 97, 120, 122, 124, 126],
    2: [0, 1],
    3: [0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 12, 13, 16, 26, 27, 42, 43, 58, 74, 75, 91, 122, 123, 154, 202, 522, 523,
        1034, 1035, 2058],
    4: [0, 1, 2, 3],
    5: list(range(9)) + [12, 28, 44, 60, 76, 92, 108, 124, 1021, 1023, 1025, 1027, 2045],
    6: [0, 1, 2, 3, 4, 5, 7, 9, 11, 13, 17, 25, 768, 772, 784, 788, 1004, 1008, 1022, 1024],
    7: [0, 1], 8: [0, 1], 9: [0, 1, 2, 5],
}
STARKNET_CELLS_PER_COLUMN = [16, 5, 4, 9, 2, 60, 4, 56, 105, 8]   # 269 cells in total
STARKNET_MAX_OFFSET = [15, 511, 256, 256, 255, 33158, 3, 1009, 32763, 15]

LAYOUTS = {
    # name: (base cols, ext cols, constraints, zerofier periods (trace rows per period), periodic column lengths)
    "recursive": (7, 3, 93, [1, 2, 4, 16, 32, 128, 1024, 2048], [2048, 2048]),
    "starknet": (9, 1, 195, [1, 2, 4, 8, 16, 64, 128, 256, 512, 1024, 16384, 32768], [512, 512, 32768, 32768, 512, 512, 512, 64, 32]),
}
N_POINT_ZEROFIERS = 10       # single-point boundary zerofiers (SURVEY.md §8a note (i))


def layout_mask(name):
    if name == "recursive":
        return sorted((c, o) for c, offs in RECURSIVE_MASK.items() for o in offs)
    cells = []
    for c, (cnt, mx) in enumerate(zip(STARKNET_CELLS_PER_COLUMN, STARKNET_MAX_OFFSET)):
        offs = set(range(min(cnt, mx + 1)))
        rng = random.Random(1000 + c)
        offs = set(list(offs)[: max(1, cnt // 2)]) | {mx}
        while len(offs) < cnt:
            offs.add(rng.randrange(mx + 1))
        cells += [(c, o) for o in sorted(offs)]
    return sorted(cells)


def make_air(name, ctx, log_n, log_blowup=1, lde_offset=3):
    """-> prover.Air whose build_program emits the synthetic composition constraint.
    Tables are generated on the device once (periodic ones random, point-zerofier ones real)."""
    nbase, next_, ncons, periods, periodic_lens = LAYOUTS[name]
    mask = layout_mask(name)
    n, N = 1 << log_n, 1 << (log_n + log_blowup)
    mask = [(c, o) for c, o in mask if o < n]
    rng = random.Random(0xA12)
    # table layout: [zerofier inverses | periodic columns | point zerofier inverses]
    lens = [min(N, p << log_blowup) for p in periods] + [min(N, l << log_blowup) for l in periodic_lens]
    lens += [N] * N_POINT_ZEROFIERS
    desc, off = [], 0
    for ln in lens:
        desc += [off, ln.bit_length() - 1]
        off += ln
    total = off
    tables = ctx.alloc(32 * total)
    host = np.random.default_rng(7).integers(0, 2**63 - 1, size=(total - N * N_POINT_ZEROFIERS, 4), dtype=np.int64).astype(np.uint64)
    host[:, 3] &= np.uint64((1 << 59) - 1)
    tables.upload(host)
    g = be.felt(lde_offset)
    w_n = pow(3, (P - 1) >> log_n, P)
    for k in range(N_POINT_ZEROFIERS):
        view = be.DeviceView(tables, 32 * (total - N * (N_POINT_ZEROFIERS - k)), 32 * N)
        ctx.inverse_table(log_n + log_blowup, g, be.felt(pow(w_n, (k * 7919) % n, P)), view)
    n_zero, n_per = len(periods), len(periodic_lens)
    cells = [ap.Trace(c, o) for c, o in mask]

    def build_program(n_, challenges, comp_coeff):
        assert n_ == n
        alpha = canonical(comp_coeff)
        ch = [ap.Const(canonical(c)) for c in challenges]
        r = random.Random(0xC0FFEE)
        # composition = sum_k alpha^k C_k / Z_k, evaluated as sum_Z (1/Z) * (sum_{k: Z_k = Z} alpha^k C_k):
        # constraints sharing a zerofier are summed before the ONE multiplication by its inverse table
        groups, apow = {}, 1                 # zerofier table index -> (table node, partial sum); insertion-ordered
        for k in range(ncons):
            a, b, c = r.choice(cells), r.choice(cells), r.choice(cells)
            if k % 7 == 3:
                body = (a + ch[k % len(ch)]) * (b - ap.Table(n_zero + k % n_per)) - c       # periodic column
            elif k % 11 == 5:
                body = a * b - c * ch[k % len(ch)]                                          # challenge term
            else:
                body = a * b - c + ap.Const(r.randrange(P))
            if k % 9 == 8:
                zi = n_zero + n_per + (k // 9) % N_POINT_ZEROFIERS                          # boundary constraint
            else:
                zi = k % n_zero
            term = body * ap.Const(apow)
            if zi in groups:
                groups[zi] = (groups[zi][0], groups[zi][1] + term)
            else:
                groups[zi] = (ap.Table(zi), term)
            apow = apow * alpha % P
        total_expr = None
        for zer, partial in groups.values():
            term = partial * zer
            total_expr = term if total_expr is None else total_expr + term
        # make sure every mask cell is read (the real AIR reads each of its mask cells)
        used = set()
        return_prog = ap.lower(total_expr + sum_cells(cells, used), P)
        return return_prog, tables, desc

    def sum_cells(cs, _used):
        acc = cs[0]
        for c in cs[1:]:
            acc = acc + c
        return acc * ap.Table(0)

    air = Air("synthetic-" + name, nbase, next_, 6, mask, build_program)
    air.table_buffer = tables
    return air
