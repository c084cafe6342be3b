#!/usr/bin/env python3
"""Per-instruction cost of the constraint VM (ss_eval_quotient): runs programs made of
1000 copies of one instruction flavour over 2^22 points and prints ns per (point x instr)
and the implied wave-instruction cycle cost.  GPU only."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sandstorm_amd import backend as be            # noqa: E402
from sandstorm_amd import air_program as ap        # noqa: E402


def main():
    ctx = be.Context(0)
    log_n, lb = 21, 1
    N = 1 << (log_n + lb)
    rng = np.random.default_rng(1)
    cols = []
    for c in range(4):
        h = rng.integers(0, 2**63 - 1, size=(N, 4), dtype=np.int64).astype(np.uint64)
        h[:, 3] &= np.uint64((1 << 59) - 1)
        cols.append(ctx.column(h))
    tab = ctx.column(rng.integers(0, 2**59, size=(4096, 4), dtype=np.int64).astype(np.uint64))
    out = ctx.alloc(32 * N)
    g = be.felt(3)
    K = 1000

    def run(name, body):
        prog = ap.Program()
        prog.consts = [12345, 67890]
        prog.n_slots = 4
        code = ap.instr(ap.OP.MOV, 0, ap.SRC.TRACE, ap.trace_payload(0, 0))
        code += ap.instr(ap.OP.MOV, 1, ap.SRC.TRACE, ap.trace_payload(1, 0))
        code += ap.instr(ap.OP.ST, 1, 0, 0)
        for k in range(K):
            code += body(k)
        code += ap.instr(ap.OP.OUT, 0, 0, 0)
        prog.code = code
        for rep in range(2):
            ctx.sync()
            t0 = time.perf_counter()
            ctx.eval_quotient(prog, tab, [0, 12], cols, log_n, lb, g, out)
            ctx.sync()
            dt = time.perf_counter() - t0
        per = dt / (N * K)
        # 1024 SIMDs x 64 lanes; cycles per wave-instruction on one SIMD at 2.1 GHz
        cyc = per * 1024 * 64 * 2.1e9
        print("%-34s %8.2f ms   %6.3f ns/(point*instr)   ~%5.0f cyc/wave-instr" % (name, dt * 1e3, per * 1e9, cyc))

    I = ap.instr
    run("MUL acc0, TRACE", lambda k: I(ap.OP.MUL, 0, ap.SRC.TRACE, ap.trace_payload(k % 4, k % 3)))
    run("MUL acc0, CONST", lambda k: I(ap.OP.MUL, 0, ap.SRC.CONST, k % 2))
    run("MUL acc0, acc1", lambda k: I(ap.OP.MUL, 0, ap.SRC.ACC, 1))
    run("MUL acc0, TABLE", lambda k: I(ap.OP.MUL, 0, ap.SRC.TABLE, 0))
    run("MUL acc0, SLOT", lambda k: I(ap.OP.MUL, 0, ap.SRC.SLOT, 0))
    run("ADD acc0, TRACE", lambda k: I(ap.OP.ADD, 0, ap.SRC.TRACE, ap.trace_payload(k % 4, k % 3)))
    run("ADD acc0, acc1", lambda k: I(ap.OP.ADD, 0, ap.SRC.ACC, 1))
    run("SUB acc0, CONST", lambda k: I(ap.OP.SUB, 0, ap.SRC.CONST, k % 2))
    run("MOV acc2, TRACE", lambda k: I(ap.OP.MOV, 2, ap.SRC.TRACE, ap.trace_payload(k % 4, k % 3)))
    run("ST acc0 -> slot", lambda k: I(ap.OP.ST, 0, 0, 1 + k % 3))
    run("MUL acc(k%4), TRACE (4 accs)", lambda k: I(ap.OP.MUL, k % 4, ap.SRC.TRACE, ap.trace_payload(k % 4, 0)))
    ctx.close()


if __name__ == "__main__":
    main()
