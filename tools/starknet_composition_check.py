"""One-off CPU check of the whole starknet chain with the oracle (about 5 minutes; the unit tests check the pieces):
the 10 columns of the 2^21-row trace of the reference's bootloader run (tests/test_layout_starknet.py, with the extra builtin instances) are extended to the 2^22-point coset, the
lowered composition program runs over it, and the result interpolates to a polynomial of degree exactly 2n - 3 - every
one of the 195 constraints is divisible by its zerofier - which satisfies the verifier's out-of-domain identity at a
random point.  Usage: python tools/starknet_composition_check.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle_py as oracle  # noqa: E402
from sandstorm_amd import air_program as ap  # noqa: E402
from sandstorm_amd.layouts import starknet as sk  # noqa: E402
from test_layout_starknet import CHALLENGES, P, bootloader_run, real_instances  # noqa: E402

t0 = time.time()
states, memory, spi, private = bootloader_run()
extra = real_instances()
private["pedersen"] += extra.pop("pedersen")
private.update(extra)
cols = sk.base_trace(states, memory, spi, private)
n = len(cols[0])
log_n, N = n.bit_length() - 1, 2 * n
mont = [oracle.to_mont(c) for c in cols]
ext, _ = oracle.build_extension_columns("starknet", {"npc": mont[sk.COL_NPC], "memory": mont[sk.COL_MEMORY], "range_check": mont[sk.COL_RANGE_CHECK]},
                                        [oracle.to_mont([c])[0] for c in CHALLENGES], n)
mont.append(ext[0])
print("trace and extension column: %.0f s" % (time.time() - t0), flush=True)
alpha = pow(5, 77, P)
hints = sk.Hints.from_public_input(spi, CHALLENGES, n)
tables = sk.Tables(n)
prog = ap.lower(sk.composition(n, hints, CHALLENGES, alpha, tables), P)
vals, desc, off = [], [], 0
for spec in tables.specs:
    v = tables.host_values(spec)
    desc += [off, len(v).bit_length() - 1]
    off += len(v)
    vals += v
print("tables (%d felts): %.0f s" % (off, time.time() - t0), flush=True)
g = oracle.to_mont([3])[0]
lde = [oracle.lde(c, 1, g)[0] for c in mont]
print("LDE: %.0f s" % (time.time() - t0), flush=True)
out = oracle.eval_program(prog.code, oracle.to_mont(prog.consts), oracle.to_mont(vals), desc, prog.n_slots, lde, log_n, 1, g)
coeffs = oracle.ntt(out, inverse=True, offset=g)
top = int(np.nonzero(coeffs.any(axis=1))[0][-1])
print("composition: degree %d (2n - 3 = %d): %.0f s" % (top, 2 * n - 3, time.time() - t0), flush=True)
assert top == 2 * n - 3
va = sk.verifier_air(spi)
z = pow(11, 1234567, P)
wn = pow(3, (P - 1) // n, P)
trace_coeffs = [oracle.ntt(c, inverse=True) for c in mont]
ood = {(c, o): int(oracle.from_mont(oracle.poly_eval(trace_coeffs[c], oracle.to_mont([z * pow(wn, o, P) % P])[0])[None])[0]) for c, o in va.mask}
lhs = ap.evaluate(va.composition(n, CHALLENGES, alpha), P, z, lambda c, o: ood[(c, o)], lambda t: va.table_at(n, z, t))
h0, h1 = np.ascontiguousarray(coeffs[0::2]), np.ascontiguousarray(coeffs[1::2])
z2 = oracle.to_mont([z * z % P])[0]
rhs = (int(oracle.from_mont(oracle.poly_eval(h0, z2)[None])[0]) + z * int(oracle.from_mont(oracle.poly_eval(h1, z2)[None])[0])) % P
print("out-of-domain identity: %s: %.0f s" % (lhs == rhs, time.time() - t0), flush=True)
assert lhs == rhs
print("ok")
