"""tests/golden/lde_offset_pin.json: the values of base-trace columns 1 and 2 (diluted-check unordered / ordered) that the
reference's shipped recursive-layout proof (/root/reference/bootloader-proof.bin, 2^18 trace rows) opens at its 40 query
positions.  For a run that uses no bitwise instance those two columns do not depend on the program at all, so they can
be regenerated here (sandstorm_amd/layouts/recursive.py) and extended: they coincide with the proof's values at every
position exactly when the low-degree extension is taken over the coset 3 * <w_N> and position q holds the point
3 * w_N^bitrev(q).  That pins, from data alone, the LDE coset offset, once more the committed
order (M3), the polynomial convention of Matrix::interpolate / evaluate on real prover output, and this repo's
generation of the diluted-check columns.  Run in the build container (reads the reference's file; data only)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sandstorm_amd import wire  # noqa: E402

with open("/root/reference/bootloader-proof.bin", "rb") as f:
    w = wire.parse(f.read())
with open(os.path.join(ROOT, "tests", "golden", "saved_proof_openings_recursive.json")) as f:
    positions = json.load(f)["positions"]
nq = len(positions)
ncols = len(w.base_rows) // nq
assert ncols == 7 and len(w.base_openings) == nq
out = {"file": "bootloader-proof.bin", "trace_len": w.trace_len, "lde_blowup": w.options[1], "positions": positions,
       "column1": [hex(w.base_rows[ncols * q + 1]) for q in range(nq)],
       "column2": [hex(w.base_rows[ncols * q + 2]) for q in range(nq)]}
with open(os.path.join(ROOT, "tests", "golden", "lde_offset_pin.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote", nq, "positions")
