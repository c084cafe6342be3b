"""The 100 base-trace rows the reference's own starknet proof opens (data; run in the build container where
/root/reference is mounted).

`example/bootloader/bootloader-proof.bin` is a proof of the run shipped beside it (trace.bin, memory.bin, the public and
private inputs), made by an earlier revision of the reference: its digests and openings are serialised differently from
the current wire format, but the opened trace rows are plain 32-byte little-endian field elements - a vector of 900
(100 queries x 9 base columns) at byte 344574.  The query positions are not in the proof (the verifier re-derives them
from the transcript): this script regenerates the base trace with sandstorm_amd/layouts/starknet.py, extends it with the
oracle, finds each row's position by its first column and keeps it only if the other eight columns agree there."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle_py as oracle  # noqa: E402
from sandstorm_amd.layouts import starknet as sk  # noqa: E402
from test_layout_starknet import bootloader_run  # noqa: E402

raw = open("/root/reference/example/bootloader/bootloader-proof.bin", "rb").read()
OFFSET = 344574
assert int.from_bytes(raw[OFFSET:OFFSET + 8], "little") == 900
rows = [[int.from_bytes(raw[OFFSET + 8 + 288 * q + 32 * c: OFFSET + 8 + 288 * q + 32 * c + 32], "little") for c in range(9)] for q in range(100)]
states, memory, pi, private = bootloader_run()
cols = sk.base_trace(states, memory, pi, private)
g = oracle.to_mont([3])[0]
lde = [oracle.from_mont(oracle.lde(oracle.to_mont(c), 1, g)[0]) for c in cols]
index = {}
for i, v in enumerate(lde[0]):
    index.setdefault(int(v), []).append(i)
positions = []
for row in rows:
    match = [p for p in index.get(row[0], []) if all(int(lde[c][p]) == row[c] for c in range(1, 9))]
    assert len(match) == 1, "an opened row is not a row of the regenerated trace"
    positions.append(match[0])
out = {"source": "example/bootloader/bootloader-proof.bin, bytes %d.." % OFFSET, "trace_len": len(cols[0]), "lde_blowup": 2, "lde_offset": 3,
       "note": "positions index the natural-order evaluation domain 3 * w^i",
       "positions": positions, "rows": [[hex(v) for v in row] for row in rows]}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "starknet_opened_rows.json"), "w") as f:
    json.dump(out, f)
print(len(positions), "rows")
