"""One short GPU call without torch (a fresh box pays 1-2 minutes for `import torch`): trace.bin / memory.bin of the reference's example
padded to 2^14 steps -> ssh_prove_files (generator thread, asynchronous uploads, prover) -> the committed fixture
tests/golden/array_sum_recursive_cairo.proof byte for byte; then the same at 2^16 steps, twice (the same bytes both times) with the call's
own clock.  What the last GPU seconds of a round are spent on after a host-only change."""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sandstorm_amd import backend as be, binary, examples, hostlib, public_input   # noqa: E402
from sandstorm_amd.layouts import recursive as rec                                  # noqa: E402


def prove(log_steps, repeats):
    states, memory, pi = examples.recursive_example(log_steps)
    trace_bin, memory_bin = binary.write_register_states(states), binary.write_memory(memory)
    n = 16 << log_steps
    ctx = be.Context(0)
    views = [np.zeros((n, 4), dtype=np.uint64) for _ in range(7)]
    dev = [ctx.alloc(32 * n) for _ in range(7)]
    air = hostlib.RecursiveHostAir(ctx, pi, log_steps + 4)
    seed = public_input.public_coin_seed(pi, be.COIN_CAIRO)
    aux_idx = (rec.COL_NPC, rec.COL_MEMORY, rec.COL_RANGE_CHECK, rec.COL_DILUTED_UNORDERED, rec.COL_DILUTED_ORDERED)
    keep, out = [], []

    def build_extension(challenges):
        keep.append(hostlib.build_extension_columns(ctx, "recursive", [dev[c] for c in aux_idx], n, challenges))
        return keep[-1].cols
    for _ in range(repeats):
        t0 = time.perf_counter()
        raw, times = hostlib.prove_files(ctx, "recursive", trace_bin, memory_bin, pi, None, views, dev, air, be.TREE_FRIENDLY, 22, be.COIN_CAIRO, seed,
                                         build_extension)
        out.append((raw, time.perf_counter() - t0, times))
    for m in keep:
        m.close()
    air.close()
    ctx.close()
    return out


def main():
    t0 = time.perf_counter()
    with open(os.path.join(ROOT, "tests", "golden", "array_sum_recursive_cairo.proof"), "rb") as f:
        want = f.read()
    (raw, sec, times), = prove(14, 1)
    print("2^14 steps: %d bytes, sha256 %s, fixture %s, %.3f s (first call of the process)" % (len(raw), hashlib.sha256(raw).hexdigest()[:16],
                                                                                             "EQUAL" if raw == want else "DIFFERENT", sec), flush=True)
    runs = prove(16, 3)
    same = all(r[0] == runs[0][0] for r in runs)
    print("2^16 steps: %d bytes, sha256 %s, three calls %s: %s s; generator thread %s s" % (
        len(runs[0][0]), hashlib.sha256(runs[0][0]).hexdigest()[:16], "EQUAL" if same else "DIFFERENT",
        " ".join("%.4f" % r[1] for r in runs), " ".join("%.4f" % r[2]["trace_gen_s"] for r in runs)), flush=True)
    print("whole script %.1f s" % (time.perf_counter() - t0), flush=True)
    return 0 if raw == want and same else 1


if __name__ == "__main__":
    sys.exit(main())
