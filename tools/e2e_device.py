"""files -> proof at 2^20 steps with the base trace made ON the device, WITHOUT torch (a fresh box pays 1-2 minutes for `import torch`):
  gen        ssh_base_trace_device alone (upload of trace.bin / memory.bin, plans, kernels, the status read): wall time of the call
  prove      the proof alone on the resident columns (ssh_prove)
  total      ONE ssh_prove_files_device call from the files to the proof
and, with E2E_HOST=1, the host-generated path (ssh_prove_files: generator thread + overlapped uploads) beside it.
python tools/e2e_device.py [starknet recursive] [log_steps] ; one line per layout, flushed as it is known."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sandstorm_amd import backend as be, binary, examples, hostlib, public_input   # noqa: E402
from sandstorm_amd.prover import ProofOptions                                       # noqa: E402


def main(layouts, log_steps=20, repeats=5):
    log_n = log_steps + 4
    n = 1 << log_n
    ctx = be.Context(0)
    for layout in layouts:
        if layout == "starknet":
            from sandstorm_amd.layouts import starknet as sk
            states, memory, xpi = examples.starknet_example(log_steps)
            nb = 9
            aux_idx = (sk.COL_NPC, sk.COL_MEMORY, sk.COL_RANGE_CHECK)
            tree_kind, n_friendly, coin_kind = be.TREE_KECCAK_M20, 0, be.COIN_SOLIDITY
            air = hostlib.StarknetHostAir(ctx, xpi, log_n, 1)
        else:
            from sandstorm_amd.layouts import recursive as rec
            states, memory, xpi = examples.recursive_example(log_steps)
            nb = 7
            aux_idx = (rec.COL_NPC, rec.COL_MEMORY, rec.COL_RANGE_CHECK, rec.COL_DILUTED_UNORDERED, rec.COL_DILUTED_ORDERED)
            tree_kind, n_friendly, coin_kind = be.TREE_FRIENDLY, 22, be.COIN_CAIRO
            air = hostlib.RecursiveHostAir(ctx, xpi, log_n, 1)
        trace_bin, memory_bin = binary.write_register_states(states), binary.write_memory(memory)
        del states, memory
        seed = public_input.public_coin_seed(xpi, coin_kind)
        dev = [ctx.alloc(32 * n) for _ in range(nb)]
        keep = []

        def build_extension(challenges):
            for m in keep:
                m.close()
            del keep[:]
            keep.append(hostlib.build_extension_columns(ctx, layout, [dev[c] for c in aux_idx], n, challenges))
            return keep[0].cols
        options = ProofOptions()
        gen_s, prove_s, total_s, inner = [], [], [], []
        for it in range(repeats + 1):
            ctx.sync()
            t0 = time.perf_counter()
            hostlib.device_base_trace(ctx, layout, trace_bin, memory_bin, xpi, None, dev)
            ctx.sync()
            if it:
                gen_s.append(time.perf_counter() - t0)
        for it in range(repeats + 1):
            ctx.sync()
            t0 = time.perf_counter()
            hostlib.prove(ctx, air, tree_kind, n_friendly, coin_kind, seed, dev, log_n, build_extension, options, want_proof=False)
            ctx.sync()
            if it:
                prove_s.append(time.perf_counter() - t0)
        for it in range(repeats + 1):
            ctx.sync()
            t0 = time.perf_counter()
            _, tm = hostlib.prove_files_device(ctx, layout, trace_bin, memory_bin, xpi, None, dev, air, tree_kind, n_friendly, coin_kind, seed, build_extension,
                                               options, want_proof=False)
            ctx.sync()
            if it:
                total_s.append(time.perf_counter() - t0)
                inner.append(tm["trace_gen_s"])
        med = lambda v: sorted(v)[len(v) // 2]
        print("%s 2^%d steps, %.1f MB of files: device generator %s s; proof alone %s s; files -> proof (device generator) %s s = %.3f x the proof "
              "(the columns final %s s into the call)"
              % (layout, log_steps, (len(trace_bin) + len(memory_bin)) / 1e6, " ".join("%.4f" % v for v in gen_s), " ".join("%.4f" % v for v in prove_s),
                 " ".join("%.4f" % v for v in total_s), med(total_s) / med(prove_s), " ".join("%.4f" % v for v in inner)), flush=True)
        for m in keep:
            m.close()
        del keep[:]
        air.close()
        for d in dev:
            d.free()
        ctx.trim()


if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if a in ("starknet", "recursive")] or ["starknet", "recursive"]
    steps = [int(a) for a in sys.argv[1:] if a.isdigit()]
    main(names, steps[0] if steps else 20)
