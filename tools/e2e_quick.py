"""files -> proof at 2^20 steps WITHOUT torch (a fresh box pays 1-2 minutes for `import torch`; this fits a call of under a minute):
what bench.py's `end_to_end` leg measures - the generator alone into pinned columns, the proof alone on resident columns, and ONE
ssh_prove_files call from the files to the proof - with the pinned columns from hipHostMalloc through ctypes (bench.py takes them from
torch).  python tools/e2e_quick.py [starknet recursive] ; one line per layout, flushed as it is known."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sandstorm_amd import _lib, backend as be, binary, examples, hostlib, public_input   # noqa: E402
from sandstorm_amd.prover import ProofOptions                                             # noqa: E402


def hip_runtime():
    """the HIP runtime the C ABI's library brought into the process"""
    _lib.load()
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                return C.CDLL(line.split()[-1])
    raise RuntimeError("libamdhip64 is not mapped")


def main(layouts, log_steps=20, repeats=3):
    hip = hip_runtime()
    hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hip.hipHostFree.argtypes = [C.c_void_p]
    hip.hipDeviceSynchronize.argtypes = []
    log_n = log_steps + 4
    n = 1 << log_n
    ctx = be.Context(0)
    for layout in layouts:
        t_start = time.perf_counter()
        if layout == "starknet":
            from sandstorm_amd.layouts import starknet as sk
            states, memory, xpi = examples.starknet_example(log_steps)
            gen, nb = hostlib.starknet_base_trace, 9
            aux_idx = (sk.COL_NPC, sk.COL_MEMORY, sk.COL_RANGE_CHECK)
            tree_kind, n_friendly, coin_kind = be.TREE_KECCAK_M20, 0, be.COIN_SOLIDITY
            air = hostlib.StarknetHostAir(ctx, xpi, log_n, 1)
        else:
            from sandstorm_amd.layouts import recursive as rec
            states, memory, xpi = examples.recursive_example(log_steps)
            gen, nb = hostlib.recursive_base_trace, 7
            aux_idx = (rec.COL_NPC, rec.COL_MEMORY, rec.COL_RANGE_CHECK, rec.COL_DILUTED_UNORDERED, rec.COL_DILUTED_ORDERED)
            tree_kind, n_friendly, coin_kind = be.TREE_FRIENDLY, 22, be.COIN_CAIRO
            air = hostlib.RecursiveHostAir(ctx, xpi, log_n, 1)
        trace_bin, memory_bin = binary.write_register_states(states), binary.write_memory(memory)
        del states, memory
        seed = public_input.public_coin_seed(xpi, coin_kind)
        ptrs, views = [], []
        for _ in range(nb):
            p = C.c_void_p()
            if hip.hipHostMalloc(C.byref(p), 32 * n, 0) != 0:
                raise RuntimeError("hipHostMalloc failed")
            ptrs.append(p)
            views.append(np.ctypeslib.as_array((C.c_uint64 * (4 * n)).from_address(p.value)).reshape(n, 4))
        dev = [ctx.alloc(32 * n) for _ in range(nb)]
        keep = []

        def build_extension(challenges):
            for m in keep:
                m.close()
            del keep[:]
            keep.append(hostlib.build_extension_columns(ctx, layout, [dev[c] for c in aux_idx], n, challenges))
            return keep[0].cols
        options = ProofOptions()
        if os.environ.get("E2E_THREADS"):
            # the generator's OpenMP threads (SSH_HOST_THREADS, read per call) against the cgroup's CPU quota: the same call with each
            # setting in turn, three rounds, after one untimed call
            settings = os.environ["E2E_THREADS"].split(",")
            by = {t: [] for t in settings}
            for rnd in range(4):
                for t in settings:
                    os.environ["SSH_HOST_THREADS"] = t
                    hip.hipDeviceSynchronize()
                    t0 = time.perf_counter()
                    _, tm = hostlib.prove_files(ctx, layout, trace_bin, memory_bin, xpi, None, views, dev, air, tree_kind, n_friendly, coin_kind, seed,
                                                build_extension, options, want_proof=False)
                    hip.hipDeviceSynchronize()
                    if rnd:
                        by[t].append((time.perf_counter() - t0, tm["trace_gen_s"]))
                    if not rnd:
                        break
            del os.environ["SSH_HOST_THREADS"]
            for t in settings:
                print("%s 2^%d steps, %s generator threads: files -> proof %s s (generator thread %s)"
                      % (layout, log_steps, t, " ".join("%.4f" % a for a, _ in by[t]), " ".join("%.4f" % b for _, b in by[t])), flush=True)
            for m in keep:
                m.close()
            del keep[:]
            air.close()
            for d in dev:
                d.free()
            del dev, views
            ctx.trim()
            for p in ptrs:
                hip.hipHostFree(p)
            continue
        gen_s, prove_s, total_s, thread_s = [], [], [], []
        for it in range(repeats + 1):
            hip.hipDeviceSynchronize()
            t0 = time.perf_counter()
            _, tm = hostlib.prove_files(ctx, layout, trace_bin, memory_bin, xpi, None, views, dev, air, tree_kind, n_friendly, coin_kind, seed, build_extension,
                                        options, want_proof=False)
            hip.hipDeviceSynchronize()
            t1 = time.perf_counter()
            if it:
                total_s.append(t1 - t0)
                thread_s.append(tm["trace_gen_s"])
        for it in range(repeats):                       # (the columns of the last call are resident: the proof alone)
            t0 = time.perf_counter()
            hostlib.prove(ctx, air, tree_kind, n_friendly, coin_kind, seed, dev, log_n, build_extension, options, want_proof=False)
            hip.hipDeviceSynchronize()
            prove_s.append(time.perf_counter() - t0)
        for it in range(repeats):
            t0 = time.perf_counter()
            gen(trace_bin, memory_bin, xpi, out=views)
            gen_s.append(time.perf_counter() - t0)
        fmt = lambda v: " ".join("%.4f" % x for x in v)
        print("%s 2^%d steps: files -> proof %s s (generator thread %s) | proof alone %s s | generator alone %s s | ratio %.2f | set-up + all %.1f s"
              % (layout, log_steps, fmt(total_s), fmt(thread_s), fmt(prove_s), fmt(gen_s), min(total_s) / min(prove_s), time.perf_counter() - t_start), flush=True)
        for m in keep:
            m.close()
        air.close()
        for d in dev:
            d.free()
        del dev, views
        ctx.trim()
        for p in ptrs:
            hip.hipHostFree(p)
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or ["starknet", "recursive"]))
