"""GPU box: the timeline of ONE files -> proof call (hostlib.prove_files / ssh_prove_files) - when the host-to-device copies of the base
columns run against the transform kernels of the columns that have already landed.
  run:        cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/rp_e2e -- python $REPO/tools/e2e_overlap_trace.py run [layout] [log_steps]
  summarise:  python tools/e2e_overlap_trace.py summary /tmp/rp_e2e
The summary lists, for the LAST traced call, every base column's upload (start, end, GB/s) and the ntt_pass launches that ran before the
last upload ended - transforms of columns that arrived earlier, overlapping the uploads and the trace generation still going on."""
import csv
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(layout, log_steps):
    import torch
    from sandstorm_amd import backend as be, binary, examples, hostlib, public_input
    from sandstorm_amd.prover import ProofOptions
    n = 16 << log_steps
    log_n = log_steps + 4
    ctx = be.Context(0)
    if layout == "starknet":
        from sandstorm_amd.layouts import starknet as lay
        states, memory, pi = examples.starknet_example(log_steps)
        nb, aux_idx = 9, (lay.COL_NPC, lay.COL_MEMORY, lay.COL_RANGE_CHECK)
        tree, nf, coin = be.TREE_KECCAK_M20, 0, be.COIN_SOLIDITY
        air = hostlib.StarknetHostAir(ctx, pi, log_n, 1)
    else:
        from sandstorm_amd.layouts import recursive as lay
        states, memory, pi = examples.recursive_example(log_steps)
        nb, aux_idx = 7, (lay.COL_NPC, lay.COL_MEMORY, lay.COL_RANGE_CHECK, lay.COL_DILUTED_UNORDERED, lay.COL_DILUTED_ORDERED)
        tree, nf, coin = be.TREE_FRIENDLY, 22, be.COIN_CAIRO
        air = hostlib.RecursiveHostAir(ctx, pi, log_n, 1)
    tb, mb = binary.write_register_states(states), binary.write_memory(memory)
    del states, memory
    seed = public_input.public_coin_seed(pi, coin)
    pinned = [torch.empty((n, 4), dtype=torch.int64).pin_memory() for _ in range(nb)]
    views = [t.numpy().view("uint64") for t in pinned]
    dev = [torch.empty((n, 4), dtype=torch.int64, device="cuda") for _ in range(nb)]
    keep = []

    def ext(ch):
        del keep[:]
        keep.append(hostlib.build_extension_columns(ctx, layout, [dev[c] for c in aux_idx], n, ch))
        return keep[0].cols
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, tm = hostlib.prove_files(ctx, layout, tb, mb, pi, None, views, dev, air, tree, nf, coin, seed, ext, ProofOptions(), want_proof=False)
        torch.cuda.synchronize()
        print("call %d: %.4f s (generator thread %.4f s)" % (it, time.perf_counter() - t0, tm["trace_gen_s"]), flush=True)
        time.sleep(0.3)                                    # a visible pause between the calls in the trace


def summary(d):
    kt = glob.glob(os.path.join(d, "*", "*kernel_trace.csv"))
    mc = glob.glob(os.path.join(d, "*", "*memory_copy_trace.csv"))
    if not kt or not mc:
        sys.exit("no kernel_trace / memory_copy_trace csv under " + d)
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kt[0]))]
    cs = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Name", "")), r) for r in csv.DictReader(open(mc[0]))]
    big = sorted([c for c in cs if c[1] - c[0] > 3_000_000 and "HOST_TO_DEVICE" in str(c[2]).upper().replace("-", "_")])       # column uploads: > 3 ms each
    if not big:
        big = sorted([c for c in cs if c[1] - c[0] > 3_000_000])
    # the last call: the last run of big copies with less than 0.25 s between them
    last = [big[-1]]
    for c in reversed(big[:-1]):
        if last[0][0] - c[1] < 250_000_000:
            last.insert(0, c)
        else:
            break
    t0 = last[0][0]
    end_up = last[-1][1]
    print("files -> proof, the last traced call: %d column uploads, first starts at t = 0, last ends at t = %.1f ms" % (len(last), (end_up - t0) / 1e6))
    for c in last:
        size = None
        for key in ("Size", "Bytes", "size"):
            if key in c[3]:
                size = float(c[3][key])
        print("  upload  %8.1f .. %8.1f ms%s" % ((c[0] - t0) / 1e6, (c[1] - t0) / 1e6, "   %.1f GB/s" % (size / (c[1] - c[0])) if size else ""))
    ntt = [k for k in ks if "ntt_pass" in k[2] and t0 - 50_000_000 <= k[0] <= end_up]
    busy = sum(k[1] - k[0] for k in ntt)
    print("transform launches that STARTED before the last upload ended: %d, %.1f ms of kernel time between t = %.1f and %.1f ms"
          % (len(ntt), busy / 1e6, (ntt[0][0] - t0) / 1e6 if ntt else 0.0, (ntt[-1][1] - t0) / 1e6 if ntt else 0.0))
    over = 0
    for k in ntt:
        for c in last:
            lo, hi = max(k[0], c[0]), min(k[1], c[1])
            if hi > lo:
                over += hi - lo
    print("of it concurrent with an upload in flight: %.1f ms" % (over / 1e6))
    after = [k for k in ks if k[0] > end_up and k[0] - end_up < 400_000_000]
    if after:
        print("kernels after the last upload (the rest of the proof): %.1f ms of wall time" % ((after[-1][1] - end_up) / 1e6))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2] if len(sys.argv) > 2 else "starknet", int(sys.argv[3]) if len(sys.argv) > 3 else 20)
    else:
        summary(sys.argv[2])
