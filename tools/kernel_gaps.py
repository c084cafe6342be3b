"""Where the device sat idle inside a run: the gaps between consecutive kernels of a rocprofv3 kernel trace.
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/rp -- python $REPO/bench.py --workload goldilocks_plain_2p20 --steps 2 --warmup 1
  python tools/kernel_gaps.py /tmp/rp [min_gap_us=150] [last_fraction=0.5]
Kernels of every stream are merged into one timeline (a gap = no kernel of the process running); only the last `last_fraction` of the trace
is reported - the timed proofs, not the warm-up.  For each gap: its length, the kernel that ended before it and the one that started after -
which names the host work in between (a digest download, a coin draw, the lowering of a composition, an allocation)."""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
    frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
    rows = []
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    cut = t1 - (t1 - t0) * frac
    rows = [r for r in rows if r[0] >= cut]
    busy_end, prev, gaps, busy = rows[0][1], rows[0][2], [], 0
    seg_start = rows[0][0]
    for s, e, name in rows[1:]:
        if s > busy_end:
            busy += busy_end - seg_start
            gaps.append(((s - busy_end) / 1e3, prev, name, (busy_end - cut) / 1e6))
            seg_start = s
        if e > busy_end:
            busy_end, prev = e, name
    busy += busy_end - seg_start
    span = rows[-1][1] - rows[0][0]
    print("window %.2f ms: device busy %.2f ms, idle %.2f ms in %d gaps" % (span / 1e6, busy / 1e6, (span - busy) / 1e6, len(gaps)))
    small = sum(g[0] for g in gaps if g[0] < min_gap)
    print("gaps under %.0f us: %d, %.2f ms together" % (min_gap, sum(1 for g in gaps if g[0] < min_gap), small / 1e3))
    by_name = {}
    for s_, e_, name in rows:
        t = by_name.setdefault(name, [0, 0])
        t[0] += e_ - s_
        t[1] += 1
    print("kernel time in the window by name (ms, launches):")
    for name, (t, k) in sorted(by_name.items(), key=lambda kv: -kv[1][0])[:16]:
        print("  %9.3f %6d  %s" % (t / 1e6, k, name[:90]))
    if os.environ.get("GAPS_SEQUENCE"):
        # the launches of the window in order, runs of one kernel folded: when, how many, their time, the idle time before each run's launches
        print("sequence (start ms, launches, kernel ms, idle us before / inside the run, kernel):")
        run, last_end = None, rows[0][0]
        for s_, e_, name in rows:
            gap = max(0, s_ - last_end) / 1e3
            if run and run[0] == name:
                run[2] += 1; run[3] += e_ - s_; run[5] += gap
            else:
                if run:
                    print("  %9.2f %5d %9.3f %8.1f %8.1f  %s" % ((run[1] - cut) / 1e6, run[2], run[3] / 1e6, run[4], run[5], run[0][:70]))
                run = [name, s_, 1, e_ - s_, gap, 0.0]
            last_end = max(last_end, e_)
        print("  %9.2f %5d %9.3f %8.1f %8.1f  %s" % ((run[1] - cut) / 1e6, run[2], run[3] / 1e6, run[4], run[5], run[0][:70]))
    pairs = {}
    for g in gaps:
        t = pairs.setdefault((g[1][:40], g[2][:40]), [0.0, 0])
        t[0] += g[0]
        t[1] += 1
    print("idle time by (kernel before, kernel after) (ms, gaps):")
    for (a, b), (t, k) in sorted(pairs.items(), key=lambda kv: -kv[1][0])[:14]:
        print("  %9.3f %6d  %-40s -> %s" % (t / 1e3, k, a, b))
    for g in gaps:
        if g[0] >= min_gap:
            print("  at %8.2f ms  %9.1f us   after %-46s before %s" % (g[3], g[0], g[1][:46], g[2][:60]))


if __name__ == "__main__":
    main()
