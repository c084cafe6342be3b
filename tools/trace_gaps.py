"""Where a proof's wall time is NOT kernels: the idle gaps of a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o kt -- python bench.py --workload W --steps K --no-cpu-baseline --no-north-star
    python tools/trace_gaps.py DIR/kt_kernel_trace.csv [proofs]

The last `proofs` (default 1) repetitions of the trace's periodic part are taken as the sample: the trace is cut at the last
`proofs` periods of an anchor kernel (--anchor=NAME; default: the kernel of the trace's last quarter that is launched least
often but at least `proofs` times).  Printed: busy time, idle time, and the largest gaps
with the kernels on either side - a gap is host work (the coin, openings, descriptor set-up, ctypes) or a synchronisation."""
import csv
import sys
from collections import Counter


def main():
    path = sys.argv[1]
    proofs = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else 1
    anchor = None
    for a in sys.argv[2:]:
        if a.startswith("--anchor="):
            anchor = a.split("=", 1)[1]
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    if anchor is None:
        tail = rows[len(rows) * 3 // 4:]
        cnt = Counter(n for _, _, n in tail)
        ok = [n for n, c in cnt.items() if c >= proofs]
        anchor = min(ok, key=lambda n: cnt[n])
    hits = [i for i, r in enumerate(rows) if anchor in r[2]]
    # one proof = from one launch of the anchor to the next launch of it that is a whole period later: take the distance between
    # the last two groups of launches separated by more than half the longest distance between consecutive hits
    d = [hits[i + 1] - hits[i] for i in range(len(hits) - 1)]
    if not d:
        lo, hi = 0, len(rows)
    else:
        big = max(d)
        starts = [hits[0]] + [hits[i + 1] for i in range(len(d)) if d[i] > big // 2]
        if len(starts) < proofs + 1:
            lo, hi = starts[0], len(rows)
        else:
            lo, hi = starts[-proofs - 1], starts[-1]
    sample = rows[lo:hi]
    t0, t1 = sample[0][0], sample[-1][1]
    busy, end, gaps = 0, sample[0][0], []
    for k, (s, e, n) in enumerate(sample):
        if s > end:
            gaps.append((s - end, sample[k - 1][2] if k else "-", n))
            busy += e - s
        else:
            busy += max(0, e - max(s, end))
        end = max(end, e)
    wall = t1 - t0
    print("anchor: %s" % anchor[:100])
    print("sample: %d launches, %d proof(s): wall %.3f ms, kernels busy %.3f ms, idle %.3f ms (%.1f %%) per proof"
          % (len(sample), proofs, wall / 1e6 / proofs, busy / 1e6 / proofs, (wall - busy) / 1e6 / proofs, 100.0 * (wall - busy) / wall))
    small = sum(g for g, _, _ in gaps if g < 20000)
    print("gaps: %d, of them < 20 us: %d totalling %.3f ms per proof (launch-to-launch)" % (len(gaps), sum(1 for g in gaps if g[0] < 20000), small / 1e6 / proofs))
    short = lambda n: n.split("(")[0].replace("void ", "").replace("ss::(anonymous namespace)::", "").replace("ss::", "")[:48]
    print("largest gaps:")
    for g, a, b in sorted(gaps, reverse=True)[:25]:
        print("  %8.3f ms   after %-48s before %s" % (g / 1e6, short(a), short(b)))
    # by (after, before) pair
    pair = Counter()
    for g, a, b in gaps:
        pair[(short(a), short(b))] += g
    print("by neighbouring kernels (total per proof):")
    for (a, b), g in pair.most_common(15):
        print("  %8.3f ms   after %-48s before %s" % (g / 1e6 / proofs, a, b))


if __name__ == "__main__":
    main()
