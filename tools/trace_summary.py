"""Summarise a rocprofv3 kernel_trace.csv: per kernel name -> calls, total/avg/min/max duration (ms)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    agg[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
tot = sum(sum(v) for v in agg.values())
print("%-86s %7s %11s %9s %9s %9s %6s" % ("kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "%"))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-86s %7d %11.3f %9.4f %9.4f %9.4f %6.2f" % (k[:86], len(v), sum(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
print("total kernel time %.3f ms over %d dispatches" % (tot, len(rows)))
