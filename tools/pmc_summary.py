"""Summarise a rocprofv3 counter_collection.csv: per kernel name, mean of each counter per dispatch."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    n = max(len(v) for v in cs.values())
    print("%-60s disp=%d" % (k, n))
    for c, v in sorted(cs.items()):
        print("    %-24s mean %.4g  sum %.4g" % (c, sum(v) / len(v), sum(v)))
