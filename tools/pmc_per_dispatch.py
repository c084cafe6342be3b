"""Per-dispatch counters of one kernel from a rocprofv3 counter_collection.csv, in dispatch order."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2]
per = collections.OrderedDict()
for r in rows:
    if pat in r["Kernel_Name"]:
        per.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for d, cs in sorted(per.items()):
    print(d, " ".join("%s=%.4g" % kv for kv in sorted(cs.items())))
