#!/bin/bash
# what the Python sharded driver runs on the GPU besides the library's kernels (one rank)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_call19; mkdir -p $O
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python bench.py --workload starknet_2p20 --mode shard --steps 2 --warmup 0 --no-cpu-baseline --no-north-star > $O/kt.json 2> $O/kt.err
f=$(find $O/kt -name 'kt_kernel_stats.csv' | head -1); python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=0
for r in rows:
    n=r["Name"]
    if not n.startswith("ss::") and "ss::" not in n[:12]:
        print(n[:110], r["Calls"], round(float(r["TotalDurationNs"])/1e6,3), "ms")
PY
f=$(find $O/kt -name 'kt_kernel_trace.csv' | head -1); python tools/trace_gaps.py $f 1 --anchor=pow_ > $O/gaps_shard.txt 2>&1; head -30 $O/gaps_shard.txt
rm -rf $O/kt
echo done
