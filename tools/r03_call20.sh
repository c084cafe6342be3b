#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_call20; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_recursive_claim.py -m gpu -x -q -k "sharded" ) 2>&1 | tail -3
timeout 600 python bench.py --workload starknet_2p20 --mode shard --steps 3 --warmup 1 --no-cpu-baseline --no-north-star > $O/bench_sn.json 2> $O/bench_sn.err
python -c "
import json; d=json.load(open('$O/bench_sn.json')); print('shard python', round(d['value'],4))"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python bench.py --workload starknet_2p20 --mode shard --steps 2 --warmup 0 --no-cpu-baseline --no-north-star > $O/kt.json 2> $O/kt.err
f=$(find $O/kt -name 'kt_kernel_stats.csv' | head -1); python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows:
    n=r["Name"]
    if "ss::" not in n[:40]:
        print(n[:110], r["Calls"], round(float(r["TotalDurationNs"])/1e6,3), "ms")
PY
rm -rf $O/kt
echo done
