#!/bin/bash
# Pedersen small levels: Jacobian additions over quads (DPP) and the inversion over a quad
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_call14; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_prove.py tests/test_gpu_recursive_claim.py -m gpu -x -q -k "pedersen or merkle or friendly or cairo or 2p14_steps or 2p16_steps_cairo" ) 2>&1 | tail -3
run() { name=$1; shift; timeout 300 python bench.py "$@" --steps 5 --warmup 1 --no-cpu-baseline --no-north-star > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "
import json; d=json.load(open('$O/bench_$name.json')); print('$name', round(d['value'],4), d.get('stage_ms_per_proof',''))" || tail -5 $O/bench_$name.err; }
run example --workload array_sum_example
run rec16 --workload recursive_2p16
run rec20 --workload recursive_2p20
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python bench.py --workload array_sum_example --steps 3 --warmup 1 --no-cpu-baseline --no-north-star > $O/kt.json 2> $O/kt.err
f=$(find $O/kt -name 'kt_kernel_stats.csv' | head -1); python - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:8]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
rm -rf $O/kt
echo done
