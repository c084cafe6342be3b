#!/bin/bash
# Pedersen small levels: one launch per level (pedersen_pairs_small_kernel) against the split kernel + finish
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_call11; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_prove.py tests/test_gpu_recursive_claim.py -m gpu -x -q -k "pedersen or merkle or friendly or cairo or 2p14_steps or 2p16_steps_cairo" ) 2>&1 | tail -3
run() { name=$1; shift; timeout 300 python bench.py "$@" --steps 5 --warmup 1 --no-cpu-baseline --no-north-star > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "
import json; d=json.load(open('$O/bench_$name.json')); print('$name', round(d['value'],4), d.get('stage_ms_per_proof',''))" || tail -5 $O/bench_$name.err; }
run example --workload array_sum_example
run rec16 --workload recursive_2p16
run rec20 --workload recursive_2p20
export SS_PED_SMALL_TWO_LAUNCHES=1
run example_old --workload array_sum_example
run rec16_old --workload recursive_2p16
run rec20_old --workload recursive_2p20
unset SS_PED_SMALL_TWO_LAUNCHES
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python bench.py --workload array_sum_example --steps 3 --warmup 1 --no-cpu-baseline --no-north-star > $O/kt.json 2> $O/kt.err
f=$(find $O/kt -name 'kt_kernel_stats.csv' | head -1); head -14 $f | cut -c1-150
f=$(find $O/kt -name 'kt_kernel_trace.csv' | head -1); python tools/trace_gaps.py $f 1 --anchor=pow_ > $O/gaps_example.txt 2>&1; head -8 $O/gaps_example.txt
rm -rf $O/kt
echo done
