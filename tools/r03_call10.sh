#!/bin/bash
# host-side gaps: the coin's Pedersen chain in 64-bit limbs, then where the remaining idle time of a proof is
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_call10; mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 python bench.py "$@" --steps 5 --warmup 1 --no-cpu-baseline --no-north-star > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "
import json; d=json.load(open('$O/bench_$name.json')); print('$name', round(d['value'],4), d.get('stage_ms_per_proof',''))" || tail -5 $O/bench_$name.err; }
run example --workload array_sum_example
run rec16 --workload recursive_2p16
run rec20 --workload recursive_2p20
run sn20 --workload starknet_2p20
for w in array_sum_example recursive_2p20 starknet_2p20; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_$w -o kt -- python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-north-star > $O/kt_$w.json 2> $O/kt_$w.err
  f=$(find $O/kt_$w -name 'kt_kernel_trace.csv' | head -1)
  echo "== $w"; python tools/trace_gaps.py $f 1 --anchor=pow_ > $O/gaps_$w.txt 2>&1; head -32 $O/gaps_$w.txt
  rm -rf $O/kt_$w
done
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_recursive_claim.py -m gpu -x -q -k "ood or pedersen or coin or 2p14_steps or 2p16_steps_cairo" ) 2>&1 | tail -3
echo done
