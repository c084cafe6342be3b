#!/bin/bash
# from how many coefficients on the point-by-point out-of-domain path wins, now that its host share runs in 64-bit limbs
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_call16; mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 python bench.py "$@" --steps 5 --warmup 1 --no-cpu-baseline --no-north-star > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "
import json; d=json.load(open('$O/bench_$name.json')); s=d['stage_ms_per_proof']; print('$name', round(d['value'],4), 'ntt', s['ntt_pass'], 'deep', s['deep'])" || tail -5 $O/bench_$name.err; }
for mn in 22 20 18 16; do
  export SS_OOD_SPARSE_MIN_LOG=$mn
  run example_$mn --workload array_sum_example
  run rec16_$mn --workload recursive_2p16
done
echo done
