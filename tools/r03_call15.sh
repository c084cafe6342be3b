#!/bin/bash
# Pedersen: up to how many hashes per level the 32-lanes-per-hash kernel wins
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_call15; mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 python bench.py "$@" --steps 5 --warmup 1 --no-cpu-baseline --no-north-star > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "
import json; d=json.load(open('$O/bench_$name.json')); print('$name', round(d['value'],4), d['stage_ms_per_proof']['merkle'])" || tail -5 $O/bench_$name.err; }
for mx in 2048 4096 8192 16384 32768; do
  export SS_PED_SMALL_MAX=$mx
  run example_$mx --workload array_sum_example
  run rec16_$mx --workload recursive_2p16
  run rec20_$mx --workload recursive_2p20
done
echo done
