#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_call22; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_extension.py tests/test_gpu_real_air.py tests/test_gpu_reference_proof.py -m gpu -x -q ) 2>&1 | tail -3
run() { name=$1; shift; timeout 300 python bench.py "$@" --steps 4 --warmup 1 --no-cpu-baseline --no-north-star > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "
import json; d=json.load(open('$O/bench_$name.json')); s=d['stage_ms_per_proof']; print('$name', round(d['value'],4), s)" || tail -5 $O/bench_$name.err; }
run sn20 --workload starknet_2p20
run rec20 --workload recursive_2p20
run rec16 --workload recursive_2p16
run example --workload array_sum_example
echo done
