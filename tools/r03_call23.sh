#!/bin/bash
# DEEP's rational form after the pruned transforms got cheaper: from how many cells on a column is worth a polynomial
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_call23; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "deep or evaluate or lde or ntt" ) 2>&1 | tail -3
run() { name=$1; shift; timeout 300 python bench.py "$@" --steps 4 --warmup 1 --no-cpu-baseline --no-north-star > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "
import json; d=json.load(open('$O/bench_$name.json')); s=d['stage_ms_per_proof']; print('$name', round(d['value'],4), 'ntt', s['ntt_pass'], 'deep', s['deep'])" || tail -5 $O/bench_$name.err; }
for mc in 24 12 8 4; do
  export SS_DEEP_RATIONAL_MIN_CELLS=$mc
  run sn20_$mc --workload starknet_2p20
  run rec20_$mc --workload recursive_2p20
done
unset SS_DEEP_RATIONAL_MIN_CELLS
run rec16 --workload recursive_2p16
SS_DEEP_TAPS=1 run rec16_taps --workload recursive_2p16
echo done
