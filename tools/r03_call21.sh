#!/bin/bash
# DEEP: the mask's large columns and the constants as rational functions (A_c / B by pruned transforms) against a tap per cell
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_call21; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_prove.py -m gpu -x -q -k "deep or prove or proof" ) 2>&1 | tail -3
run() { name=$1; shift; timeout 300 python bench.py "$@" --steps 4 --warmup 1 --no-cpu-baseline --no-north-star > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "
import json; d=json.load(open('$O/bench_$name.json')); s=d['stage_ms_per_proof']; print('$name', round(d['value'],4), 'ntt', s['ntt_pass'], 'deep', s['deep'])" || tail -5 $O/bench_$name.err; }
run sn20 --workload starknet_2p20
run rec20 --workload recursive_2p20
export SS_DEEP_TAPS=1
run sn20_taps --workload starknet_2p20
run rec20_taps --workload recursive_2p20
unset SS_DEEP_TAPS
export SS_DEEP_RATIONAL_MIN_LOG=16
run rec16_rat --workload recursive_2p16
unset SS_DEEP_RATIONAL_MIN_LOG
run rec16 --workload recursive_2p16
echo done
