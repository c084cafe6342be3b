#!/bin/bash
# DEEP's rational form after the pruned transforms got cheaper: from how many cells on a column is worth a polynomial; the row-block form
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_call23; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_sharded_host.py -m gpu -x -q -k "deep or evaluate or lde or row_block or (sharded and not 2p16)" ) 2>&1 | tail -3
run() { name=$1; shift; timeout 300 python bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-north-star > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "
import json; d=json.load(open('$O/bench_$name.json')); s=d.get('stage_ms_per_proof') or d['stage_ms_per_proof_by_rank'][0]; print('$name', round(d['value'],4), 'ntt', s['ntt_pass'], 'deep', s['deep'])" || tail -5 $O/bench_$name.err; }
for mc in 24 12 6; do
  export SS_DEEP_RATIONAL_MIN_CELLS=$mc
  run sn20_$mc --workload starknet_2p20
  run rec20_$mc --workload recursive_2p20
done
unset SS_DEEP_RATIONAL_MIN_CELLS
run sn20_shard_py --workload starknet_2p20 --mode shard
run sn20_shard_cpp --workload starknet_2p20 --mode shard --sharded-host cpp
run rec20_shard_cpp --workload recursive_2p20 --mode shard --sharded-host cpp
echo done
