#!/bin/bash
# the Python sharded driver after the zero fills went: tests, one-rank line, and its host profile over the timed proofs
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_call18; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q ) 2>&1 | tail -3
SS_BENCH_PROFILE=1 timeout 600 python bench.py --workload starknet_2p20 --mode shard --steps 3 --warmup 1 --no-cpu-baseline --no-north-star > $O/bench_sn.json 2> $O/bench_sn.err
python -c "
import json; d=json.load(open('$O/bench_sn.json')); print('shard python', round(d['value'],4), d['stage_ms_per_proof_by_rank'])"
grep -A60 "Ordered by" $O/bench_sn.err | head -64
timeout 600 python bench.py --workload recursive_2p20 --mode shard --steps 3 --warmup 1 --no-cpu-baseline --no-north-star > $O/bench_rec.json 2> $O/bench_rec.err
python -c "
import json; d=json.load(open('$O/bench_rec.json')); print('shard python rec', round(d['value'],4))"
echo done
