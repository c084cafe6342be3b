#!/bin/bash
# round 3: the decoded flags recomputed at their uses (tools/gen_quotient.py rematerialize_cheap_slots), A/B
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_call7; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/qg_bench.py starknet 20 > $O/qg_starknet.json 2> $O/qg_starknet.err; echo "qg starknet rc=$?"; cat $O/qg_starknet.json
timeout 300 python tools/qg_bench.py recursive 20 > $O/qg_recursive.json 2> $O/qg_recursive.err; echo "qg recursive rc=$?"; cat $O/qg_recursive.json
( timeout 600 python -m pytest tests/test_gpu_real_quotient.py -m gpu -x -q ) 2>&1 | tail -2
echo done
