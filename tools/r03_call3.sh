#!/bin/bash
# round 3, third GPU call: second set of constraint-kernel variants; Pedersen windows 20 / 22 / 24 with the process-wide table
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_call3; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/qg_bench.py starknet 20 > $O/qg_starknet.json 2> $O/qg_starknet.err; echo "qg starknet rc=$?"; cat $O/qg_starknet.json
timeout 300 python tools/qg_bench.py recursive 20 > $O/qg_recursive.json 2> $O/qg_recursive.err; echo "qg recursive rc=$?"; cat $O/qg_recursive.json
for W in 20 22 24 16; do
  SS_PED_WINDOW=$W timeout 200 python bench.py --workload recursive_2p20 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_rec_w$W.json 2> $O/bench_rec_w$W.err
  python -c "
import json; d=json.load(open('$O/bench_rec_w$W.json')); print('recursive_2p20 W=$W', round(d['value'],4), d['stage_ms_per_proof'])"
  SS_PED_WINDOW=$W timeout 100 python bench.py --workload array_sum_example --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_ex_w$W.json 2> $O/bench_ex_w$W.err
  python -c "
import json; d=json.load(open('$O/bench_ex_w$W.json')); print('array_sum_example W=$W', round(d['value'],4), d['stage_ms_per_proof'])"
  SS_PED_WINDOW=$W timeout 100 python bench.py --workload recursive_2p16 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_r16_w$W.json 2> $O/bench_r16_w$W.err
  python -c "
import json; d=json.load(open('$O/bench_r16_w$W.json')); print('recursive_2p16 W=$W', round(d['value'],4), d['stage_ms_per_proof'])"
  SS_PED_WINDOW=$W timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pedersen or friendly" 2>&1 | tail -1
done
echo done
