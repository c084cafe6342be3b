#!/bin/bash
# GPU box, round 6 call S: the FRI fold in limb form -> gpurun_out/r06s/
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06s
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_parity.py tests/test_gpu_reference_proof.py tests/test_gpu_recursive_claim.py -k "fri or reference or claim" 2>&1 | grep -E "passed|failed|error" | tail -2 | tee $OUT/pytest.txt
FLAGS="--no-cpu-baseline --no-north-star --no-end-to-end --no-stage-clocks --steps 10 --warmup 2"
for w in recursive_2p20 starknet_2p20; do
  timeout 300 python bench.py --workload $w $FLAGS > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python -c "import json; d=json.load(open('$OUT/bench_$w.json')); print('$w', d['value'], d['stage_ms_per_proof'])" | tee -a $OUT/summary.txt
done
