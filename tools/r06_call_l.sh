#!/bin/bash
# GPU box, round 6 call L: the Montgomery reduction with limbs 6 and 7 of p in one lazy column (a multiply-add less per step) -> gpurun_out/r06l/
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06l
rm -rf $OUT; mkdir -p $OUT
cd $R
tools/_build/mulbench > $OUT/mulbench.txt 2>&1; tail -12 $OUT/mulbench.txt
timeout 1200 python -m pytest -m gpu -q -x tests/test_gpu_parity.py tests/test_gpu_reference_proof.py tests/test_gpu_real_quotient.py tests/test_gpu_extension.py 2>&1 | tail -3 | tee $OUT/pytest.txt
FLAGS="--no-cpu-baseline --no-north-star --no-end-to-end --no-stage-clocks --steps 10 --warmup 2"
for w in recursive_2p20 starknet_2p20 recursive_2p16; do
  timeout 300 python bench.py --workload $w $FLAGS > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python -c "import json; d=json.load(open('$OUT/bench_$w.json')); print('$w', d['value'], d['stage_ms_per_proof'])" | tee -a $OUT/summary.txt
done
