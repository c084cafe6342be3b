#!/bin/bash
# GPU box, round 6 call J: the recursive layout's constraint kernel cut into 2 / 3 parts (library built with `make QG_AB=1`) -> gpurun_out/r06j/
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06j
rm -rf $OUT; mkdir -p $OUT
cd $R
for v in 1 2; do SS_QG_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_real_quotient.py -m gpu -q -x -k recursive 2>&1 | tail -1 | tee -a $OUT/pytest.txt; done
FLAGS="--no-cpu-baseline --no-north-star --no-end-to-end --no-stage-clocks --steps 10 --warmup 2"
for v in 0 1 2 0 1 2; do
  SS_QG_VARIANT=$v timeout 300 python bench.py --workload recursive_2p20 $FLAGS > $OUT/b.json 2> $OUT/b.err
  python -c "import json; d=json.load(open('$OUT/b.json')); print('recursive_2p20 variant $v', d['value'], d['stage_ms_per_proof']['quotient'])" | tee -a $OUT/summary.txt
done
