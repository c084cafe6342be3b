#!/bin/bash
# GPU box, round 6 call T: the 32-lanes-per-hash Pedersen kernel's threshold (SS_PED_SMALL_MAX) after the kernels moved to the R280 domain
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06t
rm -rf $OUT; mkdir -p $OUT
cd $R
FLAGS="--no-cpu-baseline --no-north-star --no-end-to-end --no-stage-clocks --steps 10 --warmup 2"
for m in 4096 8192 16384 32768; do
  for w in recursive_2p20 recursive_2p16 array_sum_example; do
    SS_PED_SMALL_MAX=$m timeout 300 python bench.py --workload $w $FLAGS > $OUT/b.json 2> $OUT/b.err
    python -c "import json; d=json.load(open('$OUT/b.json')); print('small_max $m $w', round(d['value'],5), d['stage_ms_per_proof']['merkle'])" | tee -a $OUT/summary.txt
  done
done
