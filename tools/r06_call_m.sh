#!/bin/bash
# GPU box, round 6 call M: the five-instruction Montgomery step (columns that start at 2^28 - 1) - three builds of the library:
#   v1 everything on the new step, the constraint kernels' wide sums take the bias inside the steps; v2 the wide sums' columns start at the
#   bias too; v3 the generated constraint kernels keep the seven-instruction step (their register allocation is the delicate one)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06m
rm -rf $OUT; mkdir -p $OUT
cd $R
FLAGS="--no-cpu-baseline --no-north-star --no-end-to-end --no-stage-clocks --steps 10 --warmup 2"
for v in v1 v2 v3; do
  cp tools/_build/variants/libss_$v.so sandstorm_amd/_build/libsandstorm_hip.so
  timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_real_quotient.py tests/test_gpu_reference_proof.py tests/test_gpu_parity.py -k "not lde_vs_oracle" 2>&1 | tail -1 | tee -a $OUT/pytest_$v.txt
  for w in recursive_2p20 starknet_2p20; do
    timeout 300 python bench.py --workload $w $FLAGS > $OUT/b.json 2> $OUT/b.err
    python -c "import json; d=json.load(open('$OUT/b.json')); print('$v $w', d['value'], d['stage_ms_per_proof'])" | tee -a $OUT/summary.txt
  done
done
cp tools/_build/variants/libss_v1.so sandstorm_amd/_build/libsandstorm_hip.so
tools/_build/mulbench 2>&1 | tail -7 | tee $OUT/mulbench_v1.txt
