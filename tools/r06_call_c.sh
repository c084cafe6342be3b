#!/bin/bash
# GPU box, round 6 call C: the FP64-FMA multiplier's measured number (VERDICT r5 "next" 4); the transform kernel's two untried items
# (3a: what a 9 x 28-bit format between passes could save at most - conversions ablated; 3b: the clock it runs at against occupancy
# and the LDS exchange) -> gpurun_out/r06c/
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06c
rm -rf $OUT; mkdir -p $OUT
cd $R
tools/_build/fma_mulbench 2>&1 | tee $OUT/fma_mulbench.txt
tools/_build/mulbench 2>&1 | tail -12 | tee $OUT/mulbench.txt
export NTT_BENCH_CLOCK=1
tools/ntt_ab.sh run base noconv occ1 nolds nolds_occ1 t256 base noconv 2>&1 | grep -v "^$" | tee $OUT/ntt_ab.txt
ls $OUT
