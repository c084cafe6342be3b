#!/bin/bash
# GPU box, round 6 call D: the CPU port at the bench's FULL size, once (the box's host cores; no kernel runs) -> gpurun_out/r06d/
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06d
rm -rf $OUT; mkdir -p $OUT
cd $R
nproc; free -g | head -2
timeout 1700 python tools/cpu_full_size.py starknet 17 20 2>&1 | grep -v Warning | tee $OUT/cpu_full_size_starknet.txt
timeout 2400 python tools/cpu_full_size.py recursive 14 16 20 2>&1 | grep -v Warning | tee $OUT/cpu_full_size_recursive.txt
ls $OUT
