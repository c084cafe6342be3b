#!/bin/bash
# GPU box, round 6 call A: the base trace made on the device - parity against the host generator (2^14 ... 2^20 steps), the wall time of
# the generator and of files -> proof through it, a kernel trace of one generation -> gpurun_out/r06a/
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06a
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_device_trace.py 2>&1 | tail -8 | tee $OUT/pytest_device_trace.txt
timeout 600 python tools/e2e_device.py starknet recursive 2>&1 | tee $OUT/e2e_device.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_trace -o trace -- python $R/tools/e2e_device.py starknet > $OUT/prof_run.txt 2>&1
cd $R
f=$(ls $OUT/prof_trace/*/*kernel_stats.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$(find $OUT/prof_trace -name "*kernel_stats*" | head -1)
[ -n "$f" ] && ( head -1 $f; grep -i "trace_\|mem_\|scan_" $f ) | cut -c1-260 | tee $OUT/trace_kernel_stats.txt
find $OUT/prof_trace -name "*.db" -delete 2>/dev/null
find $OUT/prof_trace -size +20M -delete 2>/dev/null
ls $OUT
