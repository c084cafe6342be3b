#!/bin/bash
# GPU box: the 64-bit field's kernels after a change to csrc/gl64.h / goldilocks.hip, in one short call -> gpurun_out/gl64/
#   its GPU tests (every kernel against the oracle, prove / verify / tamper, the GPU-made fixture reproduced), both bench workloads,
#   and a kernel trace of the whole proof
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/gl64
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 200 python -m pytest tests/test_goldilocks.py tests/test_goldilocks_stark.py -m gpu -x -q 2>&1 | tail -8 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
timeout 90 python bench.py --workload goldilocks_plain_2p20 --steps 3 --warmup 1 > $OUT/bench_goldilocks_plain_2p20.json 2> $OUT/bench_glp.err
timeout 60 python bench.py --workload goldilocks_lde_2p20 --steps 10 --warmup 2 > $OUT/bench_goldilocks_lde_2p20.json 2> $OUT/bench_gl.err
python -c "
import json
for w in ('goldilocks_plain_2p20','goldilocks_lde_2p20'):
    try: d=json.load(open('$OUT/bench_%s.json'%w)); print(w, d['value'], d.get('stage_ms_per_proof'), d['roofline']['frac'])
    except Exception as e: print(w, 'FAILED', e)"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_gl
timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_gl -- python $R/bench.py --workload goldilocks_plain_2p20 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/stats_run.log 2>&1
f=$(ls /tmp/rp_gl/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv
f=$(ls /tmp/rp_gl/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$f" ] && python $R/tools/trace_summary.py "$f" > $OUT/kernel_trace_summary.txt
head -12 $OUT/kernel_trace_summary.txt
cd $R
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 80 python -m pytest tests/test_gpu_reference_proof.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest_reference_proof.txt
