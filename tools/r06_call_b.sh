#!/bin/bash
# GPU box, round 6 call B: the whole GPU suite after the device generator / ADVICE changes, the driver-style bench line (files -> proof
# through the device generator in its end_to_end leg), a kernel trace of the generator -> gpurun_out/r06b/
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06b
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 2400 python -m pytest -m gpu -q -x tests 2>&1 | tail -12 | tee $OUT/pytest_gpu.txt
timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/bench_default.json 2> $OUT/bench_default.err
python -c "
import json
d=json.load(open('$OUT/bench_default.json')); print('default', d['value'], d['stage_ms_per_proof']); print('e2e', json.dumps(d.get('end_to_end'))[:1500]); print('ns', d['north_star']['value'], json.dumps(d['north_star'].get('end_to_end'))[:1500])" | tee $OUT/default_summary.txt
tail -5 $OUT/bench_default.err
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rp_trace && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_trace -- python $R/tools/e2e_device.py starknet recursive > $OUT/prof_run.txt 2>&1)
f=$(find /tmp/rp_trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && ( head -1 $f; grep -i "trace_\|mem_\|scan_" $f ) | cut -c1-300 | tee $OUT/trace_kernel_stats.txt
ls $OUT
