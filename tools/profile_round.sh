#!/bin/bash
# GPU box: the round's measurement artefacts for the default bench workload, into gpurun_out/prof/.
#   1. rocprofv3 --kernel-trace --stats  (per-kernel durations)
#   2. rocprofv3 --pmc FETCH_SIZE        (separate pass, MI355X_MICROARCH.md)
#   3. rocprofv3 --pmc WRITE_SIZE        (separate pass)
# Each on:  python bench.py --steps 1 --warmup 0 --no-cpu-baseline   (2 proofs: 1 untimed + 1 timed)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-north-star --no-end-to-end --no-stage-clocks ${WORKLOAD:+--workload $WORKLOAD}"
rm -rf /tmp/rp_stats /tmp/rp_fetch /tmp/rp_write
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats -- $CMD > $OUT/stats_run.log 2>&1
f=$(ls /tmp/rp_stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv
f=$(ls /tmp/rp_stats/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$f" ] && python $R/tools/trace_summary.py "$f" > $OUT/kernel_trace_summary.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/rp_fetch -- $CMD > $OUT/fetch_run.log 2>&1
f=$(ls /tmp/rp_fetch/*/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python $R/tools/pmc_summary.py "$f" > $OUT/pmc_fetch.txt
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/rp_write -- $CMD > $OUT/write_run.log 2>&1
f=$(ls /tmp/rp_write/*/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python $R/tools/pmc_summary.py "$f" > $OUT/pmc_write.txt
ls -la $OUT
