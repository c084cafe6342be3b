#!/bin/bash
# GPU box: HBM-side traffic (FETCH_SIZE, WRITE_SIZE: separate --pmc passes) and SQ issue counters of a whole proof over the 64-bit
# field -> gpurun_out/gl64pmc/prof_goldilocks_plain_2p20/ ; COLLECT_SRC=gpurun_out/gl64pmc python tools/collect_final.py <tag> collects
R=${GRAFT_REPO_ROOT:-/root/repo}
W=goldilocks_plain_2p20
OUT=$R/gpurun_out/gl64pmc/prof_$W
rm -rf $R/gpurun_out/gl64pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline"
rm -rf /tmp/rp_fetch /tmp/rp_write /tmp/rp_sq
timeout 110 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/rp_fetch -- $CMD > $OUT/fetch_run.log 2>&1
f=$(ls /tmp/rp_fetch/*/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python $R/tools/pmc_summary.py "$f" > $OUT/pmc_fetch.txt
timeout 60 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/rp_write -- $CMD > $OUT/write_run.log 2>&1
f=$(ls /tmp/rp_write/*/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python $R/tools/pmc_summary.py "$f" > $OUT/pmc_write.txt
timeout 60 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/rp_sq -- $CMD > $OUT/sq_run.log 2>&1
f=$(ls /tmp/rp_sq/*/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python $R/tools/pmc_summary.py "$f" > $OUT/sq_counters.txt
ls -la $OUT; grep -A2 "ntt_pass" $OUT/pmc_fetch.txt | head -12
