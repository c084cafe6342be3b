#!/bin/bash
# GPU box, round 6 call H: the inverse tables with interleaved chunks + DEEP's tables queued behind the out-of-domain values -> gpurun_out/r06h/
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06h
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_parity.py -k "deep or inverse or quotient or fri" tests/test_gpu_row_blocks.py tests/test_gpu_reference_proof.py 2>&1 | tail -4 | tee $OUT/pytest.txt
FLAGS="--no-cpu-baseline --no-north-star --no-end-to-end --no-stage-clocks --steps 10 --warmup 2"
for w in recursive_2p20 starknet_2p20; do
  for ch in 6 7 8; do
    SS_BATCH_INV_LOG_CHUNK=$ch timeout 300 python bench.py --workload $w $FLAGS > $OUT/bench_${w}_chunk$ch.json 2> $OUT/bench_${w}_chunk$ch.err
    python -c "import json; d=json.load(open('$OUT/bench_${w}_chunk$ch.json')); print('$w chunk $ch', d['value'], d['stage_ms_per_proof'])" | tee -a $OUT/summary.txt
  done
done
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rp_h && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_h -- python $R/bench.py --workload recursive_2p20 --no-cpu-baseline --no-north-star --no-end-to-end --no-stage-clocks --steps 2 --warmup 1 > /dev/null 2>&1)
f=$(find /tmp/rp_h -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 $f > $OUT/kernel_stats_recursive_2p20.csv
grep -i "batch_inverse\|fri_fold" $OUT/kernel_stats_recursive_2p20.csv
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rp_gaps_h && timeout 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_gaps_h -- python $R/bench.py --workload recursive_2p20 --steps 4 --warmup 1 --no-cpu-baseline --no-north-star --no-end-to-end --no-stage-clocks > /dev/null 2>&1; python $R/tools/kernel_gaps.py /tmp/rp_gaps_h 300 0.45 > $OUT/kernel_gaps_recursive_2p20.txt 2>&1)
head -3 $OUT/kernel_gaps_recursive_2p20.txt; grep "  at " $OUT/kernel_gaps_recursive_2p20.txt | head
