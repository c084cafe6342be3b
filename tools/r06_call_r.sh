#!/bin/bash
# GPU box, round 6 call R: the composition program lowered while the device extends the extension trace (Air::prepare_program) -> gpurun_out/r06r/
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06r
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_reference_proof.py tests/test_gpu_recursive_claim.py tests/test_gpu_prove.py 2>&1 | grep -E "passed|failed|error" | tail -2 | tee $OUT/pytest.txt
FLAGS="--no-cpu-baseline --no-north-star --no-end-to-end --no-stage-clocks --steps 10 --warmup 2"
for w in recursive_2p20 starknet_2p20 recursive_2p16; do
  timeout 300 python bench.py --workload $w $FLAGS > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python -c "import json; d=json.load(open('$OUT/bench_$w.json')); print('$w', d['value'], d['stage_ms_per_proof'])" | tee -a $OUT/summary.txt
done
for w in recursive_2p20 starknet_2p20; do
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rp_gaps_r && timeout 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_gaps_r -- python $R/bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-north-star --no-end-to-end --no-stage-clocks > /dev/null 2>&1; python $R/tools/kernel_gaps.py /tmp/rp_gaps_r 300 0.45 > $OUT/kernel_gaps_$w.txt 2>&1)
head -2 $OUT/kernel_gaps_$w.txt; grep -A6 "idle time by" $OUT/kernel_gaps_$w.txt
done
