#!/bin/bash
# GPU box, round 6 call N: the Pedersen kernels in the R280 domain (table points x 2^280, the ten-step reduction alone) -> gpurun_out/r06n/
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06n
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest -m gpu -q -x tests/test_gpu_parity.py tests/test_gpu_recursive_claim.py tests/test_gpu_sharded_host.py tests/test_gpu_real_air.py tests/test_gpu_prove.py 2>&1 | tail -3 | tee $OUT/pytest.txt
FLAGS="--no-cpu-baseline --no-north-star --no-end-to-end --no-stage-clocks --steps 10 --warmup 2"
for w in recursive_2p20 recursive_2p16 array_sum_example starknet_2p20; do
  timeout 300 python bench.py --workload $w $FLAGS > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python -c "import json; d=json.load(open('$OUT/bench_$w.json')); print('$w', d['value'], d['stage_ms_per_proof'])" | tee -a $OUT/summary.txt
done
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rp_n && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_n -- python $R/bench.py --workload recursive_2p20 --no-cpu-baseline --no-north-star --no-end-to-end --no-stage-clocks --steps 2 --warmup 1 > /dev/null 2>&1)
f=$(find /tmp/rp_n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -i "Name\|pedersen" $f | cut -d, -f1-7 > $OUT/kernel_stats_pedersen.csv; cat $OUT/kernel_stats_pedersen.csv | cut -c1-200
