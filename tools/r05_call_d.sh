#!/bin/bash
# GPU box: the re-landed r280 table products - the size test FIRST, then parity, then the stage times
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r05d; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_full_size.py -k "two_ranks_at_2p22" 2>&1 | grep -E "passed|failed|error|Error" | tail -3 | tee $OUT/pytest_2p22.txt
timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_real_quotient.py tests/test_gpu_reference_proof.py 2>&1 | grep -E "passed|failed|error|Error" | tail -3 | tee $OUT/pytest_quotient.txt
for rep in 1 2; do
for w in starknet_2p20 recursive_2p20; do
  timeout 300 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end --no-north-star > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python -c "
import json
d=json.load(open('$OUT/bench_$w.json')); print('$w', round(d['value'],4), {k:round(v,2) for k,v in d['stage_ms_per_proof'].items()})" | tee -a $OUT/summary.txt
done; done
