#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r05c; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_recursive_claim.py -k "files_to_proof" 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee $OUT/pytest_files.txt
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err
python -c "
import json
d=json.load(open('$OUT/bench_default.json')); print('default', d['value']); print('e2e', d['end_to_end']); print('ns e2e', d['north_star']['end_to_end'])" | tee $OUT/summary.txt
tail -5 $OUT/bench_default.err
