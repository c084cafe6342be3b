#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r05b; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_sharded_host.py -k "processes or self_check or rccl" tests/test_gpu_parity.py 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee $OUT/pytest_new.txt
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $OUT/bench_default.json 2> $OUT/bench_default.err
python -c "
import json
d=json.load(open('$OUT/bench_default.json')); print('default', d['value'], d['stage_ms_per_proof']); print('alu ntt', d['roofline']['alu']); print('alu dom', d['roofline_dominant']['alu']); print('ns', d['north_star']['value'], d['north_star']['roofline']['alu'], d['north_star']['roofline_dominant']['alu'])" | tee $OUT/default_summary.txt
tail -5 $OUT/bench_default.err
