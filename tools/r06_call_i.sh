#!/bin/bash
# GPU box, round 6 call I: the extension scans with fused passes, the Pedersen finish chunk -> gpurun_out/r06i/
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06i
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_extension.py tests/test_gpu_reference_proof.py tests/test_gpu_recursive_claim.py 2>&1 | tail -4 | tee $OUT/pytest.txt
FLAGS="--no-cpu-baseline --no-north-star --no-end-to-end --no-stage-clocks --steps 10 --warmup 2"
for ch in 8 16 32; do
  SS_PED_FINISH_CHUNK=$ch timeout 300 python bench.py --workload recursive_2p20 $FLAGS > $OUT/bench_recursive_2p20_finish$ch.json 2> $OUT/bench_finish$ch.err
  python -c "import json; d=json.load(open('$OUT/bench_recursive_2p20_finish$ch.json')); print('recursive_2p20 finish chunk $ch', d['value'], d['stage_ms_per_proof'])" | tee -a $OUT/summary.txt
done
for w in starknet_2p20 recursive_2p16; do
  timeout 300 python bench.py --workload $w $FLAGS > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python -c "import json; d=json.load(open('$OUT/bench_$w.json')); print('$w', d['value'], d['stage_ms_per_proof'])" | tee -a $OUT/summary.txt
done
for lc in 2 3 4; do for li in 5 6 7; do
  SS_SCAN_LOG_CHUNK=$lc SS_INV_LOG_CHUNK=$li timeout 300 python bench.py --workload recursive_2p20 --no-cpu-baseline --no-north-star --no-end-to-end --no-stage-clocks --steps 4 --warmup 1 > $OUT/b.json 2> $OUT/b.err
  python -c "import json; d=json.load(open('$OUT/b.json')); print('scan chunk $lc inv chunk $li', d['value'], d['stage_ms_per_proof']['extension_scans'])" | tee -a $OUT/summary.txt
done; done
