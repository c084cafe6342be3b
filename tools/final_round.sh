#!/bin/bash
# GPU box: the round-end evidence in one call -> gpurun_out/final/
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.txt
for w in starknet_2p20 recursive_2p20 recursive_2p16; do
  timeout 600 python bench.py --workload $w > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python -c "import json; d=json.load(open('$OUT/bench_$w.json')); print('$w', d['value'], d['stage_ms_per_proof'], d['ntt_gfield_ops_per_s'], d.get('cpu_baseline',{}).get('value'))"
done
bash tools/profile_round.sh > $OUT/profile_round.log 2>&1
cp -r $R/gpurun_out/prof $OUT/prof_starknet_2p20
WORKLOAD=recursive_2p20 bash tools/profile_round.sh > $OUT/profile_round_rec.log 2>&1
cp -r $R/gpurun_out/prof $OUT/prof_recursive_2p20
ls $OUT
