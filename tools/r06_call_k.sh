#!/bin/bash
# GPU box, round 6 call K: the two headline bench lines once more, after collect_final wrote this round's counter files (the lines then
# cite profiles/r06_final_* for their traffic and instruction counters), and the driver-style default run with its wall time
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06k
rm -rf $OUT; mkdir -p $OUT
cd $R
T0=$(date +%s.%N)
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
T1=$(date +%s.%N); echo "python bench.py (no flags): $(echo "$T1 - $T0" | bc) s wall" | tee $OUT/default_time.txt
timeout 600 python bench.py --workload recursive_2p20 > $OUT/bench_recursive_2p20.json 2> $OUT/bench_recursive_2p20.err
python - <<'PY'
import json
for f in ("bench_default", "bench_recursive_2p20"):
    d = json.load(open("gpurun_out/r06k/%s.json" % f))
    print(f, d["value"], d["stage_ms_per_proof"], d.get("stage_alu_frac"), d["roofline"]["traffic_source"], (d.get("north_star") or {}).get("value"), (d.get("north_star") or {}).get("steps"))
PY
