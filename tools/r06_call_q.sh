#!/bin/bash
# GPU box, round 6 call Q: an upper bound for a limb-form copy of the LDE columns - the recursive layout's constraint kernel built
# with its 442 re-limbings of loaded cells compiled away (tools/_build/variants/libss_abl.so: wrong values, the timing of the rest)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06q
rm -rf $OUT; mkdir -p $OUT
cd $R
FLAGS="--workload recursive_2p20 --no-cpu-baseline --no-north-star --no-end-to-end --no-stage-clocks --steps 10 --warmup 2"
cp sandstorm_amd/_build/libsandstorm_hip.so /tmp/good.so
for v in good abl good abl; do
  [ $v = abl ] && cp tools/_build/variants/libss_abl.so sandstorm_amd/_build/libsandstorm_hip.so || cp /tmp/good.so sandstorm_amd/_build/libsandstorm_hip.so
  timeout 300 python bench.py $FLAGS > $OUT/b.json 2> $OUT/b.err
  python -c "import json; d=json.load(open('$OUT/b.json')); print('$v', d['value'], d['stage_ms_per_proof']['quotient'])" | tee -a $OUT/summary.txt
done
