#!/bin/bash
# round 3, fourth GPU call: third set of constraint-kernel variants; out-of-domain evaluation point by point against the transforms;
# what a Pedersen table of 20 / 24-bit windows costs to build
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_call4; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/qg_bench.py starknet 20 > $O/qg_starknet.json 2> $O/qg_starknet.err; echo "qg starknet rc=$?"; cat $O/qg_starknet.json
timeout 300 python tools/qg_bench.py recursive 20 > $O/qg_recursive.json 2> $O/qg_recursive.err; echo "qg recursive rc=$?"; cat $O/qg_recursive.json
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_real_quotient.py -m gpu -x -q -k "ood or quotient or deep" ) > $O/pytest_ood.txt 2>&1; tail -3 $O/pytest_ood.txt
for mode in sparse transform; do
  for wl in starknet_2p20 recursive_2p20; do
    E=""; [ $mode = transform ] && E="SS_OOD_TRANSFORM=1"
    env $E timeout 200 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-north-star > $O/bench_${wl}_$mode.json 2> $O/bench_${wl}_$mode.err
    python -c "
import json; d=json.load(open('$O/bench_${wl}_$mode.json')); print('$wl $mode', round(d['value'],4), d['stage_ms_per_proof'])"
  done
done
for S in 0 2 4; do
  SS_OOD_BLOCK_LOG=$S timeout 200 python bench.py --workload starknet_2p20 --steps 3 --warmup 1 --no-cpu-baseline --no-north-star > $O/bench_sn_S$S.json 2> $O/bench_sn_S$S.err
  python -c "
import json; d=json.load(open('$O/bench_sn_S$S.json')); print('starknet_2p20 block log $S', round(d['value'],4), d['stage_ms_per_proof'])"
done
for W in 16 20 24; do
SS_PED_WINDOW=$W python - <<'PY'
import os, time, numpy as np, torch
from sandstorm_amd import backend as be
t0=time.time(); ctx=be.Context(0); n=1<<12
leaves=ctx.alloc(32*n); ctx.zero(leaves)
nodes=ctx.alloc(64*n); tags=ctx.alloc(2*n)
ctx.sync(); t1=time.time()
ctx.merkle_build(be.TREE_FRIENDLY, 22, be.LEAF_DIGEST, leaves, n, nodes, tags); ctx.sync(); t2=time.time()
ctx.merkle_build(be.TREE_FRIENDLY, 22, be.LEAF_DIGEST, leaves, n, nodes, tags); ctx.sync(); t3=time.time()
print("W=%s: context %.3f s, first friendly tree (table build) %.3f s, second %.4f s" % (os.environ["SS_PED_WINDOW"], t1-t0, t2-t1, t3-t2))
PY
done
echo done
