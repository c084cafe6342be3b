#!/bin/bash
# round 3, fifth GPU call: the re-written out-of-domain kernels (registers, not scratch), Pedersen table build in two levels
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_call5; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_real_quotient.py tests/test_gpu_prove.py -m gpu -x -q -k "ood or quotient or deep or pedersen or friendly or merkle or cairo" ) > $O/pytest_sel.txt 2>&1; tail -3 $O/pytest_sel.txt
run() { name=$1; wl=$2; shift 2; env "$@" timeout 200 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-north-star > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "
import json; d=json.load(open('$O/bench_$name.json')); print('$name', round(d['value'],4), d['stage_ms_per_proof'])" || tail -3 $O/bench_$name.err; }
run sn_default starknet_2p20 X=1
run sn_transform starknet_2p20 SS_OOD_TRANSFORM=1
run sn_S2 starknet_2p20 SS_OOD_BLOCK_LOG=2
run sn_S3 starknet_2p20 SS_OOD_BLOCK_LOG=3
run sn_S4 starknet_2p20 SS_OOD_BLOCK_LOG=4
run rec_default recursive_2p20 X=1
run rec_w20 recursive_2p20 SS_PED_WINDOW=20
run rec_transform recursive_2p20 SS_OOD_TRANSFORM=1
run r16_default recursive_2p16 X=1
run ex_default array_sum_example X=1
for W in 20 24; do
SS_PED_WINDOW=$W python - <<'PY'
import os, time
from sandstorm_amd import backend as be
t0=time.time(); ctx=be.Context(0); n=1<<12
leaves=ctx.alloc(32*n); ctx.zero(leaves)
nodes=ctx.alloc(64*n); tags=ctx.alloc(2*n)
ctx.sync(); t1=time.time()
ctx.merkle_build(be.TREE_FRIENDLY, 22, be.LEAF_DIGEST, leaves, n, nodes, tags); ctx.sync(); t2=time.time()
print("W=%s: context %.3f s, first friendly tree (two-level table build) %.3f s" % (os.environ["SS_PED_WINDOW"], t1-t0, t2-t1))
PY
done
echo done
