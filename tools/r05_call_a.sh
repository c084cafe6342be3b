#!/bin/bash
# GPU box, round 5 call A: the new multi-process / self-check tests, the probes' own clock, and the A/B of this round's kernel changes
# (CTI twiddles from LDS; XYZZ accumulation; 26-bit Pedersen windows) -> gpurun_out/r05a/
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05a
rm -rf $OUT; mkdir -p $OUT
cd $R
( timeout 20 rocm-smi --showcomputepartition --showmemorypartition 2>&1 | tail -12 ) > $OUT/partition.txt; cat $OUT/partition.txt
timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_sharded_host.py -k "processes or self_check or rccl" tests/test_gpu_parity.py 2>&1 | tail -5 | tee $OUT/pytest_new.txt
tools/_build/ubench > $OUT/ubench.txt 2>&1; tail -26 $OUT/ubench.txt
# ---- NTT: twiddles of strided CTI passes from LDS vs from global memory
tools/ntt_ab.sh run tw_global tw_lds tw_global tw_lds 2>&1 | grep -v "^$" | tee $OUT/ntt_ab.txt
# ---- Pedersen: Jacobian vs XYZZ accumulation, 24- vs 26-bit windows
bench() { name=$1; wl=$2; shift; shift; env "$@" timeout 200 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end > $OUT/bench_$name.json 2> $OUT/bench_$name.err; python -c "
import json
try:
    d=json.load(open('$OUT/bench_$name.json')); print('$name', round(d['value'],4), {k:round(v,2) for k,v in d['stage_ms_per_proof'].items()})
except Exception as e: print('$name FAILED', e)" | tee -a $OUT/ped_ab.txt; }
cp sandstorm_amd/_build/libsandstorm_hip.so /tmp/libsandstorm_hip.orig.so
for rep in 1 2; do
  cp tools/_build/variants/ped_jac/libsandstorm_hip.so sandstorm_amd/_build/libsandstorm_hip.so
  bench jac_2p16_$rep recursive_2p16 A=1
  bench jac_2p20_$rep recursive_2p20 A=1
  cp /tmp/libsandstorm_hip.orig.so sandstorm_amd/_build/libsandstorm_hip.so
  bench xyzz_2p16_$rep recursive_2p16 A=1
  bench xyzz_2p20_$rep recursive_2p20 A=1
done
bench xyzz_w26_2p16 recursive_2p16 SS_PED_WINDOW=26
bench xyzz_w26_2p20 recursive_2p20 SS_PED_WINDOW=26
# ---- the default line with the live stage clocks (no CPU legs here)
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err
python -c "
import json
d=json.load(open('$OUT/bench_default.json')); print('default', d['value'], d['stage_ms_per_proof']); print('alu', d['roofline']['alu']); print('e2e', d.get('end_to_end')); print('ns', d['north_star']['value'], d['north_star']['roofline']['alu'])" | tee $OUT/default_summary.txt
tail -5 $OUT/bench_default.err
ls $OUT
