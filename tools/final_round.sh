#!/bin/bash
# GPU box: the round-end evidence in one call -> gpurun_out/final/  (collected into profiles/ by tools/collect_final.py <tag>)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -16 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.txt
for w in starknet_2p20 recursive_2p20 recursive_2p16 array_sum_example; do
  timeout 600 python bench.py --workload $w > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python -c "import json; d=json.load(open('$OUT/bench_$w.json')); print('$w', d['value'], d['stage_ms_per_proof'], d['ntt_gfield_ops_per_s'], d.get('cpu_baseline',{}).get('measured_sample_s'))"
done
timeout 300 python bench.py --workload starknet_2p20 --mode shard --sharded-host cpp --no-cpu-baseline > $OUT/bench_starknet_2p20_shard_cpp_1gpu.json 2> $OUT/bench_shard_cpp.err
tools/_build/ubench > $OUT/ubench.txt 2>&1
tools/_build/mfma_mulbench > $OUT/mfma_mulbench.txt 2>&1
tools/_build/mulbench > $OUT/mulbench.txt 2>&1
tools/_build/fma_mulbench > $OUT/fma_mulbench.txt 2>&1
# files -> proof through the device generator, without torch; then the generator's kernels by name
timeout 300 python tools/e2e_device.py starknet recursive > $OUT/e2e_device.txt 2>&1; cat $OUT/e2e_device.txt
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rp_trace && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_trace -- python $R/tools/e2e_device.py starknet recursive > /dev/null 2>&1)
f=$(find /tmp/rp_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && ( head -1 $f; grep -i "trace_\|mem_\|scan_chunks\|scan_sums\|scan_apply_kernel_u32" $f ) > $OUT/device_trace_kernel_stats.csv
timeout 300 python bench.py --workload goldilocks_lde_2p20 --steps 10 --warmup 2 > $OUT/bench_goldilocks_lde_2p20.json 2> $OUT/bench_gl.err
timeout 300 python bench.py --workload goldilocks_plain_2p20 --steps 3 --warmup 1 > $OUT/bench_goldilocks_plain_2p20.json 2> $OUT/bench_glp.err
timeout 300 python bench.py --workload starknet_2p22 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_starknet_2p22.json 2> $OUT/bench_2p22.err
python -c "
import json
for w in ('starknet_2p20_shard_cpp_1gpu','goldilocks_lde_2p20','goldilocks_plain_2p20','starknet_2p22'):
    try: d=json.load(open('$OUT/bench_%s.json'%w)); print(w, d['value'])
    except Exception as e: print(w, 'FAILED', e)"
bash tools/profile_round.sh > $OUT/profile_round.log 2>&1
cp -r $R/gpurun_out/prof $OUT/prof_starknet_2p20
WORKLOAD=recursive_2p20 bash tools/profile_round.sh > $OUT/profile_round_rec.log 2>&1
cp -r $R/gpurun_out/prof $OUT/prof_recursive_2p20
# SQ instruction / wait counters of the default workload (own pass: --pmc only)
bash tools/pmc_run.sh sq_final "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-north-star --no-end-to-end --no-stage-clocks > $OUT/sq_counters_starknet_2p20.txt 2>&1
bash tools/pmc_run.sh sq_final_rec "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" python $R/bench.py --workload recursive_2p20 --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-stage-clocks > $OUT/sq_counters_recursive_2p20.txt 2>&1
for w in starknet_2p20 recursive_2p20; do bash tools/valu_busy.sh $w $OUT/valu_busy > $OUT/valu_busy_$w.log 2>&1; done
bash tools/gl64_pmc.sh > $OUT/gl64_pmc.log 2>&1
# where the device sat idle inside a proof (tools/kernel_gaps.py over a kernel trace of 4 proofs)
for w in starknet_2p20 recursive_2p20 goldilocks_plain_2p20; do
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rp_gaps_$w && timeout 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_gaps_$w -- python $R/bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-north-star --no-end-to-end --no-stage-clocks > /dev/null 2>&1; python $R/tools/kernel_gaps.py /tmp/rp_gaps_$w 300 0.45 > $OUT/kernel_gaps_$w.txt 2>&1)
done
ls $OUT
