# GPU box: A/B of the generated constraint-kernel variants (library built with `make QG_AB=1`), into gpurun_out/qgab/
mkdir -p gpurun_out/qgab
for v in 1 2 3; do SS_QG_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_real_quotient.py -m gpu -q -x 2>&1 | tail -1; done
run() { name=$1; wl=$2; shift; shift; env "$@" timeout 160 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/qgab/bench_$name.json 2>gpurun_out/qgab/err_$name.txt; python -c "
import json,sys
d=json.load(open('gpurun_out/qgab/bench_$name.json')); print('$name', round(d['value'],4), 'quotient', round(d['stage_ms_per_proof']['quotient'],1))" || tail -2 gpurun_out/qgab/err_$name.txt; }
for v in 0 1 2 3; do run sn_v$v starknet_2p20 SS_QG_VARIANT=$v; done
for v in 0 1 2 3; do run rec_v$v recursive_2p20 SS_QG_VARIANT=$v; done
