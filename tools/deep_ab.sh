# GPU box: DEEP kernel parity + launch-shape A/B (SS_DEEP_WGS, SS_DEEP_NO_XCD_MAP), into gpurun_out/r2c/
mkdir -p gpurun_out/r2c
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_prove.py tests/test_gpu_reference_proof.py -m gpu -q -x -k "deep or row_block or pedersen or friendly or prove or reference or merkle" > gpurun_out/r2c/pytest_deep.txt 2>&1; tail -3 gpurun_out/r2c/pytest_deep.txt
run() { name=$1; shift; env "$@" timeout 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r2c/bench_$name.json 2>/dev/null; python -c "
import json,sys
d=json.load(open('gpurun_out/r2c/bench_$name.json')); print('$name', round(d['value'],4), {k:round(v,1) for k,v in d['stage_ms_per_proof'].items()})"; }
run default A=1
run wgs512 SS_DEEP_WGS=512
run wgs2048 SS_DEEP_WGS=2048
run wgs4096 SS_DEEP_WGS=4096
run noxcd SS_DEEP_NO_XCD_MAP=1
run noxcd4096 SS_DEEP_NO_XCD_MAP=1 SS_DEEP_WGS=4096
timeout 120 python bench.py --workload recursive_2p20 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r2c/bench_recursive.json 2>/dev/null; python -c "
import json
d=json.load(open('gpurun_out/r2c/bench_recursive.json')); print('recursive', round(d['value'],4), {k:round(v,1) for k,v in d['stage_ms_per_proof'].items()})"
