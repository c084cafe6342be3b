# GPU box: Pedersen parity after a kernel change + the Merkle stage of the Pedersen-tree workloads, into gpurun_out/r2d/
mkdir -p gpurun_out/r2d
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_prove.py tests/test_gpu_real_air.py -m gpu -q -x -k "pedersen or friendly or merkle or tree or real" > gpurun_out/r2d/pytest.txt 2>&1; tail -3 gpurun_out/r2d/pytest.txt
run() { name=$1; wl=$2; shift; shift; env "$@" timeout 120 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r2d/bench_$name.json 2>/dev/null; python -c "
import json,sys
d=json.load(open('gpurun_out/r2d/bench_$name.json')); print('$name', round(d['value'],4), {k:round(v,1) for k,v in d['stage_ms_per_proof'].items()})"; }
run recursive recursive_2p20 A=1
run array_sum array_sum_example A=1
run recursive16 recursive_2p16 A=1
run sn_wgs8192 starknet_2p20 SS_DEEP_WGS=8192
run sn_wgs16384 starknet_2p20 SS_DEEP_WGS=16384
run sn_default starknet_2p20 A=1
