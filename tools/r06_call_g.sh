#!/bin/bash
# GPU box, round 6 call G: the extension trace as row blocks (ABI 12) on the MI355X -> gpurun_out/r06g/
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06g
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_extension.py tests/test_gpu_sharded_host.py 2>&1 | tail -6 | tee $OUT/pytest_blocks.txt
timeout 1500 python -m pytest -m gpu -q -x tests/test_gpu_full_size.py -k "2p20" 2>&1 | tail -6 | tee $OUT/pytest_full_size_2p20.txt
timeout 400 python bench.py --workload starknet_2p20 --mode shard --sharded-host cpp --no-cpu-baseline > $OUT/bench_starknet_2p20_shard_cpp_1gpu.json 2> $OUT/bench_shard.err
timeout 400 python bench.py --workload recursive_2p20 --mode shard --sharded-host cpp --no-cpu-baseline > $OUT/bench_recursive_2p20_shard_cpp_1gpu.json 2>> $OUT/bench_shard.err
python - <<'PY'
import json
for w in ("starknet", "recursive"):
    try:
        d = json.load(open("gpurun_out/r06g/bench_%s_2p20_shard_cpp_1gpu.json" % w))
        print(w, d["value"], d.get("stage_ms_per_proof"), d["config"].get("transport"))
    except Exception as e:
        print(w, "FAILED", e)
PY
tail -3 $OUT/bench_shard.err
