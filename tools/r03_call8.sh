#!/bin/bash
# round 3: the C++ host's sharded prover on the device (ranks as threads sharing the GPU; RCCL with a group of one), the sharded
# drivers' overhead on one rank, the final constraint kernels inside whole proofs
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_call8; mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_sharded_host.py tests/test_gpu_real_quotient.py tests/test_gpu_recursive_claim.py -m gpu -x -q --durations=6 ) > $O/pytest_sel.txt 2>&1; tail -12 $O/pytest_sel.txt
run() { name=$1; shift; timeout 300 python bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-north-star > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "
import json; d=json.load(open('$O/bench_$name.json')); print('$name', round(d['value'],4), d.get('stage_ms_per_proof',''))" || tail -5 $O/bench_$name.err; }
run sn_single --workload starknet_2p20
run sn_shard_py --workload starknet_2p20 --mode shard
run sn_shard_cpp --workload starknet_2p20 --mode shard --sharded-host cpp
run rec_single --workload recursive_2p20
run rec_shard_py --workload recursive_2p20 --mode shard
run rec_shard_cpp --workload recursive_2p20 --mode shard --sharded-host cpp
SS_SHARD_TIMING=1 timeout 300 python bench.py --workload starknet_2p20 --mode shard --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep "shard timing" | tail -14
echo done
