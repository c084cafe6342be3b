#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_call9; mkdir -p $O
export TMPDIR=/tmp
timeout 120 python tools/rccl_probe.py 2>&1 | tail -3
timeout 120 python tools/rccl_probe.py torch 2>&1 | tail -3
NCCL_SOCKET_IFNAME= timeout 200 python tools/rccl_probe.py 2>&1 | tail -2
run() { name=$1; shift; timeout 300 python bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-north-star > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "
import json; d=json.load(open('$O/bench_$name.json')); print('$name', round(d['value'],4), d.get('stage_ms_per_proof',''))" || tail -5 $O/bench_$name.err; }
run sn_shard_cpp --workload starknet_2p20 --mode shard --sharded-host cpp
run rec_shard_cpp --workload recursive_2p20 --mode shard --sharded-host cpp
( time timeout 300 python -m pytest tests/test_gpu_sharded_host.py -m gpu -x -q -k rccl ) 2>&1 | tail -5
echo done
