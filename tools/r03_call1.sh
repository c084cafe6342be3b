#!/bin/bash
# round 3, first GPU call: the new parity tests + the whole gpu suite, the driver's bench line (north_star inside), the sharded
# driver on one rank with stage timings, the 64-bit field's fixture re-made on the device
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_call1; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -25 $O/pytest_gpu.txt
timeout 600 python bench.py --steps 5 --warmup 1 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
tail -c 1500 $O/bench_default.json
SS_SHARD_TIMING=1 timeout 300 python bench.py --mode shard --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_shard1.json 2> $O/bench_shard1.err; echo "shard rc=$?"
grep "shard timing" $O/bench_shard1.err | tail -20; tail -c 600 $O/bench_shard1.json
timeout 200 python tests/golden/make_goldilocks_proof.py $O/goldilocks_plain_proof.npz > $O/gl_fixture.txt 2>&1; echo "fixture rc=$?"
cp gpurun_out/array_sum_recursive_cairo.proof $O/ 2>/dev/null
echo done
