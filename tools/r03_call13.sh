#!/bin/bash
# what a small Pedersen level waits for: SQ / instruction-cache counters of pedersen_pairs_small_kernel
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_call13; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
true
cd /root/repo
CMD="python /root/repo/bench.py --workload array_sum_example --steps 1 --warmup 0 --no-cpu-baseline --no-north-star"
bash tools/pmc_run.sh ped1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH" $CMD 2>&1 | grep -A9 "pedersen_pairs_small\|pedersen_finish\|pedersen_acc_pairs_kernel" | head -60
bash tools/pmc_run.sh ped2 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU" $CMD 2>&1 | grep -A8 "pedersen_pairs_small" | head -30
echo done
