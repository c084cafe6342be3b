#!/bin/bash
# GPU box: one --pmc pass for the hardware's own VALU utilisation per kernel:
#   VALU busy = 4 x SQ_ACTIVE_INST_VALU (quad-cycles a wave spends executing vector ALU instructions) / (SIMDs x GRBM_GUI_ACTIVE),
#   effective clock = GRBM_GUI_ACTIVE / kernel duration (the chip clocks to its power budget: MI355X_MICROARCH.md, DVFS)
# usage: tools/valu_busy.sh <workload> <out-dir>
w=${1:-starknet_2p20}; R=$(pwd); mkdir -p ${2:-gpurun_out/valu_busy}; out=$(cd ${2:-gpurun_out/valu_busy} && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_vb_$w
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d /tmp/pmc_vb_$w -- \
    python $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-north-star --no-end-to-end --no-stage-clocks 2>&1 | grep -v "amdgpu.ids\|simple_timer" | tail -3
ls /tmp/pmc_vb_$w/*/ 
cp /tmp/pmc_vb_$w/*/*counter_collection.csv $out/counters_$w.csv 2>/dev/null
cp /tmp/pmc_vb_$w/*/*kernel_trace.csv $out/kernel_trace_$w.csv 2>/dev/null
cd $R && python tools/valu_busy.py $out/counters_$w.csv $out/kernel_trace_$w.csv > $out/valu_busy_$w.txt 2>&1; head -24 $out/valu_busy_$w.txt
