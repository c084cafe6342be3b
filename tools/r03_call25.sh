#!/bin/bash
# the round's last code: the default bench line as the driver runs it, and the kernel trace of the same command
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_call25; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['stage_ms_per_proof'], d['roofline']['frac'], d['roofline_dominant']['kernel'], d['roofline_dominant']['frac'], d['north_star']['value'], d['cpu_baseline']['measured_sample_s'])"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python /root/repo/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-north-star > /root/repo/$O/kt.json 2> /root/repo/$O/kt.err
cd /root/repo
f=$(find /tmp/$O/kt $O/kt -name 'kt_kernel_trace.csv' 2>/dev/null | head -1); python tools/trace_summary.py $f > $O/kernel_trace_summary_starknet_2p20.txt
python tools/trace_gaps.py $f 1 --anchor=pow_ > $O/gaps_starknet_2p20.txt 2>&1
head -24 $O/kernel_trace_summary_starknet_2p20.txt | cut -c1-150
rm -rf $O/kt /tmp/$O
echo done
