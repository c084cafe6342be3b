#!/bin/bash
# usage: tools/pmc_run.sh <out-name> "<counters>" <command...>   (GPU box; one --pmc pass, csv into gpurun_out/)
name=$1; counters=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$name
timeout 240 rocprofv3 --pmc $counters --output-format csv -d /tmp/pmc_$name -- "$@" 2>&1 | grep -v "amdgpu.ids\|simple_timer" | tail -4
f=$(ls /tmp/pmc_$name/*/*counter_collection.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" /root/repo/gpurun_out/pmc_$name.csv; python /root/repo/tools/pmc_summary.py "$f"; else echo "no counter file"; fi
