#!/bin/bash
# scratch experiment driver (GPU box): quotient VM occupancy / ring-depth variants
cd /root/repo
for v in "4 4" "3 4" "3 6" "2 8"; do
  set -- $v
  rm -f sandstorm_amd/_build/quotient.o sandstorm_amd/_build/capi.o
  make -C sandstorm_amd/csrc EXTRA="-DSS_VM_OCC=$1 -DSS_VM_DEPTH=$2" >/dev/null 2>&1 || { echo "build fail $v"; continue; }
  echo "== SS_VM_OCC=$1 SS_VM_DEPTH=$2"
  timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "quotient" 2>&1 | tail -1
  timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['stage_ms_per_proof'])"
done
rm -f sandstorm_amd/_build/quotient.o sandstorm_amd/_build/capi.o; make -C sandstorm_amd/csrc >/dev/null 2>&1
