#!/bin/bash
# scratch experiment driver (GPU box): NTT occupancy / tile-size experiments
cd /root/repo
echo "== default (2 WG/CU)"; python tools/ntt_bench.py 24 9 3 2>&1 | grep -v amdgpu

for v in "-DSS_NTT_LOG_TILE=10 -DSS_NTT_THREADS=256 -DSS_NTT_OCC=4 -DSS_NTT_GMAX_DIF=2 -DSS_NTT_THREADS_DIF=256 -DSS_NTT_OCC_DIF=4" "-DSS_NTT_LOG_TILE=10 -DSS_NTT_THREADS=256 -DSS_NTT_OCC=4 -DSS_NTT_GMAX_DIF=3 -DSS_NTT_THREADS_DIF=128 -DSS_NTT_OCC_DIF=2"; do
  rm -f sandstorm_amd/_build/ntt.o sandstorm_amd/_build/capi.o
  make -C sandstorm_amd/csrc EXTRA="$v" >/dev/null 2>&1 || { echo "build fail $v"; continue; }
  echo "== $v"
  timeout 200 python -m pytest tests/test_gpu_parity.py -q -x -k "ntt_golden or ntt_orders or lde_vs_oracle" 2>&1 | tail -1
  python tools/ntt_bench.py 24 9 3 2>&1 | grep -v amdgpu
  python tools/ntt_bench.py 20 10 5 2>&1 | grep -v amdgpu
done
rm -f sandstorm_amd/_build/ntt.o sandstorm_amd/_build/capi.o; make -C sandstorm_amd/csrc >/dev/null 2>&1
