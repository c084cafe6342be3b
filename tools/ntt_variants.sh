#!/bin/bash
# scratch experiment driver (GPU box): rebuild ntt.o with different settings, check parity, time the batch LDE
cd /root/repo
for v in "" "-DSS_NTT_GMAX_DIF=3 -DSS_NTT_THREADS_DIF=256 -DSS_NTT_OCC_DIF=2" "-DSS_NTT_GMAX=3 -DSS_NTT_THREADS=256 -DSS_NTT_OCC=2 -DSS_NTT_GMAX_DIF=3 -DSS_NTT_THREADS_DIF=256 -DSS_NTT_OCC_DIF=2"; do
  rm -f sandstorm_amd/_build/ntt.o
  make -C sandstorm_amd/csrc EXTRA="$v" >/dev/null 2>&1 || { echo "build fail $v"; continue; }
  echo "== $v"
  timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "ntt or lde" 2>&1 | tail -1
  python tools/ntt_bench.py 24 9 2
  python tools/ntt_bench.py 20 10 5
done
rm -f sandstorm_amd/_build/ntt.o; make -C sandstorm_amd/csrc >/dev/null 2>&1
