#!/bin/bash
# scratch experiment driver (GPU box): rebuild ntt.o with different settings and time the batch LDE
cd /root/repo
for v in "" "-DSS_NTT_ABL_NOTW" "-DSS_NTT_ABL_NOMUL" "-DSS_NTT_ABL_NOTW -DSS_NTT_ABL_NOMUL" "-DSS_NTT_GMAX=3 -DSS_NTT_THREADS=256 -DSS_NTT_OCC=2 -DSS_NTT_ABL_NOTW"; do
  rm -f sandstorm_amd/_build/ntt.o
  make -C sandstorm_amd/csrc EXTRA="$v" >/dev/null 2>&1 || { echo "build fail $v"; continue; }
  echo "== $v"
  python tools/ntt_bench.py 24 9 2
  python tools/ntt_bench.py 20 10 5
done
rm -f sandstorm_amd/_build/ntt.o; make -C sandstorm_amd/csrc >/dev/null 2>&1
