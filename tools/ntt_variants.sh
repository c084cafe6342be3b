#!/bin/bash
# scratch experiment driver (GPU box): rebuild ntt.o with timing ablations and time the batch LDE
cd /root/repo
for v in "" "-DSS_NTT_ABL_NOBAR" "-DSS_NTT_ABL_NOLDS -DSS_NTT_ABL_NOBAR" "-DSS_NTT_ABL_NOGL -DSS_NTT_ABL_NOTW" "-DSS_NTT_ABL_NOGL -DSS_NTT_ABL_NOTW -DSS_NTT_ABL_NOLDS -DSS_NTT_ABL_NOBAR" "-DSS_NTT_ABL_NOMUL -DSS_NTT_ABL_NOTW -DSS_NTT_ABL_NOLDS -DSS_NTT_ABL_NOBAR" "-DSS_NTT_ABL_NOMUL -DSS_NTT_ABL_NOTW -DSS_NTT_ABL_NOGL"; do
  rm -f sandstorm_amd/_build/ntt.o
  make -C sandstorm_amd/csrc EXTRA="$v" >/dev/null 2>&1 || { echo "build fail $v"; continue; }
  echo "== $v"
  python tools/ntt_bench.py 24 9 2 2>&1 | grep -v amdgpu
done
rm -f sandstorm_amd/_build/ntt.o; make -C sandstorm_amd/csrc >/dev/null 2>&1
