#!/bin/bash
# scratch experiment driver (GPU box): rebuild ntt.o with different group/threads settings and bench
cd /root/repo
for v in "2 512 4" "3 256 2" "2 256 2" "2 256 4" "2 512 2" "3 512 2"; do
  set -- $v
  rm -f sandstorm_amd/_build/ntt.o
  make -C sandstorm_amd/csrc EXTRA="-DSS_NTT_GMAX=$1 -DSS_NTT_THREADS=$2 -DSS_NTT_OCC=$3" >/dev/null 2>&1 || { echo "build fail $v"; continue; }
  echo "== GMAX=$1 THREADS=$2 OCC=$3"
  python -m pytest tests/test_gpu_parity.py -q -x -k "ntt or lde" 2>&1 | tail -1
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['stage_ms_per_proof'].get('ntt_pass'), d['ntt_gfield_ops_per_s'])"
done
