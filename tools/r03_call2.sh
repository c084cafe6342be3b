#!/bin/bash
# round 3, second GPU call: the constraint kernels' variants side by side (library built with make QG_AB=1), Pedersen window
# widths, then the bench line with the winners
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03_call2; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/qg_bench.py starknet 20 > $O/qg_starknet.json 2> $O/qg_starknet.err; echo "qg starknet rc=$?"; cat $O/qg_starknet.json
timeout 300 python tools/qg_bench.py recursive 20 > $O/qg_recursive.json 2> $O/qg_recursive.err; echo "qg recursive rc=$?"; cat $O/qg_recursive.json
BS=$(python -c "import json;print(json.load(open('$O/qg_starknet.json'))['best_variant'])")
BR=$(python -c "import json;print(json.load(open('$O/qg_recursive.json'))['best_variant'])")
echo "best variants: starknet $BS recursive $BR"
( timeout 600 python -m pytest tests/test_gpu_real_quotient.py tests/test_gpu_parity.py -m gpu -x -q -k "quotient or pedersen or friendly or merkle" ) > $O/pytest_quotient.txt 2>&1; tail -3 $O/pytest_quotient.txt
for W in 16 18 20; do
  SS_PED_WINDOW=$W SS_QG_VARIANT_RECURSIVE=$BR timeout 200 python bench.py --workload recursive_2p20 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_rec_w$W.json 2> $O/bench_rec_w$W.err
  python -c "
import json; d=json.load(open('$O/bench_rec_w$W.json')); print('recursive_2p20 W=$W', round(d['value'],4), d['stage_ms_per_proof'])"
  SS_PED_WINDOW=$W timeout 100 python bench.py --workload array_sum_example --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_ex_w$W.json 2> $O/bench_ex_w$W.err
  python -c "
import json; d=json.load(open('$O/bench_ex_w$W.json')); print('array_sum_example W=$W', round(d['value'],4), d['stage_ms_per_proof'])"
  SS_PED_WINDOW=$W timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pedersen or friendly" 2>&1 | tail -1
done
SS_QG_VARIANT_STARKNET=$BS timeout 300 python bench.py --workload starknet_2p20 --steps 5 --warmup 1 --no-cpu-baseline --no-north-star > $O/bench_sn_best.json 2> $O/bench_sn_best.err
python -c "
import json; d=json.load(open('$O/bench_sn_best.json')); print('starknet_2p20 best', round(d['value'],4), d['stage_ms_per_proof'])"
echo done
