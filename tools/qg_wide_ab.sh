#!/bin/bash
# A/B of constraint-kernel code generation choices.  HERE:  tools/qg_wide_ab.sh build <name> [ENV=val ...]  regenerates the kernels with the
# generator's environment knobs (QG_WIDE_PARTS, QG_DEPTH, QG_WIDE_OTHER ...), compiles the parts and links
# tools/_build/variants/<name>/libsandstorm_hip.so with the library's other objects (then regenerate the default: python tools/gen_quotient.py).
# GPU box:  tools/qg_wide_ab.sh run [names...]  -> quotient stage ms of both layouts per variant, into gpurun_out/qg_wide_ab/
set -e
cd "$(dirname "$0")/.."
V=tools/_build/variants
PARTS="quotient_gen_starknet_p0 quotient_gen_starknet_p1 quotient_gen_starknet_p2 quotient_gen_starknet_p3 quotient_gen_starknet_p4 quotient_gen_starknet_p5 quotient_gen_recursive_p0"
if [ "$1" = build ]; then
    name=$2; shift 2
    mkdir -p $V/$name
    env "$@" python tools/gen_quotient.py | grep -o "[0-9]* as terms of [0-9]* constraints" | tr '\n' ';'; echo
    for f in $PARTS; do ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value -Wno-pass-failed -Iinclude \
        -c sandstorm_amd/csrc/$f.hip -o $V/$name/$f.o ) & done; wait
    objs=$(ls sandstorm_amd/_build/*.o | grep -v 'quotient_gen_.*_p[0-9]*\.o$')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/$name/libsandstorm_hip.so $V/$name/*.o $objs -Wl,-rpath,/opt/rocm/lib
    python tools/count_insts.py $V/$name/quotient_gen_*.o | sort
    rm $V/$name/*.o
    echo "built $name ($@)"
    exit 0
fi
if [ "$1" = prof ]; then          # per-kernel times of every variant: the parts are separate kernels, the best configuration is chosen per part
    shift; R=$(pwd); mkdir -p gpurun_out/qg_wide_ab
    cp sandstorm_amd/_build/libsandstorm_hip.so /tmp/libsandstorm_hip.orig.so
    names="$@"; [ -z "$names" ] && names=$(ls $V)
    for name in $names; do
        cp $V/$name/libsandstorm_hip.so sandstorm_amd/_build/libsandstorm_hip.so
        for w in starknet_2p20 recursive_2p20; do
            rm -rf /tmp/qgp; ( cd /tmp && TMPDIR=/tmp timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qgp -- python $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-north-star --no-end-to-end > /dev/null 2>&1 )
            python - "$name" $w <<'PY' | tee -a gpurun_out/qg_wide_ab/per_part.txt
import csv, glob, sys
f = glob.glob("/tmp/qgp/*/*kernel_stats.csv")
rows = [r for r in csv.DictReader(open(f[0])) if "quotient_" in r["Name"]] if f else []
print(sys.argv[1], sys.argv[2], " ".join("%s=%.3f" % (r["Name"].split("quotient_")[1].split("_kernel")[0], float(r["AverageNs"]) / 1e6) for r in sorted(rows, key=lambda r: r["Name"])))
PY
        done
    done
    cp /tmp/libsandstorm_hip.orig.so sandstorm_amd/_build/libsandstorm_hip.so
    exit 0
fi
shift || true
mkdir -p gpurun_out/qg_wide_ab
cp sandstorm_amd/_build/libsandstorm_hip.so /tmp/libsandstorm_hip.orig.so
names="$@"; [ -z "$names" ] && names=$(ls $V)
for name in $names; do
    cp $V/$name/libsandstorm_hip.so sandstorm_amd/_build/libsandstorm_hip.so
    for w in starknet_2p20 recursive_2p20; do
        timeout 160 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-north-star --no-end-to-end > gpurun_out/qg_wide_ab/bench_${name}_$w.json 2>/dev/null
        python -c "
import json
d=json.load(open('gpurun_out/qg_wide_ab/bench_${name}_$w.json')); print('$name', '$w', round(d['value'],4), 'quotient', round(d['stage_ms_per_proof']['quotient'],2))" | tee -a gpurun_out/qg_wide_ab/results.txt
    done
done
cp /tmp/libsandstorm_hip.orig.so sandstorm_amd/_build/libsandstorm_hip.so
