#!/bin/bash
# A/B of transform-kernel variants.  HERE (no GPU):  tools/ntt_ab.sh build name1:"-DFLAG ..." name2:"..."   links
# tools/_build/variants/<name>/libsandstorm_hip.so from a variant ntt.o + the library's other objects.
# On the GPU box:  tools/ntt_ab.sh run [names...]   swaps each variant in (the box's copy of the repo is scratch) and times the
# batch LDE (tools/ntt_bench.py); results into gpurun_out/ntt_ab/.
set -e
cd "$(dirname "$0")/.."
V=tools/_build/variants
if [ "$1" = build ]; then
    shift
    make -s -C sandstorm_amd/csrc
    for spec in "$@"; do
        name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
        mkdir -p $V/$name
        /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value -Wno-pass-failed -Iinclude $flags \
            -c sandstorm_amd/csrc/ntt.hip -o $V/$name/ntt.o
        objs=$(ls sandstorm_amd/_build/*.o | grep -v '/ntt.o$')
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/$name/libsandstorm_hip.so $V/$name/ntt.o $objs -Wl,-rpath,/opt/rocm/lib
        rm $V/$name/ntt.o
        echo "built $name ($flags)"
    done
    exit 0
fi
shift || true
mkdir -p gpurun_out/ntt_ab
cp sandstorm_amd/_build/libsandstorm_hip.so /tmp/libsandstorm_hip.orig.so
names="$@"; [ -z "$names" ] && names=$(ls $V)
for name in $names; do
    cp $V/$name/libsandstorm_hip.so sandstorm_amd/_build/libsandstorm_hip.so
    for rep in 1 2; do
        echo -n "$name: " | tee -a gpurun_out/ntt_ab/results.txt
        timeout 120 python tools/ntt_bench.py 24 9 5 2>&1 | tail -1 | tee -a gpurun_out/ntt_ab/results.txt
    done
done
cp /tmp/libsandstorm_hip.orig.so sandstorm_amd/_build/libsandstorm_hip.so
