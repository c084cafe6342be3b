#!/bin/bash
# TEST INFRASTRUCTURE ONLY: sandstorm_amd/csrc/*.hip compiled for the HOST over tests/hipemu/hip/hip_runtime.h
#   -> tests/hipemu/_build/libsandstorm_hipemu.so (same C ABI as the product's library; loaded by tests/test_device_code_on_host.py only)
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd)
CXX=${HIPEMU_CXX:-/opt/rocm/lib/llvm/bin/clang++}          # clang: the sources use ext_vector_type / address_space attributes
OUT=${HIPEMU_OUT:-$HERE/_build}; mkdir -p $OUT
# HIPEMU_SANITIZE=address: an AddressSanitizer build (out-of-bounds reads / writes of "device" buffers and LDS arrays inside kernels);
# run with LD_PRELOAD=$($CXX -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0
SAN=${HIPEMU_SANITIZE:+-fsanitize=$HIPEMU_SANITIZE -fno-omit-frame-pointer -g1}
FLAGS="-x c++ -std=c++17 -O2 -fPIC -w $SAN -I $HERE -I $ROOT/include -I $ROOT/sandstorm_amd/csrc"
# the generated constraint kernels' translation units: csrc/quotient_gen_sources.mk (tools/gen_quotient.py)
QG=$(sed -n 's/^QG_SRCS := //p' $ROOT/sandstorm_amd/csrc/quotient_gen_sources.mk | sed 's/\.hip//g')
SRCS="capi ntt hash pedersen fri deep quotient ext trace goldilocks $QG"
pids=()
for f in $SRCS; do
  src=$ROOT/sandstorm_amd/csrc/$f.hip; obj=$OUT/$f.o
  if [ ! -f $obj ] || [ -n "$(find $ROOT/sandstorm_amd/csrc $HERE/hip $ROOT/include -newer $obj \( -name '*.hip' -o -name '*.h' -o -name '*.inc' \) | head -1)" ]; then
    ( O=-O2; case $f in quotient_gen_*) O=-O1;; esac; $CXX $FLAGS $O -c $src -o $obj ) &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$CXX -std=c++17 -O2 -fPIC -w $SAN -I $HERE -c $HERE/hipemu.cpp -o $OUT/hipemu.o
$CXX -shared -fPIC -pthread $SAN ${HIPEMU_SANITIZE:+-shared-libsan} -o $OUT/libsandstorm_hipemu.so $OUT/hipemu.o $(for f in $SRCS; do echo $OUT/$f.o; done)
echo $OUT/libsandstorm_hipemu.so
