#!/bin/bash
# Build libsrlhip_<tag>.so next to the product library with extra compiler flags on the GEMM / convolution sources
# (timing experiments: leave-out switches such as -DSRL_GEMM3_DBG=...), reusing every other object of the normal build.
#   scripts/build_variant.sh noload -DSRL_GEMM3_DBG=1 ;  SRL_HIP_LIB=$PWD/srl_amd/csrc/libsrlhip_noload.so python3 scripts/gemm_bench.py conv
set -e
tag=$1; shift
cd "$(dirname "$0")/.."
make -j8 >/dev/null
d=srl_amd/csrc/build_$tag
mkdir -p $d
for f in gemm conv; do
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC "$@" -c -o $d/$f.o srl_amd/csrc/$f.hip &
done
wait
objs=$(ls srl_amd/csrc/build/*.o | grep -v -e /gemm.o -e /conv.o)
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -o srl_amd/csrc/libsrlhip_$tag.so $objs $d/gemm.o $d/conv.o
echo srl_amd/csrc/libsrlhip_$tag.so
