#!/bin/bash
# Build a second copy of the library with one source recompiled under extra flags, for same-box kernel A/B runs:
#   tools/ab_lib.sh conv_halo "-DMRFA_AB_NOPIN" nopin   ->  mrfa_amd/_lib/ab_nopin/libmrfa_hip.so
#   MRFA_HIP_LIB=mrfa_amd/_lib/ab_nopin/libmrfa_hip.so python tools/bench_kernels.py ...
set -e
src=$1; flags=$2; tag=$3
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/mrfa_amd/_lib/ab_$tag
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$root/include -I$root/mrfa_amd/csrc -Wno-unused-result $flags -c $root/mrfa_amd/csrc/$src.hip -o $out/$src.o
objs=""
for o in $root/mrfa_amd/_lib/*.o; do b=$(basename $o .o); if [ "$b" = "$src" ]; then objs="$objs $out/$src.o"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libmrfa_hip.so $objs
echo built $out/libmrfa_hip.so
