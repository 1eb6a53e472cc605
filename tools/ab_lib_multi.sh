#!/bin/bash
# tools/ab_lib.sh for a flag that touches several sources:  tools/ab_lib_multi.sh "conv_small norm tokenpose" "-DMRFA_AB_X" tag
#   -> mrfa_amd/_lib/ab_<tag>/libmrfa_hip.so (the listed sources recompiled under the extra flags, every other object from the in-tree build)
set -e
srcs=$1; flags=$2; tag=$3
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/mrfa_amd/_lib/ab_$tag
mkdir -p $out
for src in $srcs; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$root/include -I$root/mrfa_amd/csrc -Wno-unused-result $flags -c $root/mrfa_amd/csrc/$src.hip -o $out/$src.o &
done
wait
objs=""
for o in $root/mrfa_amd/_lib/*.o; do b=$(basename $o .o); if [ -f $out/$b.o ]; then objs="$objs $out/$b.o"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libmrfa_hip.so $objs
echo built $out/libmrfa_hip.so
