#!/bin/bash
# tools/ab_multi.sh "<tag> <tag> ..." [bench_kernels args]: the in-tree library and each tools/ab_lib.sh build, twice, on one box
tags=$1; shift
for r in 1 2; do
for v in main $tags; do
  if [ $v = main ]; then unset MRFA_HIP_LIB; else export MRFA_HIP_LIB=$PWD/mrfa_amd/_lib/ab_$v/libmrfa_hip.so; fi
  python tools/bench_kernels.py --mfma 1 --iters 20 "$@" | grep -v "^layer\|conv2 \|convo2 " | sed "s/^/$v: /"
done; done
