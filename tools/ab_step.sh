#!/bin/bash
# tools/ab_step.sh "<tag> <tag> ..." [rounds] [bench args]: ms per training step of the in-tree library and of each tools/ab_lib*.sh build, alternating on one box
tags=$1; rounds=${2:-2}; shift; shift
for r in $(seq $rounds); do
for v in main $tags; do
  if [ $v = main ]; then unset MRFA_HIP_LIB; else export MRFA_HIP_LIB=$PWD/mrfa_amd/_lib/ab_$v/libmrfa_hip.so; fi
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-forward --no-roofline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'])"
done; done
