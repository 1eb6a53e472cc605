#!/bin/bash
# tools/r5_ab.sh "<ENV=val ...>" "<ENV=val ...>" ...: ms per default training step under each environment setting, alternating, two rounds, on one box
O=gpurun_out/r5ab; mkdir -p $O
for r in 1 2; do for cfg in "$@"; do
  env $cfg python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-forward --no-roofline 2>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['value'])" || tail -3 $O/err.log
done; done | tee -a $O/ab.txt
