#!/bin/bash
# careful same-box A/B of environment settings on the default training step: ROUNDS alternating rounds of 40 timed steps each
#   ABCFGS="X=1 MRFA_PROLOGUE_FUSION=0" ROUNDS=3 bash tools/r6_ab_long.sh <tag>
O=gpurun_out/$1; mkdir -p $O
for r in $(seq ${ROUNDS:-3}); do for cfg in ${ABCFGS}; do
  env $cfg python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-forward --no-roofline 2>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['value'])" || tail -3 $O/err.log
done; done | tee $O/ab_long.txt
python - <<PY
import collections
d=collections.defaultdict(list)
for ln in open("$O/ab_long.txt"):
    k,ms,v=ln.split(); d[k].append(float(ms))
for k,v in d.items(): print(k, "mean %.2f ms  min %.2f  max %.2f  (%d runs)"%(sum(v)/len(v), min(v), max(v), len(v)))
PY
