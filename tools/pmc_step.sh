#!/bin/bash
# One PMC pass per counter over an eager training step (kernel-trace only), per-kernel means -> gpurun_out/pmc_step/<COUNTER>.csv
#   bash tools/pmc_step.sh TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD ...
set -u; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/pmc_step; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  rm -rf /tmp/ps_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/ps_$c -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-forward --no-roofline --no-graph > /tmp/ps_$c.log 2>&1
  f=$(find /tmp/ps_$c -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $R/tools/pmc_per_kernel.py $f $O/$c.csv; else echo "$c: no output"; tail -3 /tmp/ps_$c.log; fi
done
