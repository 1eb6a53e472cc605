#!/bin/bash
# One PMC pass per counter (kernel-trace only, one counter per pass) over ONE eager pass of a bench.py configuration, per-kernel means
#   bash tools/pmc_bench.sh <outdir> "<bench.py args>" COUNTER [COUNTER ...]        -> <outdir>/<COUNTER>.csv
set -u; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$1; ARGS=$2; shift; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  rm -rf /tmp/pb_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pb_$c -o p -- python3 $R/bench.py $ARGS > /tmp/pb_$c.log 2>&1
  f=$(find /tmp/pb_$c -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $R/tools/pmc_per_kernel.py $f $O/$c.csv; else echo "$c: no output"; tail -3 /tmp/pb_$c.log; fi
done
