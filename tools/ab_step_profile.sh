#!/bin/bash
# tools/ab_step_profile.sh "<bench args A>" "<bench args B>": per-kernel ms per step of the replayed training step under rocprofv3 for two bench.py settings
# (e.g. "--tune conv_halo_bn64_fill=1" "--tune conv_halo_bn64_fill=0"); writes gpurun_out/ab_step_{A,B}.csv
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for args in "$1" "$2"; do
  tag=$([ $i = 0 ] && echo A || echo B); i=1
  rm -rf /tmp/p_$tag
  rocprofv3 --kernel-trace -d /tmp/p_$tag -o rp -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-forward --no-roofline $args > /tmp/p_$tag.log 2>&1
  DB=$(find /tmp/p_$tag -name "*.db" | head -1)
  python3 $R/tools/rocprof_replay_window.py $DB $R/gpurun_out/ab_step_$tag.csv 6 | tail -1
done
