#!/bin/bash
# Round-3 step profile on one MI355X box (gpurun): rocprofv3 kernel trace of the default bench command, per-kernel totals and the
# steady-state (last 10 replayed steps) table.  Output: gpurun_out/r3prof/ (copied to profiles/r3_* afterwards).
set -u; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r3prof; mkdir -p $O
TAG=${1:-r3}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_def
rocprofv3 --kernel-trace -d /tmp/p_def -o rp -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-forward --no-roofline > $O/${TAG}_prof_default.log 2>&1
DB=$(find /tmp/p_def -name "*.db" | head -1)
python3 $R/tools/rocprof_summary.py $DB $O/${TAG}_bench_b8_kernel_stats.csv >> $O/${TAG}_prof_default.log 2>&1
python3 $R/tools/rocprof_replay_window.py $DB $O/${TAG}_replay_per_step.csv 10 >> $O/${TAG}_prof_default.log 2>&1
tail -3 $O/${TAG}_prof_default.log
head -40 $O/${TAG}_replay_per_step.csv
