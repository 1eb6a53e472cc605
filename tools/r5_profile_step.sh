#!/bin/bash
# rocprofv3 kernel trace of the default bench (hipGraph replays) -> kernel statistics of the whole run, per-step kernel table of the last 10 replayed steps,
# every launch of the last step.  $1 = tag, rest = env assignments.  Output: gpurun_out/r5prof/
set -u; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; tag=$1; shift; O=$R/gpurun_out/r5prof; mkdir -p $O
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_$tag
rocprofv3 --kernel-trace --stats -d /tmp/p_$tag -o rp -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-forward --no-roofline > $O/prof_$tag.log 2>&1
DB=$(find /tmp/p_$tag -name "*.db" | head -1)
python3 $R/tools/rocprof_summary.py $DB $O/${tag}_bench_b8_kernel_stats.csv >> $O/prof_$tag.log 2>&1
python3 $R/tools/rocprof_replay_window.py $DB $O/${tag}_replay_per_step.csv 10 >> $O/prof_$tag.log 2>&1
python3 $R/tools/rocprof_step_list.py $DB > $O/${tag}_step_list.txt 2>>$O/prof_$tag.log
python3 $R/tools/step_timeline.py $DB 2 > $O/${tag}_step_timeline.txt 2>&1
tail -3 $O/prof_$tag.log
