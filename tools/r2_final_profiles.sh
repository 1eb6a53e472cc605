#!/bin/bash
# Round-2 evidence run on one MI355X box (gpurun): tests, bench lines of every configuration, rocprofv3 kernel statistics and PMC
# traffic passes.  Everything lands in gpurun_out/r2final/ (copied to profiles/ afterwards).
set -u; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r2final; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; tail -2 $O/gputest.log
python bench.py --steps 10 --warmup 3 > $O/bench_default.log 2>&1
python bench.py --steps 10 --warmup 3 --prior fomm --no-cpu-baseline > $O/bench_fomm.log 2>&1
python bench.py --steps 10 --warmup 3 --loss reference --no-cpu-baseline --no-forward > $O/bench_refloss.log 2>&1
python bench.py --steps 10 --warmup 3 --background --mfma bf16 --batch 16 --no-cpu-baseline --no-forward > $O/bench_config4.log 2>&1
python bench.py --size 512 --batch 4 --inference --steps 20 --warmup 3 > $O/bench_config5.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/p_def -o rp -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-forward --no-roofline > $O/prof_default.log 2>&1
DB=$(find /tmp/p_def -name "*.db" | head -1)
python3 $R/tools/rocprof_summary.py $DB $O/r2_final_bench_b8_kernel_stats.csv >> $O/prof_default.log 2>&1
python3 $R/tools/rocprof_replay_window.py $DB $O/r2_final_replay_per_step.csv 10 >> $O/prof_default.log 2>&1
rocprofv3 --kernel-trace -d /tmp/p_c5 -o rp -- python3 $R/bench.py --size 512 --batch 4 --inference --steps 20 --warmup 3 --no-cpu-baseline > $O/prof_config5.log 2>&1
DB5=$(find /tmp/p_c5 -name "*.db" | head -1)
python3 $R/tools/rocprof_summary.py $DB5 $O/r2_config5_kernel_stats.csv >> $O/prof_config5.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc5_$c -o p -- python3 $R/bench.py --size 512 --batch 4 --inference --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $O/pmc_config5_$c.log 2>&1
  python3 $R/tools/pmc_per_kernel.py $(find /tmp/pmc5_$c -name "*counter_collection.csv" | head -1) $O/r2_config5_pmc_${c}_per_kernel.csv >> $O/pmc_config5_$c.log 2>&1
done
ls -la $O
