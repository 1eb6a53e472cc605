#!/bin/bash
# Round-6 GPU trips (gpurun): bash tools/r6_gpu.sh <tag> [part ...]   parts: leantests tests lean ab phases prof pmc
#   leantests  conv_lean.hip + every convolution / finalize kernel test      tests   the whole -m gpu suite + smoke
#   lean       tools/ubench/lean_bench (per-launch times against conv_small / conv_halo)
#   ab         the default training step with MRFA_CONV_LEAN=1 / 0, alternating, two rounds
#   phases     in-graph phase times of the replayed step                      prof    rocprofv3 kernel statistics of the step (whole run, per-step window, launches)
set -u; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O
PARTS=${*:-leantests lean ab phases}
cd $R
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has leantests; then
python -m pytest tests/test_kernels_gpu.py -x -q -k "lean or conv2d or fused_finalize or statistic or patch_tiled or wgrad" 2>&1 | tail -15 > $O/pytest_conv.txt; tail -3 $O/pytest_conv.txt
fi
if has tests; then
python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1; grep -E "passed|failed" $O/gputest.log | tail -1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
fi
if has lean; then
NONLY=16 ./tools/ubench/bin/lean_bench > $O/lean_bench.txt 2>&1; cat $O/lean_bench.txt
for g in 3 4 6; do NONLY=16 GEO=$g ONLY_LEAN=1 ./tools/ubench/bin/lean_bench 2>&1 | grep -v " -1.00" | grep -v "^shape" | cut -c1-52; done | tee $O/lean_bench_one_tile_per_wave.txt
fi
if has gemm; then
GEMM=1 ./tools/ubench/bin/lean_bench > $O/gemm_lean_bench.txt 2>&1; cat $O/gemm_lean_bench.txt
fi
if has ab; then
# ABCFGS="ENV=val ENV=val ..." (space-separated single settings; default: conv_lean on / off)
for r in 1 2; do for cfg in ${ABCFGS:-MRFA_CONV_LEAN=1 MRFA_CONV_LEAN=0}; do
  env $cfg python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-forward --no-roofline 2>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['value'])" || tail -3 $O/err.log
done; done | tee $O/ab.txt
fi
if has phases; then
python tools/step_phases.py 8 mtia 20 2>/dev/null | grep -v amdgpu > $O/step_phases.txt; cat $O/step_phases.txt
fi
if has probe; then
python tools/headline_probe.py 2 _head 2>/dev/null | grep -v amdgpu > $O/headline_probe_bf16x6.txt
MRFA_MFMA=f32 python tools/headline_probe.py 2 _head 2>/dev/null | grep -v amdgpu > $O/headline_probe_f32.txt
grep -A1 "kp_head  \|kp_img_head  " $O/headline_probe_bf16x6.txt $O/headline_probe_f32.txt | cut -c1-200
fi
if has prof; then
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_def
rocprofv3 --kernel-trace --stats -d /tmp/p_def -o rp -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-forward --no-roofline > $O/prof_default.log 2>&1
DB=$(find /tmp/p_def -name "*.db" | head -1)
python3 $R/tools/rocprof_summary.py $DB $O/bench_b8_kernel_stats.csv >> $O/prof_default.log 2>&1
python3 $R/tools/rocprof_replay_window.py $DB $O/replay_per_step.csv 10 >> $O/prof_default.log 2>&1
python3 $R/tools/rocprof_step_list.py $DB > $O/step_launches.txt 2>>$O/prof_default.log
tail -2 $O/prof_default.log
cd $R
python tools/profile_step.py 8 mtia 200 2>/dev/null | grep -v amdgpu > $O/profile_step_mtia.txt
fi
ls $O
