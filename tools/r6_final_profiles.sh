#!/bin/bash
# Round-6 evidence run on one MI355X box (gpurun): tests, bench lines of every configuration, rocprofv3 kernel statistics (whole run + steady-state
# window + every launch of one step), in-graph phase times, the PMC passes behind profiles/r6_pmc_summary.json / r6_traffic.json, and the PMC passes of
# config 4 (celebvhq bs=16 bf16) and config 5 (512^2 inference) of their own.  Everything lands in gpurun_out/r6final/ (copied to profiles/ afterwards).
#   bash tools/r6_final_profiles.sh [part ...]   parts: tests bench prof pmc pmc4 pmc5 extra (default: all)
set -u; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r6final; mkdir -p $O
PARTS=${*:-tests bench prof pmc pmc4 pmc5 extra}
cd $R
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has tests; then
python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; grep -E "passed|failed" $O/gputest.log | tail -1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
fi
if has bench; then
S0=$SECONDS; python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.log 2>$O/bench_default.err; echo "python bench.py --gpus 1 --steps 20 --warmup 5: $((SECONDS - S0)) s of wall time" > $O/bench_default.time
python bench.py --steps 10 --warmup 3 --prior fomm --no-cpu-baseline > $O/bench_fomm.log 2>/dev/null
python bench.py --steps 10 --warmup 3 --loss reference --no-cpu-baseline --no-forward > $O/bench_refloss.log 2>/dev/null
python bench.py --steps 10 --warmup 3 --background --mfma bf16 --batch 16 --no-cpu-baseline --no-forward > $O/bench_config4.log 2>/dev/null
python bench.py --size 512 --batch 4 --inference --steps 20 --warmup 3 > $O/bench_config5.log 2>/dev/null
MRFA_SYNCBN_GRAPH=1 MRFA_SYNCBN_FORCE_COLLECTIVE=1 python bench.py --sync-bn --force-exchange --steps 5 --warmup 2 --no-cpu-baseline --no-forward --no-roofline > $O/bench_syncbn_graph_one_rank.log 2>$O/bench_syncbn_graph_one_rank.err
python bench.py --force-exchange --steps 5 --warmup 2 --no-cpu-baseline --no-forward --no-roofline > $O/bench_force_exchange_one_rank.log 2>/dev/null
for f in default fomm refloss config4 config5 syncbn_graph_one_rank force_exchange_one_rank; do tail -1 $O/bench_$f.log | cut -c1-330; done
fi
if has prof; then
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_def
rocprofv3 --kernel-trace --stats -d /tmp/p_def -o rp -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-forward --no-roofline > $O/prof_default.log 2>&1
DB=$(find /tmp/p_def -name "*.db" | head -1)
python3 $R/tools/rocprof_summary.py $DB $O/r6_final_bench_b8_kernel_stats.csv >> $O/prof_default.log 2>&1
python3 $R/tools/rocprof_replay_window.py $DB $O/r6_final_replay_per_step.csv 10 >> $O/prof_default.log 2>&1
python3 $R/tools/rocprof_step_list.py $DB > $O/r6_final_step_launches.txt 2>>$O/prof_default.log
python3 $R/tools/step_timeline.py $DB 2 > $O/r6_final_step_timeline.txt 2>&1
tail -2 $O/prof_default.log
cd $R
python tools/step_phases.py 8 mtia 20 2>/dev/null | grep -v amdgpu > $O/step_phases.txt
python tools/step_phases.py 8 fomm 20 2>/dev/null | grep -v amdgpu > $O/step_phases_fomm.txt
python tools/profile_step.py 8 mtia 200 2>/dev/null | grep -v amdgpu > $O/profile_step_mtia.txt
NONLY=16 ./tools/ubench/bin/lean_bench > $O/lean_bench.txt 2>&1
fi
if has pmc; then
rm -rf $R/gpurun_out/pmc_step
bash tools/pmc_step.sh FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT > $O/pmc.log 2>&1
mkdir -p $O/pmc; cp $R/gpurun_out/pmc_step/*.csv $O/pmc/ 2>/dev/null
python tools/pmc_derive.py $R/gpurun_out/pmc_step $O/r6_pmc_summary.json $O/r6_traffic.json conv_halo_kernel wgrad_halo_kernel conv_bf16x6_kernel wgrad_bf16x6_kernel conv_lean_kernel wgrad_lean_kernel >> $O/pmc.log 2>&1; tail -3 $O/pmc.log
fi
if has pmc4; then
bash tools/pmc_bench.sh $O/pmc4 "--background --mfma bf16 --batch 16 --steps 1 --warmup 1 --no-cpu-baseline --no-forward --no-roofline --no-graph" FETCH_SIZE WRITE_SIZE > $O/pmc4.log 2>&1
python tools/pmc_traffic.py $O/pmc4 $O/r6_config4_traffic.json "config 4 (celebvhq.yaml wiring, bs=16, plain-bf16 products): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate --kernel-trace-only passes over bench.py --background --mfma bf16 --batch 16 --steps 1 --warmup 1 --no-cpu-baseline --no-forward --no-roofline --no-graph, per kernel by tools/pmc_per_kernel.py." config4 | cut -c1-300
fi
if has pmc5; then
bash tools/pmc_bench.sh $O/pmc5 "--size 512 --batch 4 --inference --steps 2 --warmup 1 --no-cpu-baseline --no-graph" FETCH_SIZE WRITE_SIZE > $O/pmc5.log 2>&1
python tools/pmc_traffic.py $O/pmc5 $O/r6_config5_traffic.json "config 5 (512x512 inference, B=4): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate --kernel-trace-only passes over bench.py --size 512 --batch 4 --inference --steps 2 --warmup 1 --no-cpu-baseline --no-graph, per kernel by tools/pmc_per_kernel.py.  One six-level feature-warp set = one <16> launch (64 ch @512^2) + one <32> (128 ch @256^2) + four <64> (256 / 512 ch @128^2 .. 16^2; the per-launch mean of <64> mixes the four levels)." config5 | cut -c1-300
fi
if has extra; then
python tools/headline_probe.py 3 _head 2>/dev/null | grep -v amdgpu > $O/headline_probe_bf16x6.txt
MRFA_MFMA=f32 python tools/headline_probe.py 3 _head 2>/dev/null | grep -v amdgpu > $O/headline_probe_f32.txt
for k in "MRFA_CONV_LEAN=0" "MRFA_WGRAD_LEAN=0" "MRFA_PROLOGUE_FUSION=1" "MRFA_BRANCH_STREAMS=0" "MRFA_DEFER_WGRADS=0" "X=0"; do echo "$k $(env $k python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-forward --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")"; done > $O/switch_ablation.txt 2>&1
cat $O/switch_ablation.txt
python tools/soak_train.py 300 > $O/soak_300_steps.log 2>&1; tail -2 $O/soak_300_steps.log
fi
ls $O
