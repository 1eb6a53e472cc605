#!/bin/bash
# Round-5 evidence run on one MI355X box (gpurun): tests, bench lines of every configuration, rocprofv3 kernel statistics (whole run + steady-state
# window + every launch of one step), in-graph phase times and the PMC passes behind profiles/r5_pmc_summary.json.  Everything lands in gpurun_out/r5final/
# (copied to profiles/ afterwards).   bash tools/r5_final_profiles.sh [part ...]   parts: tests bench prof pmc extra (default: all)
set -u; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r5final; mkdir -p $O
PARTS=${*:-tests bench prof pmc extra}
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
for f in default fomm refloss config4 config5 syncbn_graph_one_rank force_exchange_one_rank; do tail -1 $O/bench_$f.log | cut -c1-220; done
fi
if has prof; then
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_def
rocprofv3 --kernel-trace --stats -d /tmp/p_def -o rp -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-forward --no-roofline > $O/prof_default.log 2>&1
DB=$(find /tmp/p_def -name "*.db" | head -1)
python3 $R/tools/rocprof_summary.py $DB $O/r5_final_bench_b8_kernel_stats.csv >> $O/prof_default.log 2>&1
python3 $R/tools/rocprof_replay_window.py $DB $O/r5_final_replay_per_step.csv 10 >> $O/prof_default.log 2>&1
python3 $R/tools/rocprof_step_list.py $DB > $O/r5_final_step_launches.txt 2>>$O/prof_default.log
python3 $R/tools/step_timeline.py $DB 2 > $O/r5_final_step_timeline.txt 2>&1
tail -2 $O/prof_default.log
cd $R
python tools/step_phases.py 8 mtia 20 2>/dev/null | grep -v amdgpu > $O/step_phases.txt
python tools/step_phases.py 8 fomm 20 2>/dev/null | grep -v amdgpu > $O/step_phases_fomm.txt
python tools/profile_step.py 8 mtia 200 2>/dev/null | grep -v amdgpu > $O/profile_step_mtia.txt
fi
if has pmc; then
rm -rf $R/gpurun_out/pmc_step
bash tools/pmc_step.sh FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT > $O/pmc.log 2>&1
mkdir -p $O/pmc; cp $R/gpurun_out/pmc_step/*.csv $O/pmc/ 2>/dev/null
python tools/pmc_derive.py $R/gpurun_out/pmc_step $O/r5_pmc_summary.json $O/r5_traffic.json >> $O/pmc.log 2>&1; tail -2 $O/pmc.log
fi
if has extra; then
python tools/headline_probe.py 3 2>/dev/null | grep -v amdgpu > $O/headline_probe.txt
for k in "MRFA_FUSED_SPLITK=0" "MRFA_BN_FIN_FUSED=0" "MRFA_BN_BWD_IN_DGRAD=0" "MRFA_PACK_STREAM=0" "MRFA_BRANCH_STREAMS=0" "MRFA_DEFER_WGRADS=0" "X=0"; do echo "$k $(env $k python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-forward --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")"; done > $O/switch_ablation.txt 2>&1
cat $O/switch_ablation.txt
python tools/soak_train.py 300 > $O/soak_300_steps.log 2>&1; tail -2 $O/soak_300_steps.log
fi
ls $O
