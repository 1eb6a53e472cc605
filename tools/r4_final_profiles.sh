#!/bin/bash
# Round-4 evidence run on one MI355X box (gpurun): tests, bench lines of every configuration, rocprofv3 kernel statistics (whole run + steady-state
# window) and the PMC passes behind profiles/r4_pmc_summary.json.  Everything lands in gpurun_out/r4final/ (copied to profiles/ afterwards).
set -u; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r4final; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; grep -E "passed|failed" $O/gputest.log | tail -1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
S0=$SECONDS; python bench.py --steps 10 --warmup 3 > $O/bench_default.log 2>$O/bench_default.err; echo "python bench.py --steps 10 --warmup 3: $((SECONDS - S0)) s of wall time" > $O/bench_default.time
python bench.py --steps 10 --warmup 3 --prior fomm --no-cpu-baseline > $O/bench_fomm.log 2>/dev/null
python bench.py --steps 10 --warmup 3 --loss reference --no-cpu-baseline --no-forward > $O/bench_refloss.log 2>/dev/null
python bench.py --steps 10 --warmup 3 --background --mfma bf16 --batch 16 --no-cpu-baseline --no-forward > $O/bench_config4.log 2>/dev/null
python bench.py --size 512 --batch 4 --inference --steps 20 --warmup 3 > $O/bench_config5.log 2>/dev/null
MRFA_SYNCBN_GRAPH=1 MRFA_SYNCBN_FORCE_COLLECTIVE=1 python bench.py --sync-bn --force-exchange --steps 5 --warmup 2 --no-cpu-baseline --no-forward --no-roofline > $O/bench_syncbn_graph_one_rank.log 2>$O/bench_syncbn_graph_one_rank.err
python bench.py --force-exchange --steps 5 --warmup 2 --no-cpu-baseline --no-forward --no-roofline > $O/bench_force_exchange_one_rank.log 2>/dev/null
[ -x tools/ubench/bin/mfma_lds_mix ] || { mkdir -p tools/ubench/bin; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_lds_mix.hip -o tools/ubench/bin/mfma_lds_mix 2>/dev/null; }
./tools/ubench/bin/mfma_lds_mix 4 0 > $O/mfma_lds_mix.txt 2>/dev/null; ./tools/ubench/bin/mfma_lds_mix 4 1 >> $O/mfma_lds_mix.txt 2>/dev/null
for f in default fomm refloss config4 config5 syncbn_graph_one_rank force_exchange_one_rank wino; do tail -1 $O/bench_$f.log | cut -c1-220; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_def
rocprofv3 --kernel-trace -d /tmp/p_def -o rp -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-forward --no-roofline > $O/prof_default.log 2>&1
DB=$(find /tmp/p_def -name "*.db" | head -1)
python3 $R/tools/rocprof_summary.py $DB $O/r4_final_bench_b8_kernel_stats.csv >> $O/prof_default.log 2>&1
python3 $R/tools/rocprof_replay_window.py $DB $O/r4_final_replay_per_step.csv 10 >> $O/prof_default.log 2>&1
python3 $R/tools/step_timeline.py $DB 2 > $O/r4_final_step_timeline.txt 2>&1
tail -2 $O/prof_default.log
cd $R
python tools/bench_attention.py > $O/attention_ubench.txt 2>/dev/null
python tools/graph_two_encoders.py 2>/dev/null | grep parallel > $O/two_encoders.txt
[ -x tools/ubench/bin/small_kernels ] && { for c in "" 2,1,2,2 2,1,1,4 1,1,1,4; do echo "== MRFA_CONV_LDS=${c:+1} MRFA_LDS_CFG=$c"; MRFA_CONV_LDS=${c:+1} MRFA_LDS_CFG=$c ./tools/ubench/bin/small_kernels | head -4; done > $O/small_kernels_conv_lds.txt 2>&1; }
./tools/ubench/bin/small_kernels > $O/small_kernels.txt 2>&1
MRFA_WGRAD_SMALL_ROWS=0 ./tools/ubench/bin/small_kernels > $O/small_kernels_wgrad_general_form.txt 2>&1
MRFA_WGRAD_SMALL_B64=0 ./tools/ubench/bin/small_kernels > $O/small_kernels_wgrad_32x32_blocks.txt 2>&1
python tools/step_phases.py 8 mtia 20 2>/dev/null | grep -v amdgpu > $O/step_phases.txt
python tools/step_phases.py 8 fomm 20 2>/dev/null | grep -v amdgpu > $O/step_phases_fomm.txt
python tools/enc_chain_probe.py mrfa_bn_act_fwd mrfa_conv2d_nhwc 2>/dev/null | grep -v amdgpu > $O/enc_chain_probe.txt
python tools/sweep_splitk.py 2>/dev/null | grep -v amdgpu > $O/sweep_splitk.txt
for k in "MRFA_BN_FIN_FUSED=0" "MRFA_BN_BWD_IN_DGRAD=0" "MRFA_PACK_STREAM=0" "MRFA_ENC_WGRAD_MULTI=0" "MRFA_WGRAD_SMALL_B64=0" "MRFA_HG_LANES=1" "X=0"; do echo "$k $(env $k python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-forward --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")"; done > $O/switch_ablation.txt 2>&1
MRFA_WINO=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-forward > $O/bench_wino.log 2>/dev/null
for v in 0 1; do python tools/bench_kernels.py --mfma 1 --iters 20 --only "3x3 @" --tune conv_halo_wino=$v 2>/dev/null | grep -v "hr \|ups\|@8\|@4\|->2 \|->1 "; done > $O/wino_ubench.txt
ls $O
