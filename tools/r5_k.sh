mkdir -p gpurun_out/r5final
for i in 1 2 3 4; do
for k in "MRFA_BN_FIN_FUSED=0" "MRFA_FUSED_SPLITK=0" "X=0" "MRFA_FUSED_SPLITK=0 MRFA_BN_FIN_FUSED=0"; do echo "$k $(env $k python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-forward --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")"; done
done | tee gpurun_out/r5final/ab_fusions.txt
python tools/soak_train.py 300 > gpurun_out/r5final/soak_300_steps.log 2>&1; tail -2 gpurun_out/r5final/soak_300_steps.log
