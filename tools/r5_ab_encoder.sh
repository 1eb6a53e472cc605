#!/bin/bash
# round 5, first GPU pass: kernel / graph / headline tests of the statistic groups, then the batched encoder pass against the two concurrent passes
# (alternating on one box), and the phase times of both schedules.  Output under gpurun_out/r5a/.
O=gpurun_out/r5a; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "groups or bn_ or test_conv2d or finalize or attention" > $O/kernels.log 2>&1; tail -3 $O/kernels.log
python -m pytest tests/test_graph_gpu.py -x -q -m gpu -k "batched or concurrent or three_passes" > $O/graph.log 2>&1; tail -3 $O/graph.log
python -m pytest tests/test_headline.py tests/test_tokenpose.py -x -q -m gpu > $O/headline.log 2>&1; tail -3 $O/headline.log
for r in 1 2 3; do for v in 1 0; do
  MRFA_BATCHED_ENCODER=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-forward --no-roofline 2>$O/bench_err_$v.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batched=$v', d['ms_per_step'], d['value'])"
done; done | tee $O/ab.txt
MRFA_BATCHED_ENCODER=1 python tools/step_phases.py > $O/phases_batched.txt 2>&1; cat $O/phases_batched.txt
MRFA_BATCHED_ENCODER=0 python tools/step_phases.py > $O/phases_two_passes.txt 2>&1; head -3 $O/phases_two_passes.txt
