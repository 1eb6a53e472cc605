#!/bin/bash
# full GPU suite + smoke + the default bench line on one MI355X box -> gpurun_out/r5suite/
set -u; O=gpurun_out/r5suite; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1; grep -E "passed|failed" $O/gputest.log | tail -1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>$O/bench_default.err; tail -1 $O/bench_default.log | cut -c1-400
