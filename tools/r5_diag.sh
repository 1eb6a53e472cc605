#!/bin/bash
# round-5 diagnostics on one box: where the ATen fill / copy launches of a step come from, the phase times of the default schedule
O=gpurun_out/r5f; mkdir -p $O
python tools/trace_fills.py mtia > $O/trace_fills.txt 2>&1; tail -40 $O/trace_fills.txt
python tools/step_phases.py 8 mtia 20 2>/dev/null | grep -v amdgpu > $O/step_phases.txt; cat $O/step_phases.txt
