#!/bin/bash
# PMC passes (one counter per pass, --kernel-trace only) over one shape of tools/ubench/small_kernels: SHAPE=<row> bash tools/ubench/pmc_small.sh
set -u; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/pmc_small_$SHAPE; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_TA_BUSY_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE; do
  rm -rf /tmp/pm_$c
  ITERS=20 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pm_$c -o p -- $R/tools/ubench/bin/small_kernels > /tmp/pm_$c.log 2>&1
  f=$(find /tmp/pm_$c -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $R/tools/pmc_per_kernel.py $f $O/$c.csv > /dev/null 2>&1; echo "$c $(grep -v '^kernel' $O/$c.csv | head -3 | awk -F, '{print $(NF-1), $NF}' | tr '\n' ' ')"; else echo "$c: no output"; tail -2 /tmp/pm_$c.log; fi
done
