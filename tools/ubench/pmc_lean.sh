#!/bin/bash
# PMC passes (one counter per pass, --kernel-trace only) over tools/ubench/lean_bench at N = 16, conv_lean.hip only:  bash tools/ubench/pmc_lean.sh [GEO]
set -u; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; O=$R/gpurun_out/pmc_lean${1:+_geo$1}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export NONLY=16 ONLY_LEAN=1 ITERS=20
if [ -n "${1:-}" ]; then export GEO=$1; fi
for c in SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE; do
  rm -rf /tmp/pl_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pl_$c -o p -- $R/tools/ubench/bin/lean_bench > /tmp/pl_$c.log 2>&1
  f=$(find /tmp/pl_$c -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $R/tools/pmc_per_kernel.py $f $O/$c.csv > /dev/null 2>&1; echo "$c $(grep -v '^kernel' $O/$c.csv | grep lean | awk -F, '{print $(NF-1), $NF}' | tr '\n' ' ')"; else echo "$c: no output"; tail -2 /tmp/pl_$c.log; fi
done
