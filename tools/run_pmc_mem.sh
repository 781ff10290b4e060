#!/bin/bash
# Vector-memory-path PMC passes (TCP / TD / SQ), one small group per run, each under its own timeout
# (a TA_* group hung rocprofv3 on this pool once: TA counters are deliberately not collected).
# Usage: tools/run_pmc_mem.sh <tag>
tag=${1:-r02}
export TMPDIR=/tmp
S="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline"
i=0
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"; do
  i=$((i+1))
  echo "group $i: $grp"
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/prof_${tag}_mem$i -- $S > gpurun_out/prof_${tag}_mem$i.log 2>&1 || echo "group $i failed/timeout"
done
echo done
