#!/bin/bash
# GPU-box side: PMC passes over the chain microbenchmark itself (tools/ubench/chain --brief): is the CU's vector-memory path what saturates
# at the ceiling bench.py's roofline.chain is taken against? Two counters per pass for the TA/TD/TCP blocks (tools/ta_only.sh explains why).
#   tools/ubench_pmc.sh <tag>  -> gpurun_out/prof_<tag>_ub*/ ; summarise with tools/ubench_pmc_summary.py <tag>
export TMPDIR=/tmp
out=gpurun_out; tag=${1:-r03}; mode=${2:---brief}      # --brief-transposed: the round-4 per-lane / quad-transposed pairs
i=0
for grp in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum" "TD_TD_BUSY_sum TD_TCP_STALL_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  echo "pass $i: $grp"
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/prof_${tag}_ub$i -- tools/ubench/chain $mode > $out/prof_${tag}_ub$i.log 2>&1 || { echo "pass $i failed (see $out/prof_${tag}_ub$i.log); stopping"; exit 1; }
done
echo done
