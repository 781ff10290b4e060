#!/bin/bash
# GPU-box side: busy / stall counters of the CU's vector-memory pipeline (TA address unit, TD data return, TCP = L1) for the
# trace kernel, two counters per pass (more per pass "exceeds the capabilities of the hardware" for these blocks).
# Usage: [PROF_ARGS="--diag-mix3"] tools/ta_only.sh <tag>
export TMPDIR=/tmp
out=gpurun_out; tag=${1:-r03}
S="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras $PROF_ARGS"
i=0
for grp in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_BUSY_avr" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TD_TD_BUSY_sum TD_TCP_STALL_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  i=$((i+1))
  echo "pass $i: $grp"
  # a pass that fails or hangs (these blocks have hung rocprofv3 before when asked for too many counters at once) ends the
  # script: no further GPU step after a timeout
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/prof_${tag}_vm$i -- $S > $out/prof_${tag}_vm$i.log 2>&1 || { echo "pass $i failed (see $out/prof_${tag}_vm$i.log); stopping"; exit 1; }
done
echo done
