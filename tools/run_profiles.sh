#!/bin/bash
# Runs on the GPU box (through gpurun, from the repo root): kernel-trace stats of a bench command plus separate PMC passes
# (one counter group per run; never combined with other tracing).
# Usage: [PROF_ARGS="--scene multi-1M-dense"] tools/run_profiles.sh <tag> [full|lite]   -> gpurun_out/prof_<tag>_*/
#        summarise with tools/profile_summary.py <tag>. `lite`: HBM traffic, L1/L2 hit rates and wave-wait counters only.
set -o pipefail
tag=${1:-r03}
mode=${2:-full}
export TMPDIR=/tmp
out=gpurun_out
# --no-extras: the extra views / sizes would launch the timed region's kernel on other workloads and blur its per-launch means
# --timed-region-only: the trace then holds launches of the timed region's mode only (3 warm-up + 30 timed), so its average
# duration can be read against the bench line's own HIP-event figure (kernel_ms.crt_trace_kernel_launch_mean) of the same run
B="python3 bench.py --steps 30 --warmup 3 --timed-region-only $PROF_ARGS"
S="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras $PROF_ARGS"
pass() { # name, counters...
  local name=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/prof_${tag}_$name -- $S > $out/prof_${tag}_$name.log 2>&1 || { echo "pass $name failed (see $out/prof_${tag}_$name.log); stopping"; exit 1; }
}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_stats -- $B > $out/prof_${tag}_stats.log 2>&1 || { echo "stats pass failed"; exit 1; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass misc GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY
pass sq2 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR
( pass sq3 SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM_RD ) || echo "(optional pass sq3 not collected)"
if [ "$mode" = full ]; then PROF_ARGS="$PROF_ARGS" tools/ta_only.sh $tag || exit 1; fi
echo done
