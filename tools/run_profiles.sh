#!/bin/bash
# Runs on the GPU box (through gpurun, from the repo root): kernel-trace stats of the default bench
# command plus separate PMC passes (one counter group per run; never combined with other tracing).
# Usage: tools/run_profiles.sh <tag>      -> gpurun_out/prof_<tag>_*/ ; summarise with tools/profile_summary.py
set -o pipefail
tag=${1:-r02}
export TMPDIR=/tmp
out=gpurun_out
# --no-config5: the extra 3840x2160 point would launch the timed region's kernel at another size and blur its per-launch means
B="python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-config5"
S="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-config5"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_stats -- $B > $out/prof_${tag}_stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/prof_${tag}_fetch -- $S > $out/prof_${tag}_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/prof_${tag}_write -- $S > $out/prof_${tag}_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/prof_${tag}_tcc -- $S > $out/prof_${tag}_tcc.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $out/prof_${tag}_sq1 -- $S > $out/prof_${tag}_sq1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR --output-format csv -d $out/prof_${tag}_sq2 -- $S > $out/prof_${tag}_sq2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $out/prof_${tag}_misc -- $S > $out/prof_${tag}_misc.log 2>&1 || true
tools/ta_only.sh $tag
echo done
