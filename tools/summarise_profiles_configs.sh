#!/bin/bash
# Build container side: digest what tools/run_profiles_configs.sh left under gpurun_out/ into profiles/<tag>_summary.{md,json}.
r=${1:-r05}
P="crt_trace_kernel<false, false, false, false, false>"; S="crt_trace_kernel<false, false, true, false, false>"
for t in cfg2 cfg3 cfg5 dense sponza nano; do [ -d gpurun_out/prof_${r}${t}_stats ] && python tools/profile_summary.py ${r}${t} "$P" > /dev/null; done
for t in cfg3s shadow; do [ -d gpurun_out/prof_${r}${t}_stats ] && python tools/profile_summary.py ${r}${t} "$S" > /dev/null; done
[ -d gpurun_out/prof_${r}refill_stats ] && python tools/profile_summary.py ${r}refill "crt_trace_refill_kernel<false, false>" > /dev/null
[ -d gpurun_out/prof_${r}block_stats ] && python tools/profile_summary.py ${r}block "crt_trace_block_kernel<false, false>" > /dev/null
[ -d gpurun_out/prof_${r}wavefront_stats ] && python tools/profile_summary.py ${r}wavefront "crt_bounce_kernel<false>" > /dev/null
[ -d gpurun_out/prof_${r}_stats ] && python tools/profile_summary.py ${r} "$P" > /dev/null
ls profiles/${r}*_summary.md
