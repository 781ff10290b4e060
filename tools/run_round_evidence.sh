#!/bin/bash
# GPU-box side: everything profiles/<tag>_* is made from that is not a PMC pass (those: tools/run_profiles.sh), in one call.
#   tools/run_round_evidence.sh r03   -> gpurun_out/{<tag>_bench_lines.jsonl, <tag>_wave_timeline.txt, <tag>_cpu_baseline.md, <tag>_ubench_gather.txt, <tag>_ubench_chain.*}
tag=${1:-r03}
out=gpurun_out
tools/run_bench_lines.sh $tag
python tools/wave_timeline.py multi-1M 1920 1080 > $out/${tag}_wave_timeline.txt 2>&1
CRT_TL_RANKS=8 python tools/wave_timeline.py multi-1M 3840 2160 >> $out/${tag}_wave_timeline.txt 2>&1
python tools/cpu_baseline.py > $out/${tag}_cpu_baseline.md 2>&1
timeout -k 10 120 tools/ubench/gather > $out/${tag}_ubench_gather.txt 2>&1
timeout -k 10 300 tools/ubench/chain $out/${tag}_ubench_chain.json > $out/${tag}_ubench_chain.txt 2>&1
echo evidence done
