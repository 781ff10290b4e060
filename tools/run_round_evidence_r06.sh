#!/bin/bash
# GPU-box side: round 6's evidence on the final library -- the headline kernel's rocprofv3 passes (full), the N = 1 bench line at 200 steps and at the
# driver's step counts, one line per workload, the wave timeline and the CPU baseline. Summaries: tools/profile_summary.py r06 afterwards.
out=gpurun_out
PROF_ARGS="" tools/run_profiles.sh r06 full > $out/r06_profiles.log 2>&1 || { tail -5 $out/r06_profiles.log; exit 1; }
python bench.py > $out/r06_bench_line.json 2> $out/r06_bench_line.err || { tail -5 $out/r06_bench_line.err; exit 1; }
python bench.py --steps 20 --warmup 5 > $out/r06_bench_line_driver_steps.json 2>> $out/r06_bench_line.err || exit 1
tools/run_bench_lines.sh r06 || exit 1
python tools/wave_timeline.py multi-1M 1920 1080 > $out/r06_wave_timeline.txt 2>&1
python tools/cpu_baseline.py > $out/r06_cpu_baseline.md 2>&1
echo evidence done
