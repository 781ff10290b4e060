#!/bin/bash
# GPU-box side: bench.py under different environment settings (runtime knobs read by crt_init).
# Usage: tools/env_sweep.sh "" "CRT_SPLIT_BETA=1.0 CRT_SPLIT=64" ...   (BENCH_ARGS adds bench.py arguments)
cd "$(dirname "$0")/.."
for setting in "$@"; do
  r=$(env $setting timeout -k 10 240 python bench.py --steps ${AB_STEPS:-100} --warmup 10 --no-cpu-baseline --no-extras $BENCH_ARGS 2>gpurun_out/sweep.err | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d.get('synchronous_frames',{})
print(d['value'], d['ms_per_step'], 'sync', s.get('value'), s.get('ms_per_step'))")
  echo "[$setting] Mrays/s, ms/frame: $r" | tee -a gpurun_out/sweep_results.txt
done
