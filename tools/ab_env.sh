#!/bin/bash
# GPU-box side: bench.py under several environment settings in ONE call (knobs read by crt_init), alternating with the baseline.
#   tools/ab_env.sh "" "CRT_DEEP=8" "" "CRT_DEEP=16 CRT_DEEP_BETA=1.8"       AB_SCENES / BENCH_ARGS as in tools/ab_scenes.sh
cd "$(dirname "$0")/.."
for setting in "$@"; do
  line="[${setting:-baseline}]"
  for sc in ${AB_SCENES:-multi-1M multi-1M-dense}; do
    line="$line | $sc $(env $setting timeout -k 10 240 python bench.py --scene $sc --no-cpu-baseline --no-extras $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['synchronous_frames']['value'], d['synchronous_frames']['ms_per_step'])")"
  done
  echo "$line" | tee -a gpurun_out/ab_env.txt
done
