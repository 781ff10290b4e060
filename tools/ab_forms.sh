#!/bin/bash
# GPU-box side: the default kernel against opt-in kernel forms on several scenes in ONE call, alternating (no rebuild: CRT_KERNEL is read by crt_init).
#   tools/ab_forms.sh default ldstop default ldstop      AB_SCENES / BENCH_ARGS as in tools/ab_scenes.sh
cd "$(dirname "$0")/.."
for form in "$@"; do
  line="$form"
  for sc in ${AB_SCENES:-multi-1M multi-1M-dense sponza-sibenik}; do
    line="$line | $sc $(CRT_KERNEL=$form timeout -k 10 240 python bench.py --scene $sc --no-cpu-baseline --no-extras $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['synchronous_frames']['value'])")"
  done
  echo "$line" | tee -a gpurun_out/ab_forms.txt
done
