#!/bin/bash
# GPU-box side of an A/B over several scenes: tools/ab_scenes.sh name1 name2 ...  (variants built by tools/ab_build.sh)
cd "$(dirname "$0")/.."
cp clraytracer_amd/csrc/libcrt_hip.so /tmp/libcrt_hip.default.so
for name in "$@"; do
  cp build/ab/$name/libcrt_hip.so clraytracer_amd/csrc/libcrt_hip.so
  line="$name"
  for sc in multi-1M sponza-sibenik multi-1M-dense nanosuit-demo; do
    line="$line | $sc $(python bench.py --scene $sc --no-cpu-baseline --no-config5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['synchronous_frames']['value'])")"
  done
  echo "$line"
done
cp /tmp/libcrt_hip.default.so clraytracer_amd/csrc/libcrt_hip.so
