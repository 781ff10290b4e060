#!/bin/bash
# A/B on the GPU box: rebuild libcrt_hip.so with extra -D flags and bench; the default library is restored afterwards.
# Usage: tools/ab_define.sh "-DX=1" "-DX=2" ...   (BENCH_ARGS in the environment adds bench.py arguments, e.g. "--frames-in-flight 1")
cd "$(dirname "$0")/.."
cp clraytracer_amd/csrc/libcrt_hip.so /tmp/libcrt_hip.default.so
trap 'cp /tmp/libcrt_hip.default.so clraytracer_amd/csrc/libcrt_hip.so' EXIT
for def in "$@"; do
  rm -f /tmp/libcrt_hip.variant.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function $def \
      -shared -o /tmp/libcrt_hip.variant.so clraytracer_amd/csrc/crt_shim.hip > /tmp/ab_define.log 2>&1 || { echo "[$def] BUILD FAILED"; grep error /tmp/ab_define.log | head -3; continue; }
  cp /tmp/libcrt_hip.variant.so clraytracer_amd/csrc/libcrt_hip.so
  r=$(python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras $BENCH_ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms'])")
  echo "[$def] Mrays/s, ms/frame, kernel: $r"
done
