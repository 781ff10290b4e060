#!/bin/bash
# A/B on the GPU box: rebuild libcrt_hip.so with extra -D flags and bench. Usage: tools/ab_define.sh "-DX=1" "-DX=2" ...
# (BENCH_ARGS in the environment adds bench.py arguments, e.g. BENCH_ARGS="--frames-in-flight 1")
cd "$(dirname "$0")/.."
for def in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function $def \
      -shared -o clraytracer_amd/csrc/libcrt_hip.so clraytracer_amd/csrc/crt_shim.hip 2>&1 | grep -E "error" || true
  r=$(python bench.py --steps 50 --warmup 5 --no-cpu-baseline $BENCH_ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms'])")
  echo "[$def] Mrays/s, ms/frame, kernel: $r"
done
