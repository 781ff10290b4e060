#!/bin/bash
# A/B on the GPU box: rebuild libcrt_hip.so with different occupancy targets / LDS stack sizes and bench.
# Usage: tools/ab_occupancy.sh "5:16 6:16 8:16 5:32"   (waves_per_simd:lds_slots)
set -e
cd "$(dirname "$0")/.."
for cfg in $1; do
  w=${cfg%%:*}; l=${cfg##*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -DCRT_WAVES_PER_SIMD=$w -DCRT_LDS_SLOTS=$l \
      -shared -o clraytracer_amd/csrc/libcrt_hip.so clraytracer_amd/csrc/crt_shim.hip 2>&1 | grep -E "error" || true
  for k in tile persistent; do
    r=$(CRT_KERNEL=$k CRT_WAVES_PER_CU=$((w*4)) python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms']['crt_trace_kernel_mean'])")
    echo "waves/SIMD=$w lds_slots=$l kernel=$k : Mrays/s, kernel ms = $r"
  done
done
