#!/bin/bash
# Build-container side of an A/B: compile libcrt_hip.so variants (extra compiler flags each) into build/ab/<name>/.
# The built files travel to the GPU box with the snapshot; tools/ab_run.sh benches them there.
# Usage: tools/ab_build.sh name1 "flags1" name2 "flags2" ...
cd "$(dirname "$0")/.."
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  mkdir -p build/ab/$name
  echo "$flags" > build/ab/$name/flags.txt
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function $flags \
      -shared -o build/ab/$name/libcrt_hip.so clraytracer_amd/csrc/crt_shim.hip 2>&1 | grep -E "error" &
done
wait
ls -la build/ab/*/libcrt_hip.so
