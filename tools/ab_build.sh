#!/bin/bash
# Build-container side of an A/B: compile libcrt_hip.so variants (extra compiler flags each) into build/ab/<name>/.
# The built files travel to the GPU box with the snapshot; tools/ab_run.sh benches them there.
# Usage: tools/ab_build.sh name1 "flags1" name2 "flags2" ...
# A variant whose build fails leaves NO library behind (the target is deleted first), so a stale one is never benchmarked.
cd "$(dirname "$0")/.."
fail=0
pids=()
names=()
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  mkdir -p build/ab/$name
  rm -f build/ab/$name/libcrt_hip.so
  echo "$flags" > build/ab/$name/flags.txt
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function $flags \
      -shared -o build/ab/$name/libcrt_hip.so.tmp clraytracer_amd/csrc/crt_shim.hip > build/ab/$name/build.log 2>&1 \
      && mv build/ab/$name/libcrt_hip.so.tmp build/ab/$name/libcrt_hip.so ) &
  pids+=($!); names+=($name)
done
for i in "${!pids[@]}"; do
  if ! wait ${pids[$i]}; then echo "BUILD FAILED: ${names[$i]} (build/ab/${names[$i]}/build.log)"; grep -E "error" build/ab/${names[$i]}/build.log | head -5; rm -f build/ab/${names[$i]}/libcrt_hip.so.tmp; fail=1; fi
done
ls -la build/ab/*/libcrt_hip.so
exit $fail
