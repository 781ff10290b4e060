#!/bin/bash
# A/B on the GPU box: rebuild libcrt_hip.so with extra -D flags and bench; the default library is restored afterwards.
# Usage: tools/ab_define.sh "-DX=1" "-DX=2" ...   (BENCH_ARGS in the environment adds bench.py arguments, e.g. "--frames-in-flight 1")
cd "$(dirname "$0")/.."
# the default library waits in a file of this run's own (two A/B runs on one box, or a killed run's leftovers, cannot restore the
# wrong one) and its hash is checked after the restore
saved=$(mktemp /tmp/libcrt_hip.default.XXXXXX.so) || exit 1
cp clraytracer_amd/csrc/libcrt_hip.so "$saved"
want=$(sha256sum < "$saved")
restore() {
  cp "$saved" clraytracer_amd/csrc/libcrt_hip.so
  [ "$(sha256sum < clraytracer_amd/csrc/libcrt_hip.so)" = "$want" ] || echo "WARNING: the restored libcrt_hip.so is not the library this run started with -- rebuild with make" >&2
  rm -f "$saved" "$variant"
}
trap restore EXIT
variant=$(mktemp /tmp/libcrt_hip.variant.XXXXXX.so) || exit 1
log=$(mktemp /tmp/ab_define.XXXXXX.log)
for def in "$@"; do
  rm -f "$variant"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function $def \
      -shared -o "$variant" clraytracer_amd/csrc/crt_shim.hip > "$log" 2>&1 || { echo "[$def] BUILD FAILED"; grep error "$log" | head -3; continue; }
  cp "$variant" clraytracer_amd/csrc/libcrt_hip.so
  r=$(python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras $BENCH_ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms'])")
  echo "[$def] Mrays/s, ms/frame, kernel: $r"
done
