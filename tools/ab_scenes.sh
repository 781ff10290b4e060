#!/bin/bash
# GPU-box side of an A/B over several scenes: tools/ab_scenes.sh name1 name2 ...  (variants built by tools/ab_build.sh);
# the default library is restored afterwards (also when interrupted). AB_SCENES overrides the scene list.
cd "$(dirname "$0")/.."
# the default library waits in a file of this run's own (two A/B runs on one box, or a killed run's leftovers, cannot restore the
# wrong one) and its hash is checked after the restore
saved=$(mktemp /tmp/libcrt_hip.default.XXXXXX.so) || exit 1
cp clraytracer_amd/csrc/libcrt_hip.so "$saved"
want=$(sha256sum < "$saved")
restore() {
  cp "$saved" clraytracer_amd/csrc/libcrt_hip.so
  [ "$(sha256sum < clraytracer_amd/csrc/libcrt_hip.so)" = "$want" ] || echo "WARNING: the restored libcrt_hip.so is not the library this run started with -- rebuild with make" >&2
  rm -f "$saved"
}
trap restore EXIT
for name in "$@"; do
  [ -f build/ab/$name/libcrt_hip.so ] || { echo "$name | no library (build failed?)" | tee -a gpurun_out/ab_scenes.txt; continue; }
  cp build/ab/$name/libcrt_hip.so clraytracer_amd/csrc/libcrt_hip.so
  line="$name"
  for sc in ${AB_SCENES:-multi-1M sponza-sibenik multi-1M-dense nanosuit-demo}; do
    line="$line | $sc $(timeout -k 10 240 python bench.py --scene $sc --no-cpu-baseline --no-extras $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['synchronous_frames']['value'])")"
  done
  echo "$line" | tee -a gpurun_out/ab_scenes.txt
done
