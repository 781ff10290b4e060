#!/bin/bash
# GPU-box side of an A/B: bench every prebuilt variant under build/ab/ (tools/ab_build.sh), restoring the default library
# afterwards (also when interrupted). BENCH_ARGS adds bench.py arguments. Usage: tools/ab_run.sh [name ...]   (default: all)
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
names="$@"; [ -z "$names" ] && names=$(ls build/ab)
for name in $names; do
  [ -f build/ab/$name/libcrt_hip.so ] || { echo "[$name] no library (build failed?)" | tee -a gpurun_out/ab_results.txt; continue; }
  cp build/ab/$name/libcrt_hip.so clraytracer_amd/csrc/libcrt_hip.so
  r=$(timeout -k 10 240 python bench.py --steps ${AB_STEPS:-100} --warmup 10 --no-cpu-baseline --no-extras $BENCH_ARGS 2>gpurun_out/ab_$name.err | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d.get('synchronous_frames',{})
print(d['value'], d['ms_per_step'], 'sync', s.get('value'), s.get('ms_per_step'))")
  echo "[$name: $(cat build/ab/$name/flags.txt)] Mrays/s, ms/frame: $r" | tee -a gpurun_out/ab_results.txt
done
