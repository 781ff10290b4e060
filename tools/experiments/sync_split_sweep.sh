#!/bin/bash
# Sweep of CRT_SYNC_SPLIT, a knob that exists ONLY with tools/experiments/r04_sync_split_cu_mask.patch applied (round 4's reserved-CU
# experiment, DESIGN_HISTORY 7); against the shipped library every row measures the same thing, so this refuses to run without the patch.
cd "$(dirname "$0")/../.."
grep -q CRT_SYNC_SPLIT clraytracer_amd/csrc/*.h clraytracer_amd/csrc/*.hip || { echo "CRT_SYNC_SPLIT is not in the tree: git apply tools/experiments/r04_sync_split_cu_mask.patch (written against round 4's crt_shim.hip) and rebuild first"; exit 1; }
run() { CRT_SYNC_SPLIT="$1" timeout -k 10 120 python bench.py --frames-in-flight 1 --no-cpu-baseline --no-extras $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split', '$1', '$2', d['value'], d['ms_per_step'])"; }
run "0,0"
for r in 1 2 4; do for h in 8 16 32 64 128; do run "$r,$h"; done; done
run "0,0"
