#!/bin/bash
# GPU-box side: per-launch durations of the stand-alone per-pixel kernels (crt_postprocess_kernel, crt_quantize_kernel,
# crt_fxaa_kernel, crt_pack_unorm8_kernel) from rocprofv3 --kernel-trace --stats. The default Trace kernel applies
# PostProcess / RGBA8 in its epilogue, so the stand-alone forms are run through the wavefront variant (CRT_KERNEL=wavefront).
export TMPDIR=/tmp
out=gpurun_out
CRT_KERNEL=wavefront rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_stage_wf -- python3 tools/stage_cost.py > $out/prof_stage_wf.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_stage_def -- python3 tools/stage_cost.py > $out/prof_stage_def.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob
for tag in ("wf", "def"):
    f = glob.glob(f"gpurun_out/prof_stage_{tag}/*/*_kernel_stats.csv")[0]
    print("==", tag)
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("postprocess", "quantize", "fxaa", "pack_unorm8")):
            print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"]) / 1e3:8.1f} us  min {float(r["MinNs"]) / 1e3:8.1f} us')
PY
