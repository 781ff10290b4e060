#!/usr/bin/env python3
"""Digest gpurun_out/prof_bvh/bvh_results.db (rocprofv3 --kernel-trace of tools/bvh_build_time.py, rocpd format) into profiles/<tag>_bvh_build.md:
the last of the run's three device builds, kernel by kernel.
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bvh -o bvh -- python3 tools/bvh_build_time.py multi-1M > gpurun_out/bvh_prof.log
    python tools/bvh_build_profile.py r03"""
import collections
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
db = sqlite3.connect(os.path.join(ROOT, "gpurun_out/prof_bvh/bvh_results.db"))
rows = db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "crt_bvh_centroids" in r[0]]
s = starts[-1]
e = next(i for i in range(s, len(rows)) if "crt_make_root_refs" in rows[i][0])
seg = rows[s:e + 1]
acc = collections.defaultdict(lambda: [0, 0, 0])
for n, a, b, g, w in seg:
    k = n.split("(")[0]
    acc[k][0] += 1; acc[k][1] += b - a; acc[k][2] = max(acc[k][2], b - a)
line = [l for l in open(os.path.join(ROOT, "gpurun_out/bvh_prof.log")) if l.startswith("multi-1M") or ": " in l and "triangles in" in l]
with open(os.path.join(ROOT, "profiles", f"{tag}_bvh_build.md"), "w") as f:
    f.write(f"# `crt_build_bvh` on MI355X — rocprofv3 `--kernel-trace` of `python3 tools/bvh_build_time.py multi-1M` (last of 3 builds, round {tag[1:]})\n\n")
    if line:
        f.write(line[-1].strip() + " (timings under the profiler)\n\n")
    f.write(f"{len(seg)} launches, {sum(v[1] for v in acc.values()) / 1e6:.2f} ms of kernel time inside a {(seg[-1][2] - seg[0][1]) / 1e6:.2f} ms span (one control record published to pinned host memory per level; r3: a 16-byte copy + stream synchronisation per level).\n\n")
    f.write("| kernel | launches | ms | longest launch us |\n|---|---|---|---|\n")
    for k, v in sorted(acc.items(), key=lambda x: -x[1][1]):
        f.write(f"| `{k}` | {v[0]} | {v[1] / 1e6:.3f} | {v[2] / 1e3:.0f} |\n")
print(open(os.path.join(ROOT, "profiles", f"{tag}_bvh_build.md")).read())
