#!/usr/bin/env python3
"""Compile crt_shim.hip for gfx950 with -Rpass-analysis=kernel-resource-usage and print one line per kernel:
VGPRs, AGPRs, scratch bytes per lane, occupancy, LDS bytes. Runs without a GPU (hipcc cross-compiles).

    python tools/kernel_resources.py [filter-substring] [-D...]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    flt = [a for a in sys.argv[1:] if not a.startswith("-")]
    defs = [a for a in sys.argv[1:] if a.startswith("-")]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
           "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(ROOT, "clraytracer_amd/csrc/crt_shim.hip"), "-o", "/dev/null"] + defs
    p = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    if p.returncode:
        sys.stderr.write(p.stderr)
        raise SystemExit(p.returncode)
    cur = None
    rows = []
    for line in p.stderr.splitlines():
        m = re.search(r"remark: +Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[a-zA-Z/]+\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], stdout=subprocess.PIPE, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        if flt and not any(f in name for f in flt):
            continue
        print(f"{name:70s} VGPR {r.get('VGPRs', -1):3d} AGPR {r.get('AGPRs', -1):3d} scratch {r.get('ScratchSize', -1):4d} occ {r.get('Occupancy', -1):2d} LDS {r.get('LDS Size', -1):6d} SGPR {r.get('TotalSGPRs', -1):3d} spillV {r.get('VGPRs Spill', -1):3d}")


if __name__ == "__main__":
    main()
