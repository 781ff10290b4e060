#!/usr/bin/env python3
"""CPU rows of BASELINE.md's measurement plan, timed on this box's host cores (run on the GPU box):
  C1  the mirrored CPU_RayCast (CPURayTrace.cpp:186-249), one call per pixel, config 1 (cornell-1k 640x480), 1 and all threads
  C2  the scalar Trace oracle (restatement of kernel_main.cl:164-275), configs 2-4 at 1920x1080, 1 and all threads
  C3  BuildBVH (BVH.cpp:218-255) through the mirrored ResourceManager::PushMeshesToGPU (per-mesh parallel), every scene
plus the PCIe leg of the boundary: crt_read_output of a 1920x1080 float4 frame.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from clraytracer_amd import driver, scenes  # noqa: E402
import oracle_lib  # noqa: E402

cores = os.cpu_count() or 1
affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else cores
try:
    quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
    cgroup = "unlimited" if quota == "max" else f"{float(quota) / float(period):.1f} CPUs ({quota}/{period} us)"
    usable = affinity if quota == "max" else max(1, min(affinity, int(float(quota) / float(period))))
except Exception:
    cgroup, usable = "unknown", affinity
threads = int(os.environ.get("CRT_CPU_THREADS", usable))       # the rule bench.py's cpu_baseline leg uses: the CPUs this process may actually use
model = open('/proc/cpuinfo').read().split('model name')[1].split(':')[1].splitlines()[0].strip()
print(f"host: {model}; {cores} logical CPUs, {affinity} in this process's affinity mask, cgroup CPU quota {cgroup} -> {usable} usable CPUs.")
print(f"Rows are timed on 1 thread and on {threads} threads (= the usable CPUs; more threads than that are time-sliced by the quota and measure nothing new). "
      f"bench.py's `cpu_baseline` object is the same experiment with the same thread rule: `cores` there = {threads} on this box.\n")
print("| row | scene | frame | threads | time | Mrays/s |\n|---|---|---|---|---|---|")

# C3 + C1 on cornell
for name in ("cornell-1k", "sponza-class-250k", "multi-1M"):
    sc = scenes.get(name)
    t0 = time.perf_counter()
    s = driver.Session(64, 48, host_only=True)
    s.h.crth_prepare_meshes(); s.h.crth_import_texture(sc.skybox.encode())
    for p in sc.meshes:
        s.h.crth_import_mesh(p.encode())
    t1 = time.perf_counter()
    s.h.crth_push_meshes()
    t2 = time.perf_counter()
    print(f"| C3 OBJ import | {name} | - | 1 | {t1 - t0:.3f} s | - |")
    print(f"| C3 BuildBVH | {name} | {sc.num_tris} tris | up to {len(sc.meshes)} (one per mesh) | {t2 - t1:.3f} s | - |")
    s.close()

sc = scenes.get("cornell-1k")
w, h = 640, 480
with driver.Session(w, h, host_only=True) as s:
    s.load_scene(sc)
    iv, ip, pos = s.camera()
    orc = oracle_lib.Oracle(s.arenas(), nthreads=threads)
    rays = orc.raygen(w, h, iv, ip).reshape(-1, 3)
    origins = np.tile(pos, (len(rays), 1)).astype(np.float32)
    for nt in (1, threads):
        for sse in (False, True):
            dt = None
            for _ in range(5 if nt > 1 else 2):
                t0 = time.perf_counter(); s.cpu_raycast(origins, rays, nthreads=nt, sse=sse); d = time.perf_counter() - t0
                dt = d if dt is None else min(dt, d)
            print(f"| C1 CPU_RayCast ({'SSE flavour: rcpps/dpps as upstream' if sse else 'scalar, IEEE divide'}) | cornell-1k | 640x480 | {nt} | {dt * 1e3:.1f} ms | {len(rays) / dt / 1e6:.2f} |")

for name in ("cornell-1k", "sponza-class-250k", "multi-1M"):
    sc = scenes.get(name)
    w, h = 1920, 1080
    with driver.Session(w, h, host_only=True) as s:
        s.load_scene(sc)
        iv, ip, pos = s.camera()
        arenas = s.arenas()
    for nt in (1, threads):
        orc = oracle_lib.Oracle(arenas, nthreads=nt)
        rays = orc.raygen(w, h, iv, ip)
        t0 = time.perf_counter(); _, st = orc.trace(rays, pos, sc.sun_angle); dt = time.perf_counter() - t0
        print(f"| C2 Trace oracle | {name} | 1920x1080 | {nt} | {dt * 1e3:.1f} ms | {st['rays'] / dt / 1e6:.2f} |")

try:
    with driver.Session(1920, 1080, device=0) as s:
        s.load_scene(scenes.get("multi-1M"))
        for _ in range(3):
            s.render_raw(0)
        ts = []
        for _ in range(10):
            s.render_raw(0)
            t0 = time.perf_counter(); s.read_output(); ts.append(time.perf_counter() - t0)
        print(f"\nPCIe leg: crt_read_output of a 1920x1080 float4 frame (33.2 MB, pageable host memory): median {np.median(ts) * 1e3:.2f} ms "
              f"= {33.1776 / np.median(ts) / 1e3:.1f} GB/s; frame kernel time {s.kernel_ms(2):.3f} ms")
except Exception as e:  # no GPU here
    print("\n(no GPU: PCIe leg skipped)", e)
