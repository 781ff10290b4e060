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
threads = min(cores, 64)
print(f"host: {cores} logical CPUs ({open('/proc/cpuinfo').read().split('model name')[1].split(':')[1].splitlines()[0].strip()}); using 1 and {threads} threads\n")
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
        t0 = time.perf_counter(); s.cpu_raycast(origins, rays, nthreads=nt); dt = time.perf_counter() - t0
        print(f"| C1 CPU_RayCast | cornell-1k | 640x480 | {nt} | {dt * 1e3:.1f} ms | {len(rays) / dt / 1e6:.2f} |")

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
