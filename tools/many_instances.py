#!/usr/bin/env python3
"""Frame time against the number of instances (the reference loops over every instance for every ray, Renderer.hpp:16
allows 401): `tiny`'s two meshes instanced n times on a grid. Run on the GPU box."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clraytracer_amd import _lib, driver, scenes

sc = scenes.get("tiny")
W, H = 1920, 1080
SPACING = float(os.environ.get("CRT_MI_SPACING", "6.0"))
for n, tlas in ((16, "0"), (64, "0"), (128, "0"), (128, "1"), (256, "0"), (256, "1"), (401, "0"), (401, "1")):
    os.environ["CRT_TLAS"] = tlas                      # read by crt_init: 0 = linear sphere loop, 1 = instance tree
    with driver.Session(W, H, device=0) as s:
        s.load_scene(sc)
        s.h.crth_begin_instances()
        for k in range(len(sc.instances), n):
            m = scenes._trs(0.6 + 0.1 * (k % 5), (0.3, 1.0, 0.2), 0.37 * k, (float((k % 21) - 10) * SPACING, float((k // 21) - 9) * SPACING, -float(k % 7) * 2.0))
            p, keep = _lib.fptr(m)
            s.h.crth_register_instance(k % 2, 0xFFFF, p)
        s.h.crth_end_instances()
        s.set_camera((0.0, 0.0, 23.0 * SPACING), scenes._normalize((0.0, 0.0, -1.0)))
        s.render(postprocess=False)
        s.render_raw(8); c = s.counters()
        hip = _lib.hip()
        a, iv, ip = s.trace_args()
        fp = C.POINTER(C.c_float)
        args = (C.byref(a), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp))
        for _ in range(5): hip.crt_render(*args, 4)
        hip.crt_sync(); t0 = time.perf_counter()
        for _ in range(30): hip.crt_render(*args, 4)
        hip.crt_sync(); dt = (time.perf_counter() - t0) / 30
        print(f"{n:4d} instances, {'tree  ' if tlas == '1' else 'linear'}: {dt * 1e3:7.3f} ms/frame, {c['rays'] / dt / 1e9:6.2f} Gray/s, hits {c['hits']}, algorithmic inner visits/ray {c['innerVisits'] / c['rays']:.1f}")
