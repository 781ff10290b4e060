#!/usr/bin/env python3
"""Linear sphere loop vs instance tree on a scene of LARGE, overlapping instances: multi-1M's eight 125 k-triangle meshes instanced 16 ... 64 times
on a grid (the generator's own placement rule), from the bench camera raised to see the grid. Companion of instances_threshold.py (small sparse
instances). Run on the GPU box."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clraytracer_amd import _lib, driver, scenes
sc = scenes.get("multi-1M")
W, H = 1920, 1080
for n in (16, 32, 48, 64):
    per_row = int(np.ceil(np.sqrt(n)))
    for tlas in ("0", "1"):
        os.environ["CRT_TLAS"] = tlas
        with driver.Session(W, H, device=0) as s:
            s.load_scene(sc)
            s.h.crth_clear_instances() if hasattr(s.h, "crth_clear_instances") else None
            s.h.crth_begin_instances()
            for k in range(len(sc.instances), n):
                gx, gz = k % per_row, k // per_row
                r = lambda j: float(scenes._rand01(555, k, j))
                m = scenes._trs(0.5 + 1.5 * r(0), (r(1) - 0.5, r(2) + 0.2, r(3) - 0.5), 2 * np.pi * r(4),
                                ((gx - (per_row - 1) / 2.0) * 7.5 + 2.0 * (r(5) - 0.5), 3.0 + 3.0 * r(7), -(gz * 7.5) + 2.0 * (r(6) - 0.5)))
                p, keep = _lib.fptr(m)
                s.h.crth_register_instance(k % 8, 0xFFFF, p)
            s.h.crth_end_instances()
            s.set_camera((0.0, 16.0, 20.0), scenes._normalize((0.0, -0.5, -1.0)))
            s.render_raw(8); c = s.counters()
            hip = _lib.hip(); a, iv, ip = s.trace_args(); fp = C.POINTER(C.c_float)
            args = (C.byref(a), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp))
            for _ in range(5): hip.crt_render(*args, 4)
            hip.crt_sync(); t0 = time.perf_counter()
            for _ in range(40): hip.crt_render(*args, 4)
            hip.crt_sync(); dt = (time.perf_counter() - t0) / 40
            print(f"{n:4d} large instances, {'tree  ' if tlas == '1' else 'linear'}: {dt * 1e3:7.3f} ms/frame, {c['rays'] / dt / 1e9:6.2f} Gray/s, primary hits {c['secondary']}, inner visits/ray {c['innerVisits'] / c['rays']:.1f}")
