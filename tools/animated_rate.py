#!/usr/bin/env python3
"""Frames per second of multi-1M 1920x1080 with every instance moved before every frame (crt_upload_instances of the
whole table per frame), frames in flight, against the static scene. Run on the GPU box."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clraytracer_amd import _lib, driver, scenes

sc = scenes.get("multi-1M")
with driver.Session(1920, 1080, device=0) as s:
    s.load_scene(sc)
    hip = _lib.hip()
    s.render_raw(8); rays = s.counters()["rays"]
    a, iv, ip = s.trace_args(); fp = C.POINTER(C.c_float)
    args = (C.byref(a), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp))
    inst = s.arenas()["instances"].copy()
    K = 200
    def run(animated, flags):
        for _ in range(6): hip.crt_render(*args, flags)
        hip.crt_sync(); t0 = time.perf_counter()
        for k in range(K):
            if animated:
                inst["inv"][:, 3, 1] += np.float32(0.001)          # every instance drifts a little
                hip.crt_upload_instances(inst.ctypes.data, 0, len(inst))
            hip.crt_render(*args, flags)
        hip.crt_sync()
        return (time.perf_counter() - t0) / K
    for name, animated, flags in (("static, frames in flight", False, 4), ("animated (16 instances re-uploaded per frame), frames in flight", True, 4),
                                  ("static, synchronous", False, 0), ("animated, synchronous", True, 0)):
        dt = run(animated, flags)
        print(f"{name:70s}: {dt * 1e3:6.3f} ms/frame  ~{rays / dt / 1e9:5.2f} Gray/s")
