#!/usr/bin/env python3
"""A moving camera against a static one (multi-1M, 1920x1080): the camera orbits the scene, a new view every frame (0.25 degrees
per frame, 720 frames = half an orbit). Nothing the path does may depend on the view standing still: frames in flight use
the plain tile order, synchronous frames sort their launch lists by the PREVIOUS frame's per-tile cost. Reported: ms/frame
over the orbit with the camera moving every frame, and -- the same views, each held still -- the mean of static ms/frame at
24 points of the orbit. Run on the GPU box."""
import ctypes as C
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clraytracer_amd import _lib, driver, scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "multi-1M"
sc = scenes.get(name)
hip = _lib.hip()
fp = C.POINTER(C.c_float)
with driver.Session(1920, 1080, device=0) as s:
    s.load_scene(sc)
    c0 = np.array(sc.camera_pos, np.float64)
    radius, height = math.hypot(c0[0], c0[2]), c0[1]
    ang0 = math.atan2(c0[0], c0[2])
    pitch = float(np.array(sc.camera_front, np.float64)[1])

    def view(k, step=0.25):
        a = ang0 + math.radians(step * k)
        pos = (radius * math.sin(a), height, radius * math.cos(a))
        front = scenes._normalize((-pos[0], pitch * radius, -pos[2]))
        s.set_camera(pos, front)
        targs, iv, ip = s.trace_args()
        return targs, iv, ip

    views = [view(k) for k in range(720)]
    packed = [(C.byref(t), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp)) for t, iv, ip in views]

    def moving(flags):
        for a in packed[:8]:
            hip.crt_render(*a, flags)
        hip.crt_sync(); t0 = time.perf_counter()
        for a in packed:
            hip.crt_render(*a, flags)
        hip.crt_sync()
        return (time.perf_counter() - t0) / len(packed) * 1e3

    def held(flags):
        tot = 0.0
        for a in packed[::30]:
            for _ in range(6):
                hip.crt_render(*a, flags)
            hip.crt_sync(); t0 = time.perf_counter()
            for _ in range(30):
                hip.crt_render(*a, flags)
            hip.crt_sync()
            tot += (time.perf_counter() - t0) / 30 * 1e3
        return tot / len(packed[::30])

    print(f"{name} 1920x1080, orbit of 720 views (0.25 degrees apart):")
    for label, flags in (("frames in flight", 4), ("synchronous frames", 0)):
        m, h = moving(flags), held(flags)
        print(f"  {label:20s}: camera moving every frame {m:.4f} ms/frame; the same views held still (24 of them) {h:.4f} ms/frame; moving / still = {m / h:.3f}", flush=True)
