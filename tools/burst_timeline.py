#!/usr/bin/env python3
"""Diagnostic (run on the GPU box): start / end of every frame of a short burst of frames in flight -- the driver's bench run is
20 frames after a synchronisation -- to see what the pipeline's fill and drain cost.   python tools/burst_timeline.py [frames] [scene]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clraytracer_amd import _lib, driver, scenes  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
name = sys.argv[2] if len(sys.argv) > 2 else "multi-1M"
with driver.Session(1920, 1080, device=0) as s:
    s.load_scene(scenes.get(name))
    hip = _lib.hip()
    a, iv, ip = s.trace_args()
    fp = C.POINTER(C.c_float)
    q = (C.byref(a), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp))
    for rep in range(3):
        for _ in range(5):
            hip.crt_render(*q, 4)
        hip.crt_sync()
        hip.crt_frame_time_stats(None, 1)
        for _ in range(K):
            hip.crt_render(*q, 4)
        hip.crt_sync()
        n = C.c_size_t(0)
        t = np.zeros((256, 2), np.float64)
        _lib.check(hip.crt_debug_read_frame_times(t.ctypes.data, 256, C.byref(n)), "crt_debug_read_frame_times")
        t = t[:n.value]
        t = t[np.argsort(t[:, 0])]
        print(f"burst {rep}: {n.value} frames, extent {t[:, 1].max():.3f} ms = {t[:, 1].max() / K:.4f} ms per frame")
        print("  start:", np.round(t[:, 0], 3))
        print("  end  :", np.round(t[:, 1], 3))
        ends = np.sort(t[:, 1])
        print("  gaps between completions:", np.round(np.diff(ends), 3))
