#!/usr/bin/env python3
"""Predict the N-GPU line of bench.py on ONE GPU: render every rank's share of the 3840x2160 frame (16-row bands,
rank r of N) one after the other and combine total rays / slowest rank. Run on the GPU box.
CRT_FLIGHT=2 renders with two frames in flight (CRT_RENDER_ASYNC); default 1 = synchronous frames."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clraytracer_amd import _lib, driver, scenes  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
FLIGHT = int(os.environ.get("CRT_FLIGHT", "1"))
if FLIGHT > 1:
    os.environ["CRT_FRAMES_IN_FLIGHT"] = str(FLIGHT)
FLAGS = 4 if FLIGHT > 1 else 0
with driver.Session(W, H, device=0) as s:
    s.load_scene(scenes.get("multi-1M"))
    targs, iv, ip = s.trace_args()
    fp = C.POINTER(C.c_float)
    a = (C.byref(targs), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp))
    hip = _lib.hip()
    for n in [int(x) for x in os.environ.get("CRT_PRED_N", "1,2,4,8").split(",")]:
        times, rays = [], 0
        for r in range(n):
            s.set_row_bands(int(os.environ.get("CRT_BAND", "16")), r, n)
            hip.crt_render(*a, 8); rays += s.counters()["rays"]
            for _ in range(6):
                hip.crt_render(*a, FLAGS)
            hip.crt_sync()
            t0 = time.perf_counter()
            for _ in range(40):
                hip.crt_render(*a, FLAGS)
            hip.crt_sync()
            times.append((time.perf_counter() - t0) / 40)
        print(f"flight={FLIGHT} {W}x{H} N={n}: per-rank ms/frame {np.round(np.array(times) * 1e3, 3)} -> predicted {rays / max(times) / 1e6:.0f} Mrays/s "
              f"({rays / max(times) / 1e6 / n:.0f} per GPU), imbalance max/mean {max(times) / np.mean(times):.3f}")
