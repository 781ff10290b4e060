#!/usr/bin/env python3
"""Predict the N-GPU line of bench.py on ONE GPU: render every rank's share of the 3840x2160 frame (16-row bands,
rank r of N) one after the other and combine total rays / slowest rank. Run on the GPU box."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clraytracer_amd import _lib, driver, scenes  # noqa: E402

W, H = 3840, 2160
with driver.Session(W, H, device=0) as s:
    s.load_scene(scenes.get("multi-1M"))
    targs, iv, ip = s.trace_args()
    fp = C.POINTER(C.c_float)
    a = (C.byref(targs), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp))
    hip = _lib.hip()
    for n in (1, 2, 4, 8):
        times, rays = [], 0
        for r in range(n):
            s.set_row_bands(16, r, n)
            hip.crt_render(*a, 8); rays += s.counters()["rays"]
            for _ in range(5):
                hip.crt_render(*a, 0)
            t0 = time.perf_counter()
            for _ in range(30):
                hip.crt_render(*a, 0)
            times.append((time.perf_counter() - t0) / 30)
        print(f"N={n}: per-rank ms/frame {np.round(np.array(times) * 1e3, 3)} -> predicted {rays / max(times) / 1e6:.0f} Mrays/s "
              f"({rays / max(times) / 1e6 / n:.0f} per GPU), imbalance max/mean {max(times) / np.mean(times):.3f}")
