#!/usr/bin/env python3
"""Frames delivered to HOST memory (the PCIe-inclusive rate of DESIGN.md section 7): multi-1M 1920x1080 with
CRT_RENDER_READBACK, frames in flight, as float4 (33.2 MB/frame) and as RGBA8 (8.3 MB/frame), against crt_read_output
after every synchronous frame. Run on the GPU box."""
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
    ptr, n = C.c_void_p(), C.c_size_t()
    K = 60
    def timed(flags, consume):
        for _ in range(6): hip.crt_render(*args, flags)
        hip.crt_sync(); t0 = time.perf_counter()
        for _ in range(K):
            hip.crt_render(*args, flags)
            consume()
        hip.crt_sync(); hip.crt_map_host_frame(C.byref(ptr), C.byref(n)) if flags & 128 else None
        return (time.perf_counter() - t0) / K
    buf = np.empty((1080, 1920, 4), np.float32)
    rows = [("resident (no copy), frames in flight", timed(4, lambda: None)),
            ("crt_read_output after every synchronous frame (pageable)", timed(0, lambda: hip.crt_read_output(buf.ctypes.data, buf.size))),
            ("READBACK float4, frames in flight", timed(4 | 128, lambda: None)),
            ("READBACK RGBA8 (UNORM8), frames in flight", timed(4 | 128 | 64, lambda: None))]
    for name, dt in rows:
        print(f"{name:60s}: {dt * 1e3:6.3f} ms/frame  {rays / dt / 1e9:5.2f} Gray/s")
