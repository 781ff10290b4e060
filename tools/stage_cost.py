#!/usr/bin/env python3
"""ms/frame of the per-pixel stages behind Trace (PostProcess, the RGBA8 target, FXAA) with frames in flight and synchronous:
what upstream's Render() -- which always runs PostProcess into an RGBA8 texture -- costs on top of the HDR float frame."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clraytracer_amd import _lib, driver, scenes  # noqa: E402

W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
name = sys.argv[1] if len(sys.argv) > 1 else "multi-1M"
hip = _lib.hip()
with driver.Session(W, H, device=0) as s:
    s.load_scene(scenes.get(name))
    targs, iv, ip = s.trace_args()
    fp = C.POINTER(C.c_float)
    a = (C.byref(targs), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp))

    def ms(flags, n=200):
        for _ in range(10):
            hip.crt_render(*a, flags)
        hip.crt_sync()
        t0 = time.perf_counter()
        for _ in range(n):
            hip.crt_render(*a, flags)
        hip.crt_sync()
        return (time.perf_counter() - t0) / n * 1e3

    print(f"{name} {W}x{H}                       in flight   synchronous")
    for label, f in (("Trace only (float4 HDR)", 0), ("+ PostProcess", 1), ("+ RGBA8 target + PostProcess (upstream's Render)", 1 | 64),
                     ("+ RGBA8 + PostProcess, bytes delivered to host memory", 1 | 64 | 128),
                     ("+ FXAA + PostProcess", 1 | 512), ("+ RGBA8 + FXAA + PostProcess", 1 | 64 | 512),
                     ("+ RGBA8 + FXAA + PostProcess, bytes delivered to host memory", 1 | 64 | 512 | 128)):
        print(f"  {label:62s} {ms(f | 4):.4f} ms   {ms(f):.4f} ms", flush=True)
