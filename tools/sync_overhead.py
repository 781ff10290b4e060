#!/usr/bin/env python3
"""Synchronous frames (the reference's Render() + clFinish, Renderer.cpp:305-367): wall time per frame against the Trace kernel's own duration
(HIP events) -- what the host side of a frame costs: submission, start latency, completion wake-up. Run on the GPU box.
    python tools/sync_overhead.py [scene]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clraytracer_amd import _lib, driver, scenes

name = sys.argv[1] if len(sys.argv) > 1 else "multi-1M"
hip = _lib.hip()
with driver.Session(1920, 1080, device=0) as s:
    s.load_scene(scenes.get(name))
    a, iv, ip = s.trace_args(); fp = C.POINTER(C.c_float)
    args = (C.byref(a), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp))
    for flags, label in ((0, "plain"), (1 | 64, "PostProcess + RGBA8 (upstream's Render)")):
        for _ in range(20):
            hip.crt_render(*args, flags)
        K = 200
        st = _lib.CrtFrameStats()
        hip.crt_frame_time_stats(None, 1)
        t0 = time.perf_counter()
        for _ in range(K):
            hip.crt_render(*args, flags)
        wall = (time.perf_counter() - t0) / K * 1e3
        hip.crt_frame_time_stats(C.byref(st), 0)
        kern = st.sumMs[2] / st.frames; frame = st.sumMs[0] / st.frames
        print(f"{name} synchronous, {label}: wall {wall:.4f} ms per frame; Trace kernel {kern:.4f} ms, frame start -> end events {frame:.4f} ms; host + start + wake-up = {1e3 * (wall - frame):.1f} us per frame")
