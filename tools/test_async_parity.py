#!/usr/bin/env python3
"""Quick check on the GPU box: frames rendered with two in flight are bit-identical to synchronous frames."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clraytracer_amd import _lib, driver, scenes
W, H = 640, 368
with driver.Session(W, H, device=0) as s:
    s.load_scene(scenes.get("sponza-class-250k"))
    targs, iv, ip = s.trace_args()
    fp = C.POINTER(C.c_float)
    a = (C.byref(targs), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp))
    hip = _lib.hip()
    hip.crt_render(*a, 0)
    ref = s.read_output().copy()
    for k in range(7):
        hip.crt_render(*a, 4)
    out = s.read_output()
    print("async==sync:", np.array_equal(ref.view(np.uint32), out.view(np.uint32)))
    sums = (C.c_double * 4)(); n = C.c_ulonglong()
    hip.crt_frame_time_stats(sums, C.byref(n), 1)
    print("frames timed", n.value, "trace ms sum", sums[2])
