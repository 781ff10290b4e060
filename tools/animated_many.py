#!/usr/bin/env python3
"""401 instances (upstream's limit, Renderer.hpp:16) of tiny's two meshes on a grid, every one re-uploaded before every frame (upstream's Engine_Tick ->
SetMeshPosition -> dirty range -> clEnqueueWriteBuffer, Renderer.cpp:268-298,312-320) against the static scene, frames in flight and synchronous;
plus the host time of crt_upload_instances (memcpy + bounding spheres + cull ranges + median-split instance tree). Run on the GPU box."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clraytracer_amd import _lib, driver, scenes

W, H = 1920, 1080
tiny = scenes.get("tiny")
hip = _lib.hip()
with driver.Session(W, H, device=0) as s:
    s.load_scene(tiny)
    s.h.crth_begin_instances()
    for k in range(len(tiny.instances), 401):
        m = scenes._trs(0.6 + 0.1 * (k % 5), (0.3, 1.0, 0.2), 0.37 * k, (float((k % 21) - 10) * 6.0, float((k // 21) - 9) * 6.0, -float(k % 7) * 2.0))
        pm, keep = _lib.fptr(m)
        s.h.crth_register_instance(k % 2, 0xFFFF, pm)
    s.h.crth_end_instances()
    s.set_camera((0.0, 0.0, 23.0 * 6.0), scenes._normalize((0.0, 0.0, -1.0)))
    s.render_raw(8); rays = s.counters()["rays"]
    a, iv, ip = s.trace_args(); fp = C.POINTER(C.c_float)
    args = (C.byref(a), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp))
    inst = s.arenas()["instances"].copy()
    K = 200

    def run(animated, flags):
        for _ in range(6):
            hip.crt_render(*args, flags)
        hip.crt_sync(); t_up = 0.0; t0 = time.perf_counter()
        for k in range(K):
            if animated:
                inst["inv"][:, 3, 1] += np.float32(1e-4)
                tu = time.perf_counter(); hip.crt_upload_instances(inst.ctypes.data, 0, len(inst)); t_up += time.perf_counter() - tu
            hip.crt_render(*args, flags)
        hip.crt_sync()
        return (time.perf_counter() - t0) / K, t_up / K
    out = []
    for name, animated, flags in (("static in flight", False, 4), ("animated in flight", True, 4), ("static synchronous", False, 0), ("animated synchronous", True, 0)):
        dt, up = run(animated, flags)
        out.append(f"{name}: {dt * 1e3:.4f} ms ({rays / dt / 1e9:.2f} Gray/s)" + (f", upload {up * 1e6:.0f} us" if animated else ""))
    print("401 instances | " + " | ".join(out))
