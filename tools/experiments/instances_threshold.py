import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
from clraytracer_amd import _lib, driver, scenes
sc = scenes.get("tiny")
W, H = 1920, 1080
for n in (16, 24, 32, 48, 64):
    for tlas in ("0", "1"):
        os.environ["CRT_TLAS"] = tlas
        with driver.Session(W, H, device=0) as s:
            s.load_scene(sc)
            s.h.crth_begin_instances()
            for k in range(len(sc.instances), n):
                m = scenes._trs(0.6 + 0.1 * (k % 5), (0.3, 1.0, 0.2), 0.37 * k, (float((k % 9) - 4) * 6.0, float((k // 9) - 3) * 6.0, -float(k % 7) * 2.0))
                p, keep = _lib.fptr(m)
                s.h.crth_register_instance(k % 2, 0xFFFF, p)
            s.h.crth_end_instances()
            s.set_camera((0.0, 0.0, 60.0), scenes._normalize((0.0, 0.0, -1.0)))
            s.render_raw(8); c = s.counters()
            hip = _lib.hip(); a, iv, ip = s.trace_args(); fp = C.POINTER(C.c_float)
            args = (C.byref(a), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp))
            for _ in range(5): hip.crt_render(*args, 4)
            hip.crt_sync(); t0 = time.perf_counter()
            for _ in range(60): hip.crt_render(*args, 4)
            hip.crt_sync(); dt = (time.perf_counter() - t0) / 60
            print(f"{n:4d} instances, {'tree  ' if tlas == '1' else 'linear'}: {dt * 1e3:7.3f} ms/frame, {c['rays'] / dt / 1e9:6.2f} Gray/s, primary hits {c['secondary']}")
