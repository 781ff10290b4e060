import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
from clraytracer_amd import _lib, driver, scenes
sc = scenes.get("multi-1M")
with driver.Session(1920, 1080, device=0) as s:
    s.load_scene(sc)
    hip = _lib.hip()
    s.set_camera((0.0, 500.0, 0.0), scenes._normalize((0.0, 1.0, 0.001)))      # far above the scene, looking up: every ray is sky
    fp = C.POINTER(C.c_float)
    for nm in (16, 0):
        a, iv, ip = s.trace_args()
        a.numMeshes = nm
        q = (C.byref(a), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp))
        for fl in (4, 0):
            for _ in range(20): hip.crt_render(*q, fl)
            hip.crt_sync(); t0 = time.perf_counter()
            for _ in range(200): hip.crt_render(*q, fl)
            hip.crt_sync(); dt = (time.perf_counter() - t0) / 200
            s.render_raw(8) if nm == 16 else None
            print(f"all-sky frame, numMeshes={nm}, {'in flight' if fl else 'synchronous'}: {dt*1e3:.4f} ms/frame")
