#!/usr/bin/env python3
"""Time crt_build_bvh (device BuildBVH) against the host builder on a scene. Run on the GPU box.
    python tools/bvh_build_time.py [scene]
    CRT_DEBUG_HOOKS=1 CRT_DEBUG_BVH_REPLAY=1 BVH_BUILDS=6 python tools/bvh_build_time.py   the floor of a one-submission build (crt_bvh_driver.h)"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clraytracer_amd import _lib, driver, scenes

name = sys.argv[1] if len(sys.argv) > 1 else "multi-1M"
sc = scenes.get(name)
with driver.Session(64, 48, device=0) as s:
    t0 = time.perf_counter(); s.load_scene(sc); t_load = time.perf_counter() - t0
    a = s.arenas()
    H, hip = _lib.host(), _lib.hip()
    counts = []
    for m in range(H.crth_num_meshes()):
        info = np.zeros(4, np.uint32); H.crth_mesh_info(m, info.ctypes.data); counts.append(int(info[0]))
    tris = a["tris"].copy()
    rng = np.random.RandomState(1)
    st = np.concatenate([[0], np.cumsum(counts)])
    for m in range(len(counts)):
        tris[st[m]:st[m + 1]] = tris[st[m]:st[m + 1]][rng.permutation(counts[m])]
    c = np.asarray(counts, np.uint32)
    nodes = np.zeros(2 * len(tris) + 64, _lib.NODE_DTYPE); roots = np.zeros(len(c), np.uint32)
    t = tris.copy(); t0 = time.perf_counter()
    used_h = H.crth_build_bvh(t.ctypes.data, c.ctypes.data, len(c), nodes.ctypes.data, roots.ctypes.data)
    t_host = time.perf_counter() - t0
    best = 1e9
    times = []
    for _ in range(int(os.environ.get("BVH_BUILDS", "3"))):
        assert hip.crt_upload_triangles(tris.ctypes.data, 0, tris.nbytes) == 0
        used = C.c_uint32(0); t0 = time.perf_counter()
        assert hip.crt_build_bvh(0, c.ctypes.data, len(c), 0, 0, C.byref(used)) == 0
        times.append(time.perf_counter() - t0)
        best = min(best, times[-1])
    nd = np.zeros(used.value, _lib.NODE_DTYPE)
    assert hip.crt_download_bvh_nodes(nd.ctypes.data, 0, nd.nbytes) == 0
    same = used.value == used_h and nd.tobytes() == nodes[:used_h].tobytes()
    if os.environ.get("CRT_DEBUG_BVH_REPLAY"):
        print(f"CRT_DEBUG_BVH_REPLAY: build 1 records every level's list sizes ({times[0] * 1e3:.2f} ms, the shipped path), builds 2.. enqueue all launches back to back "
              f"from the recording, no publish kernel, no host wait per level: {', '.join(f'{t * 1e3:.2f}' for t in times[1:])} ms; nodes identical to the host builder's: {same}")
    print(f"{name}: {len(tris)} triangles in {len(c)} meshes -> {used.value} nodes (host {used_h}); host BuildBVH {t_host * 1e3:.1f} ms, "
          f"crt_build_bvh {best * 1e3:.1f} ms (incl. relayout for rendering); whole load_scene {t_load * 1e3:.0f} ms")
