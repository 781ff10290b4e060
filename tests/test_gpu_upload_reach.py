"""The reach of bounce-ray origins comes from the triangle pool, not from uploaded boxes (own module: one device session at a time;
tests/test_gpu_cull_bound.py keeps a module-wide session open). Reference: the per-ray instance loop without any cull, kernel_main.cl:198-217."""
import ctypes as C

import numpy as np
import pytest

from clraytracer_amd import _lib, driver, scenes
import oracle_lib
from util import bits

pytestmark = pytest.mark.gpu


def _cull_range(s, n):
    lim = np.zeros(n, np.float32)
    scene_lim = C.c_float(); reach = C.c_float(); frames = C.c_uint64()
    _lib.check(s.hip.crt_get_cull_range(lim.ctypes.data_as(C.POINTER(C.c_float)), n, C.byref(scene_lim), C.byref(reach), C.byref(frames)), "crt_get_cull_range")
    return lim, float(scene_lim.value), float(reach.value), int(frames.value)


def test_bounce_reach_follows_the_triangles_not_the_boxes(nthreads):
    """nodes that arrive through crt_upload_bvh_nodes need not bound their triangles, and bounce rays start at object-space hit
    points (hazard H6) -- so the reach the cull's proven range is checked against comes from the triangle pool itself (a device reduction over the
    pool). It grows with a far vertex (instances whose O_i it passes stop being culled), shrinks back when that vertex is overwritten (the whole
    pool is reduced again), still covers the triangles when the uploaded boxes are shrunk around them -- and the frame equals the oracle's each time."""
    sc = scenes.get("tiny")
    with driver.Session(240, 136, device=0) as s:
        s.load_scene(sc)
        a = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in s.arenas().items()}
        n_inst = len(a["instances"])

        def frame_equals_oracle(arenas):
            orc = oracle_lib.Oracle(arenas, nthreads=nthreads)
            iv, ip, pos = s.camera()
            want, st = orc.trace(orc.raygen(s.width, s.height, iv, ip), pos, sc.sun_angle)
            s.render_raw(8)
            assert s.counters() == st
            assert ((bits(s.read_output()) != bits(want)).any(axis=2)).sum() <= 2
        lim0, _, reach0, _ = _cull_range(s, n_inst)
        far_vertex = float(np.sqrt(max((a["tris"][k][:, :3].astype(np.float64) ** 2).sum(axis=1).max() for k in ("v0", "v1", "v2"))))
        assert reach0 >= far_vertex and (lim0 > 0).any()                      # the boxes of a BuildBVH tree bound the triangles: the usual case
        frame_equals_oracle(a)
        # (1) one far triangle appended to the pool (no node refers to it): the reach must cover it; small instances are no longer culled
        n_tris = len(a["tris"])
        extra = a["tris"][:1].copy()
        extra["v0"][0, :3] = (1.0e5, 0.0, 0.0); extra["v1"][0, :3] = (1.0e5, 1.0, 0.0); extra["v2"][0, :3] = (1.0e5, 0.0, 1.0)
        assert s.hip.crt_upload_triangles(extra.ctypes.data, n_tris * 80, 80) == 0
        lim1, _, reach1, _ = _cull_range(s, n_inst)
        assert reach1 >= 1.0e5 and (lim1 <= lim0).all() and (lim1[lim0 > 0] == 0).any(), (reach1, lim0, lim1)
        frame_equals_oracle(a)
        # (2) the far triangle overwritten by a near one: the reduction runs over the whole pool again and the reach comes back
        assert s.hip.crt_upload_triangles(a["tris"][:1].ctypes.data, n_tris * 80, 80) == 0
        lim2, _, reach2, _ = _cull_range(s, n_inst)
        assert abs(reach2 - reach0) <= 1e-3 * reach0 and np.array_equal(lim2, lim0)
        frame_equals_oracle(a)
        # (3) every box shrunk to a tenth around its centre: boxes no longer bound their triangles, the reach still does
        nodes = a["nodes"].copy()
        c = 0.5 * (nodes["min"][:, :3] + nodes["max"][:, :3]); h = 0.05 * (nodes["max"][:, :3] - nodes["min"][:, :3])
        nodes["min"][:, :3] = c - h; nodes["max"][:, :3] = c + h
        assert s.hip.crt_upload_bvh_nodes(nodes.ctypes.data, 0, nodes.nbytes) == 0
        _, _, reach3, _ = _cull_range(s, n_inst)
        assert reach3 >= far_vertex
        b = dict(a); b["nodes"] = nodes
        frame_equals_oracle(b)
