"""Config 1: the CPU_RayCast mirror (CPURayTrace.cpp:186-249) over a 640x480 frame of cornell-1k, against
the oracle's restatement -- HitRecords must be identical field for field."""
import numpy as np

from clraytracer_amd import driver, scenes
import oracle_lib


def test_cpu_raycast_640x480_matches_oracle(nthreads):
    sc = scenes.get("cornell-1k")
    w, h = 640, 480
    with driver.Session(w, h, host_only=True) as s:
        s.load_scene(sc)
        iv, ip, pos = s.camera()
        orc = oracle_lib.Oracle(s.arenas(), nthreads=nthreads)
        rays = orc.raygen(w, h, iv, ip).reshape(-1, 3)
        origins = np.tile(pos, (len(rays), 1)).astype(np.float32)
        got = s.cpu_raycast(origins, rays, nthreads=nthreads)
        ref = orc.cpu_raycast(origins, rays)
    hit = ref["distance"] < 1e29
    assert 30000 < hit.sum() < w * h
    assert np.array_equal(got["index"], ref["index"]) and np.array_equal(got["color"], ref["color"])
    for f in ("distance", "normal", "uv"):
        assert np.array_equal(np.ascontiguousarray(got[f]).view(np.uint32), np.ascontiguousarray(ref[f]).view(np.uint32)), f
    # misses carry the skybox texel, the default normal and distance 1e30 (CPURayTrace.cpp:217-224)
    assert np.all(ref["distance"][~hit] == np.float32(1e30)) and np.all(ref["normal"][~hit] == [0, 1, 0])
    assert len(np.unique(ref["color"][~hit])) > 4


def test_cpu_raycast_agrees_with_trace_oracle_on_visibility(nthreads):
    # CPU_RayCast and the Trace kernel share IntersectBVH; with IEEE reciprocals pinned on both sides the
    # closest-hit triangle of a primary ray must be the same
    sc = scenes.get("tiny")
    w, h = 96, 64
    with driver.Session(w, h, host_only=True) as s:
        s.load_scene(sc)
        iv, ip, pos = s.camera()
        orc = oracle_lib.Oracle(s.arenas(), nthreads=nthreads)
        rays = orc.raygen(w, h, iv, ip).reshape(-1, 3)
        origins = np.tile(pos, (len(rays), 1)).astype(np.float32)
        rec = s.cpu_raycast(origins, rays)
        hits, _ = orc.closest_hits(origins, rays)
    hit = hits["instance"] >= 0
    assert np.array_equal(hit, rec["distance"] < 1e29)
    assert np.array_equal(hits["t"][hit].view(np.uint32), rec["distance"][hit].view(np.uint32)) or np.allclose(hits["t"][hit], rec["distance"][hit], rtol=1e-5)
