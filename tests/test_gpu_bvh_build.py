"""crt_build_bvh: BuildBVH (BVH.cpp:218-255) on the device must write the same bytes as the sequential builder -- the
reordered triangles with their centroid lanes, the nodes (numbered by the recursion's allocation order), the roots --
including the degenerate cases: identical centroids, empty bins, partitions that move nothing or everything (the
node stays a leaf but its triangles stay permuted), signed zeros in the bounds, big leaves."""
import ctypes as C
import time

import numpy as np
import pytest

from clraytracer_amd import _lib, driver, scenes
import oracle_lib
from test_bvh import random_tris, check_invariants
from util import bits

pytestmark = pytest.mark.gpu


def device_build(hip, tris, counts, first_tri=0, first_node=0, first_mesh=0):
    t = np.ascontiguousarray(tris)
    c = np.ascontiguousarray(counts, np.uint32)
    assert hip.crt_upload_triangles(t.ctypes.data, first_tri * 80, t.nbytes) == 0
    used = C.c_uint32(0)
    t0 = time.perf_counter()
    rc = hip.crt_build_bvh(first_tri, c.ctypes.data, len(c), first_node, first_mesh, C.byref(used))
    dt = time.perf_counter() - t0
    assert rc == 0, rc
    out_t = np.zeros(len(t), _lib.TRI_DTYPE)
    out_n = np.zeros(used.value, _lib.NODE_DTYPE)
    out_r = np.zeros(len(c), np.uint32)
    assert hip.crt_download_triangles(out_t.ctypes.data, first_tri * 80, out_t.nbytes) == 0
    assert hip.crt_download_bvh_nodes(out_n.ctypes.data, first_node * 32, out_n.nbytes) == 0
    assert hip.crt_download_bvh_roots(out_r.ctypes.data, first_mesh, len(c)) == 0
    return out_t, out_n, out_r, used.value, dt


def special_tris(kind, n, seed):
    rng = np.random.RandomState(seed)
    t = random_tris(n, seed)
    if kind == "same-centroid":          # boundsMax == boundsMin on every axis: never split, one big leaf
        for k in ("v0", "v1", "v2"):
            t[k] = t[k][0]
    elif kind in ("degenerate-right", "degenerate-left"):
        # no candidate plane (one centroid) but n * area > 1e30, so the cost test cannot refuse: upstream "splits" at plane 0 of axis 0,
        # every triangle lands on one side, the node stays a leaf and its triangles stay permuted (BVH.cpp:194)
        sgn = np.float32(1.0 if kind == "degenerate-right" else -1.0)
        t["v0"][:] = np.array([3e15, 1.0, 2.0], np.float32) * sgn; t["v1"][:] = np.array([-1e15, 0.0, 1.0], np.float32) * sgn
        t["v2"][:] = np.array([5e14, 2.0, -3.0], np.float32) * sgn
    elif kind == "signed-zeros":         # bounds that are +0 / -0 depending on the order of the fold
        z = np.where(rng.rand(n) < 0.5, np.float32(0.0), np.float32(-0.0))
        t["v0"][:, 0] = z; t["v1"][:, 0] = -z; t["v2"][:, 0] = np.abs(rng.normal(size=n)).astype(np.float32)
        t["v0"][:, 1] = -np.abs(t["v0"][:, 1]); t["v1"][:, 1] = -np.abs(t["v1"][:, 1]); t["v2"][:, 1] = np.where(rng.rand(n) < 0.5, 0.0, -0.0)
    elif kind == "two-clusters":         # all centroids in two points: most bins empty, partitions of equal elements
        a = rng.rand(n) < 0.3
        for k in ("v0", "v1", "v2"):
            t[k][a] = t[k][0]; t[k][~a] = t[k][1]
    elif kind == "grid":                 # many exactly equal coordinates (split planes through vertices)
        for k in ("v0", "v1", "v2"):
            t[k] = np.round(t[k])
    elif kind == "sorted":               # already ordered along x: partitions where the front is all-left
        o = np.argsort(t["v0"][:, 0] + t["v1"][:, 0] + t["v2"][:, 0])
        t = t[o]
    elif kind == "reversed":
        o = np.argsort(-(t["v0"][:, 0] + t["v1"][:, 0] + t["v2"][:, 0]))
        t = t[o]
    return np.ascontiguousarray(t)


CASES = [("random", [1], 1), ("random", [2], 2), ("random", [3, 5, 7], 3), ("random", [500], 4), ("random", [300, 1, 200], 5),
         ("random", [4000, 2500, 1], 7), ("random", [70000], 8), ("same-centroid", [200], 9), ("signed-zeros", [3000], 10),
         ("two-clusters", [1000], 11), ("grid", [5000, 3000], 12), ("sorted", [6000], 13), ("reversed", [6000], 14),
         # nodes cut into chunks (> 1024 triangles): leaf by the cost test, zero bounds, empty bins, the failed partition, deep levels of chunks
         ("same-centroid", [5000], 15), ("signed-zeros", [20000], 16), ("two-clusters", [9000], 17), ("degenerate-right", [5000], 18),
         ("degenerate-left", [5000], 19), ("degenerate-right", [300, 6, 1025], 20), ("grid", [40000, 1025, 1024], 21), ("random", [200000, 9, 8], 22)]


@pytest.fixture
def session():
    with driver.Session(64, 48, device=0) as s:
        yield s


@pytest.mark.parametrize("kind,counts,seed", CASES)
def test_device_build_bit_identical(session, kind, counts, seed):
    hip = _lib.hip()
    tris = special_tris(kind, sum(counts), seed)
    ot, on, oroots, ou = oracle_lib.build_bvh(tris, counts)
    dt_, dn, dr, du, _ = device_build(hip, tris, counts)
    assert du == ou and np.array_equal(dr, oroots)
    assert dt_.tobytes() == ot.tobytes()
    assert dn.tobytes() == on.tobytes()
    if kind.startswith("degenerate"):
        assert du == len(counts) and ot.tobytes() != tris.tobytes()   # one leaf per mesh, and the failed partition did move triangles
    if kind == "random" and sum(counts) > 100:
        check_invariants(dn, dr, dt_, counts)


def test_device_build_with_offsets(session):
    """A second push: triangles, nodes and roots appended behind an existing mesh (absolute indices)."""
    hip = _lib.hip()
    t0 = special_tris("random", 900, 31); t1 = special_tris("grid", 2100, 32)
    a_t, a_n, a_r, a_u, _ = device_build(hip, t0, [900])
    b_t, b_n, b_r, b_u, _ = device_build(hip, t1, [1500, 600], first_tri=900, first_node=a_u, first_mesh=1)
    ot, on, oroots, ou = oracle_lib.build_bvh(t1, [1500, 600], counter_start=a_u)
    on = on[a_u:].copy()
    leaf = on["triCount"] > 0
    on["leftFirst"][leaf] += 900                              # the oracle indexes triangles from its own pointer
    assert b_u == ou and np.array_equal(b_r, oroots)
    assert b_t.tobytes() == ot.tobytes() and b_n.tobytes() == on.tobytes()
    again = np.zeros(a_u, _lib.NODE_DTYPE)
    assert hip.crt_download_bvh_nodes(again.ctypes.data, 0, again.nbytes) == 0 and again.tobytes() == a_n.tobytes()


@pytest.mark.parametrize("name", ["tiny", "cornell-1k", "sponza-class-250k", "multi-1M"])
def test_scene_built_on_device_renders_identically(name, nthreads):
    sc = scenes.get(name)
    with driver.Session(320, 184, device=0) as s:
        s.load_scene(sc)                                       # host BuildBVH
        s.render_raw(0)
        ref = s.read_output()
        a = s.arenas()
        hip = _lib.hip()
        H = _lib.host()
        counts = []
        for m in range(H.crth_num_meshes()):
            info = np.zeros(4, np.uint32)
            H.crth_mesh_info(m, info.ctypes.data)
            counts.append(int(info[0]))
        starts = np.concatenate([[0], np.cumsum(counts)])
        assert starts[-1] == len(a["tris"])
        # the host arena is already reordered; shuffle every mesh's triangles to get a different input order and
        # compare the device build of THAT input with the oracle's build of the same input
        rng = np.random.RandomState(3)
        tris = a["tris"].copy()
        for m in range(len(counts)):
            seg = tris[starts[m]:starts[m + 1]]
            tris[starts[m]:starts[m + 1]] = seg[rng.permutation(len(seg))]
        tris["cx"] = 0; tris["cy"] = 0; tris["cz"] = 0
        ot, on, oroots, ou = oracle_lib.build_bvh(tris, counts)
        d_t, d_n, d_r, d_u, dt = device_build(hip, tris, counts)
        print(f"{name}: device BuildBVH of {len(tris)} triangles / {d_u} nodes in {dt * 1e3:.1f} ms")
        assert d_u == ou and np.array_equal(d_r, oroots) and d_t.tobytes() == ot.tobytes() and d_n.tobytes() == on.tobytes()
        # and the renderer now runs on the device-built tree: compare with the oracle tracing the same arenas
        b = dict(a); b.update(tris=ot, nodes=on, roots=oroots)
        orc = oracle_lib.Oracle(b, nthreads=nthreads)
        iv, ip, pos = s.camera()
        want, st = orc.trace(orc.raygen(s.width, s.height, iv, ip), pos, sc.sun_angle)
        s.render_raw(8)
        assert np.array_equal(bits(s.read_output()), bits(want)) and s.counters() == st


@pytest.mark.parametrize("name", ["tiny", "sponza-class-250k", "multi-1M"])
def test_mirrored_resource_manager_with_device_build(name):
    """ResourceManager::SetDeviceBVHBuild(true): PushMeshesToGPU builds on the device and fills the host arenas from
    it; arenas and frames equal those of the host build."""
    sc = scenes.get(name)
    with driver.Session(256, 144, device=0) as s:
        t0 = time.perf_counter(); s.load_scene(sc); t_host = time.perf_counter() - t0
        s.render_raw(0)
        ref = s.read_output()
        a = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in s.arenas().items()}
    with driver.Session(256, 144, device=0) as s:
        t0 = time.perf_counter(); s.load_scene(sc, device_bvh_build=True); t_dev = time.perf_counter() - t0
        b = s.arenas()
        for k in ("tris", "nodes", "roots", "instances"):
            assert a[k].tobytes() == b[k].tobytes(), k
        s.render_raw(0)
        assert np.array_equal(bits(s.read_output()), bits(ref))
        print(f"{name}: load_scene {t_host * 1e3:.0f} ms with the host BuildBVH, {t_dev * 1e3:.0f} ms with crt_build_bvh")


def test_build_errors_are_codes(session):
    hip = _lib.hip()
    tris = special_tris("random", 300, 41)
    assert hip.crt_upload_triangles(tris.ctypes.data, 0, tris.nbytes) == 0
    used = C.c_uint32(0)
    ok = np.array([300], np.uint32); zero = np.array([100, 0, 200], np.uint32); big = np.array([10 ** 7], np.uint32)
    assert hip.crt_build_bvh(0, None, 1, 0, 0, C.byref(used)) == -2
    assert hip.crt_build_bvh(0, zero.ctypes.data, 3, 0, 0, C.byref(used)) == -2            # an empty mesh has no root leaf
    assert hip.crt_build_bvh(0, big.ctypes.data, 1, 0, 0, C.byref(used)) == -2             # triangles that were never uploaded
    assert hip.crt_build_bvh(0, ok.ctypes.data, 1, 0, 128, C.byref(used)) == -3            # mesh table is 128 entries
    assert hip.crt_build_bvh(0, ok.ctypes.data, 1, 2400000 - 10, 0, C.byref(used)) == -3    # node pool
    assert hip.crt_build_bvh(0, ok.ctypes.data, 1, 0, 0, None) == 0                         # nodesUsedOut is optional
    buf = np.zeros(64, np.uint8)
    assert hip.crt_download_triangles(buf.ctypes.data, 0, 81) == -2 and hip.crt_download_bvh_nodes(buf.ctypes.data, 0, 33) == -2
    assert hip.crt_download_bvh_roots(buf.ctypes.data, 127, 2) == -3 and hip.crt_download_triangles(None, 0, 80) == -2


def test_a_refused_device_build_falls_back_to_the_host_builder(monkeypatch, capfd):
    """ResourceManager::PushMeshesToGPU with SetDeviceBVHBuild(true): when crt_build_bvh refuses (forced here through the
    CRT_DEBUG_HOOKS test hook; in production: a size beyond the builder's scratch layout or a failed consistency check), the
    host BuildBVH (BVH.cpp:218-255 mirror) takes over -- same arenas, same frame as a session that never asked for the device build."""
    sc = scenes.get("tiny")
    with driver.Session(160, 96, device=0) as s:
        s.load_scene(sc)
        a = s.arenas()
        want_nodes, want_tris = a["nodes"].tobytes(), a["tris"].tobytes()
        s.render_raw(0); want = s.read_output().copy()
    monkeypatch.setenv("CRT_DEBUG_HOOKS", "1")
    # refusal codes: OUT_OF_RANGE (1 -> -3), the builder's scratch not fitting (hipErrorOutOfMemory = 2), BAD_ARGUMENT (-2) -- ADVICE r5
    for code in ("1", "2", "-2"):
        monkeypatch.setenv("CRT_DEBUG_FAIL_BVH_BUILD", code)
        with driver.Session(160, 96, device=0) as s:
            s.load_scene(sc, device_bvh_build=True)
            a = s.arenas()
            assert a["nodes"].tobytes() == want_nodes and a["tris"].tobytes() == want_tris, code
            s.render_raw(0)
            assert np.array_equal(s.read_output().view(np.uint32), want.view(np.uint32)), code
        assert "building on the host" in capfd.readouterr().err, code
    # a sticky HIP error (719 = hipErrorLaunchFailure) is NOT papered over: PushMeshesToGPU reports it, nothing of the push counts as built
    monkeypatch.setenv("CRT_DEBUG_FAIL_BVH_BUILD", "719")
    with driver.Session(160, 96, device=0) as s:
        with pytest.raises(driver.CrtError) as ei:
            s.load_scene(sc, device_bvh_build=True)
        assert "719" in str(ei.value)
        assert s.h.crth_last_error() == 719 and s.h.crth_num_nodes() == 0          # lastError set, node arena untouched
