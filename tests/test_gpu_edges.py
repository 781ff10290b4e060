"""Edge cases of the boundary on the GPU: ragged frame sizes, the 401-instance maximum (candidate chunks of 64),
single-leaf meshes (hazard H3), leaves with >= 128 triangles (the `bigLeaf` escape of the packed node reference),
partial instance updates, and the error paths of the upload API."""
import ctypes as C

import numpy as np
import pytest

from clraytracer_amd import _lib, driver, scenes
import oracle_lib
from util import bits, seeded_rays

pytestmark = pytest.mark.gpu


def frame_equal(s, orc, sc, flags=8):
    s.render_raw(flags)
    gpu = s.read_output()
    iv, ip, pos = s.camera()
    ref, st = orc.trace(orc.raygen(s.width, s.height, iv, ip), pos, sc.sun_angle)
    d = np.abs(gpu[..., :3].astype(np.float64) - ref[..., :3].astype(np.float64))
    assert np.sqrt(np.mean(d * d)) < 1e-4 and (d.max(-1) > 1e-5).sum() <= max(1, int(1e-5 * d[..., 0].size))
    if flags & 8:
        assert s.counters() == st
    return gpu, st


@pytest.mark.parametrize("size", [(250, 130), (17, 16), (16, 33), (641, 479)])
def test_ragged_frame_sizes(size, nthreads):
    w, h = size
    sc = scenes.get("tiny")
    with driver.Session(w, h, device=0) as s:
        s.load_scene(sc)
        orc = oracle_lib.Oracle(s.arenas(), nthreads=nthreads)
        frame_equal(s, orc, sc)
        # resize below 16 is ignored (Renderer.cpp:200), a real resize reallocates
        s.resize(8, 8)
        assert (s.width, s.height) == (w, h)
        s.resize(96, 40)
        frame_equal(s, orc, sc)


def write_mesh(tmp, name, mesh):
    return scenes._write_mesh(str(tmp), name, mesh, [((0.8, 0.6, 0.4), None)])


@pytest.fixture(params=["0", "1"], ids=["linear", "tree"])
def candidate_search(request, monkeypatch):
    """CRT_TLAS (read by crt_init) forces how a ray finds its candidate instances: the linear sphere loop or the instance
    tree, which is otherwise used above 64 instances."""
    monkeypatch.setenv("CRT_TLAS", request.param)
    return request.param


def test_max_instances_single_leaf_and_big_leaf(tmp_path, candidate_search, nthreads):
    # mesh 0: one triangle (root is a leaf: tested with no box test, never culled)
    # mesh 1: 200 identical triangles -> SAH cannot split -> one leaf with 200 triangles (bigLeaf escape)
    # mesh 2: a small icosphere (ordinary tree)
    rng = np.random.RandomState(3)
    one = scenes.Mesh(np.array([[0, 0, 0], [1, 0, 0.1], [0, 1, 0.2]], np.float32), np.zeros((3, 2), np.float32),
                      np.tile([0, 0, 1], (3, 1)).astype(np.float32), np.array([[0, 1, 2]], np.int32), np.zeros(1, np.int32))
    # 200 copies of one triangle: all centroids identical -> boundsMax == boundsMin on every axis -> no split
    pos = np.tile(np.array([[1, 0, 0.3], [-0.5, 0.8660254, -0.1], [-0.5, -0.8660254, -0.2]], np.float32), (200, 1))
    fan = scenes.Mesh(pos, np.zeros((600, 2), np.float32), np.tile([0, 0, 1], (600, 1)).astype(np.float32),
                      np.arange(600, dtype=np.int32).reshape(200, 3), np.zeros(200, np.int32))
    ico = scenes._icosphere(2, 0.8)
    paths = [write_mesh(tmp_path, "one", one), write_mesh(tmp_path, "fan", fan), write_mesh(tmp_path, "ico", ico)]
    sky = str(tmp_path / "sky.ppm")
    scenes.write_ppm(sky, scenes._skybox(64, 32))
    insts = []
    for k in range(401):       # Renderer::MaxNumInstances (Renderer.hpp:16); > 64 exercises the candidate chunks
        t = (float((k % 21) - 10) * 2.2, float((k // 21) - 9) * 2.2, -float(k % 7))
        insts.append(scenes.Instance(k % 3, 0xFFFF, scenes._trs(0.5 + (k % 5) * 0.2, (0.3, 1.0, 0.2), 0.37 * k, t)))
    sc = scenes.Scene("edge-401", str(tmp_path), sky, paths, insts, (0.0, 0.0, 40.0), (0.0, 0.0, -1.0))
    with driver.Session(320, 200, device=0) as s:
        s.load_scene(sc)
        a = s.arenas()
        assert len(a["instances"]) == 401
        nodes = a["nodes"]
        assert (nodes["triCount"] >= 128).any() and nodes["triCount"][a["roots"][0]] == 1
        orc = oracle_lib.Oracle(a, nthreads=nthreads)
        gpu, st = frame_equal(s, orc, sc)
        assert st["hits"] > 1000 and st["traversals"] == st["rays"] * 401
        iv, ip, pos_ = s.camera()
        o, d = seeded_rays(a, pos_, 16384, seed=5)
        got = s.query_hits(o, d)
        ref, hst = orc.closest_hits(o, d)
        assert got.tobytes() == ref.tobytes() and s.counters() == hst
        # a 402nd instance is refused (the reference exits, Renderer.cpp:229)
        p, keep = _lib.fptr(np.eye(4, dtype=np.float32))
        assert s.h.crth_register_instance(0, 0xFFFF, p) == 0xFFFFFFFF
        assert s.h.crth_last_error() == -3


def test_partial_instance_update_and_material_override(nthreads):
    sc = scenes.get("tiny")
    with driver.Session(160, 96, device=0) as s:
        s.load_scene(sc)
        s.render(postprocess=False)
        before = s.output().copy()
        m = sc.instances[1].matrix.copy(); m[3, :3] += np.array([1.5, -0.5, 0.25], np.float32)
        p, keep = _lib.fptr(m)
        s.h.crth_set_mesh_matrix(1, p)                 # dirty range -> partial upload on the next Render (Renderer.cpp:288-320)
        s.h.crth_set_instance_material(0, 0)           # NoneMaterial: default material 0
        s.render(postprocess=False)
        after = s.output()
        assert not np.array_equal(before, after)
        orc = oracle_lib.Oracle(s.arenas(), nthreads=nthreads)
        iv, ip, pos = s.camera()
        ref, _ = orc.trace(orc.raygen(s.width, s.height, iv, ip), pos, sc.sun_angle)
        assert np.array_equal(bits(after), bits(ref))


def test_upload_errors_are_codes_not_crashes():
    hip = _lib.hip()
    sc = scenes.get("tiny")
    with driver.Session(64, 48, device=0) as s:
        s.load_scene(sc)
        buf = np.zeros(1024, np.uint8)
        assert hip.crt_upload_triangles(buf.ctypes.data, 0, 81) == -2                       # not a multiple of 80
        assert hip.crt_upload_triangles(buf.ctypes.data, 80 * 2400000, 80) == -3             # beyond the pool
        assert hip.crt_upload_materials(buf.ctypes.data, 250, 10) == -3
        assert hip.crt_upload_texture_table(buf.ctypes.data, 33) == -3
        assert hip.crt_upload_instances(None, 0, 1) == -2
        assert hip.crt_upload_bvh_roots(buf.ctypes.data, 127, 2) == -3
        assert hip.crt_set_row_bands(12, 0, 1) == -2 and hip.crt_set_row_bands(16, 2, 2) == -2
        assert hip.crt_read_output(buf.ctypes.data, 5) == -2
        bad = np.full(2, 0xFFFF, np.uint16)
        inst = np.zeros(1, _lib.INSTANCE_DTYPE); inst["meshIndex"] = 500
        assert hip.crt_upload_instances(inst.ctypes.data, 0, 1) == -2                        # mesh index out of range
        # a cyclic / out-of-range BVH is rejected at upload and rendering is refused until it is fixed
        a = s.arenas()
        nodes = a["nodes"].copy()
        good = nodes.copy()
        inner = np.where(nodes["triCount"] == 0)[0]
        nodes["leftFirst"][inner[3]] = inner[0]                                              # child before parent: cycle
        assert hip.crt_upload_bvh_nodes(nodes.ctypes.data, 0, nodes.nbytes) == -2
        args, iv, ip = s.trace_args()
        fp = C.POINTER(C.c_float)
        assert hip.crt_render(C.byref(args), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp), 0) == -2
        assert hip.crt_upload_bvh_nodes(good.ctypes.data, 0, good.nbytes) == 0
        assert hip.crt_render(C.byref(args), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp), 0) == 0


def test_feedback_launch_lists_never_change_pixels(nthreads):
    """The megakernel reorders (and splits) tiles by the previous frame's per-tile cost. Pixels must not depend on it:
    first frame (identity order), steady state, and frames rendered with costs that are stale because the camera,
    the frame size or the band ownership changed in between."""
    sc = scenes.get("tiny")
    with driver.Session(200, 120, device=0) as s:
        s.load_scene(sc)
        orc = oracle_lib.Oracle(s.arenas(), nthreads=nthreads)

        def check():
            s.render_raw(0)
            gpu = s.read_output()
            iv, ip, pos = s.camera()
            ref, _ = orc.trace(orc.raygen(s.width, s.height, iv, ip), pos, sc.sun_angle)
            assert np.array_equal(bits(gpu), bits(ref))

        for _ in range(4):                       # identity order, then lists built from real costs (with split tiles)
            check()
        s.render_raw(8)                          # steady state: every tile exactly once (or as four quadrants) -> counters match
        iv, ip, pos = s.camera()
        _, st = orc.trace(orc.raygen(s.width, s.height, iv, ip), pos, sc.sun_angle)
        assert s.counters() == st
        for k in range(3):                       # moving camera: every frame runs on the previous view's costs
            s.set_camera((1.5 * k - 2.0, 9.0 + k, 12.0 - k), scenes._normalize((0.1 * k, -0.45, -1.0)))
            check()
        s.resize(96, 64); check(); check()       # geometry change resets the lists
        s.set_row_bands(16, 1, 2)
        s.render_raw(0)
        part = s.read_output()
        s.set_row_bands(16, 0, 1)
        s.render_raw(0)
        full = s.read_output()
        own = np.array([_lib.hip().crt_row_owner(y, 16, 2) == 1 for y in range(s.height)])
        assert np.array_equal(bits(part[own]), bits(full[own]))


@pytest.mark.parametrize("slots", [3, 8, 1])
def test_frames_in_flight_match_synchronous_frames(nthreads, monkeypatch, slots):
    """CRT_RENDER_ASYNC frames rotate over the frame slots (own stream / output buffer / launch lists; 3 by default, 8 is
    what bench.py asks for at 8 ranks, 1 degenerates to queued frames on one stream). Pixels must be those of a synchronous
    frame; uploads between ASYNC frames must be ordered after the frames already queued and before the next one; the
    accumulated event timing covers every frame."""
    monkeypatch.setenv("CRT_FRAMES_IN_FLIGHT", str(slots))                # read by crt_init
    sc = scenes.get("tiny")
    ASYNC = 4
    with driver.Session(200, 120, device=0) as s:
        s.load_scene(sc)
        orc = oracle_lib.Oracle(s.arenas(), nthreads=nthreads)
        hip = _lib.hip()

        def oracle_frame():
            iv, ip, pos = s.camera()
            return orc.trace(orc.raygen(s.width, s.height, iv, ip), pos, sc.sun_angle)[0]

        ref0 = oracle_frame()
        assert hip.crt_frame_time_stats(None, 1) == 0
        for _ in range(7):
            s.render_raw(ASYNC)
        assert np.array_equal(bits(s.read_output()), bits(ref0))          # read waits for the frames in flight
        st = _lib.CrtFrameStats()
        assert hip.crt_frame_time_stats(C.byref(st), 1) == 0
        assert st.frames == 7 and st.sumMs[2] > 0 and 0 < st.extentMs < 50.0   # every frame is in the sums; the extent (first start -> last end, incl. idle gaps between these 0.09 ms frames) is finite
        # every slot's buffer holds the same frame: 8 more frames end on the other slot parity
        s.render_raw(ASYNC)
        assert np.array_equal(bits(s.read_output()), bits(ref0))
        # scene edit between ASYNC frames: the upload waits for queued frames, the next frame sees it
        m = sc.instances[1].matrix.copy(); m[3, :3] += np.array([0.8, 0.4, -0.3], np.float32)
        p, keep = _lib.fptr(m)
        s.render_raw(ASYNC); s.render_raw(ASYNC)
        s.h.crth_set_mesh_matrix(1, p)
        s.render(postprocess=False)                                       # mirrored Renderer: uploads the dirty range, then a synchronous frame
        moved = s.output().copy()
        s.render_raw(ASYNC); s.render_raw(ASYNC); s.render_raw(ASYNC)
        orc2 = oracle_lib.Oracle(s.arenas(), nthreads=nthreads)
        iv, ip, pos = s.camera()
        ref1 = orc2.trace(orc2.raygen(s.width, s.height, iv, ip), pos, sc.sun_angle)[0]
        assert not np.array_equal(bits(ref0), bits(ref1))
        assert np.array_equal(bits(moved), bits(ref1)) and np.array_equal(bits(s.read_output()), bits(ref1))
        # instrumented and synchronous frames still work in between, with exact counters
        s.render_raw(ASYNC); s.render_raw(8)
        _, stc = orc2.trace(orc2.raygen(s.width, s.height, iv, ip), pos, sc.sun_angle)
        assert s.counters() == stc
        assert np.array_equal(bits(s.read_output()), bits(ref1))


def test_short_bursts_are_never_held_back(monkeypatch):
    """The start-up stagger (crt_state.h State::burstFrames, crt_frame.h crt1_render) holds back the first frame of slots 1.. when a burst of frames in
    flight starts on an idle device -- for a caller that STREAMS. A caller that submits two or three ASYNC frames and then reads
    would only pay it as latency: such bursts must never be held back. crt_debug_staggered_frames counts the delay
    launches, so the rule is checked exactly; the burst latency with the stagger on and forced off is printed and loosely bounded."""
    import time
    monkeypatch.delenv("CRT_STAGGER_US", raising=False)
    monkeypatch.setenv("CRT_FRAMES_IN_FLIGHT", "3")
    sc = scenes.get("cornell-1k")
    ASYNC = 4
    hip = _lib.hip()

    def staggered():
        v = C.c_uint64()
        assert hip.crt_debug_staggered_frames(C.byref(v)) == 0
        return int(v.value)

    def short_bursts(s, n=40):
        lat = []
        for _ in range(n):
            t0 = time.perf_counter()
            for _ in range(3):
                s.render_raw(ASYNC)
            s.sync()
            lat.append(time.perf_counter() - t0)
        return float(np.median(lat)) * 1e3

    with driver.Session(1920, 1080, device=0) as s:
        s.load_scene(sc)
        s.render_raw(0); ref = s.read_output().copy()
        on = short_bursts(s)
        assert staggered() == 0                                        # nothing but 3-frame bursts so far: never held back
        for _ in range(12):
            s.render_raw(ASYNC)                                        # a streaming caller: a burst longer than the slot count ...
        s.sync()
        assert staggered() == 0                                        # ... (itself started like any first burst) ...
        for _ in range(12):
            s.render_raw(ASYNC)
        assert np.array_equal(bits(s.read_output()), bits(ref))
        assert staggered() == 2                                        # ... makes the NEXT burst start staggered: slots 1 and 2 once
        short_bursts(s, 3)                                             # the burst before was long: this one is staggered, the following are not
        assert staggered() == 4
    monkeypatch.setenv("CRT_STAGGER_US", "0")
    with driver.Session(1920, 1080, device=0) as s:
        s.load_scene(sc)
        s.render_raw(0)
        off = short_bursts(s)
        for _ in range(12):
            s.render_raw(ASYNC)
        s.sync()
        for _ in range(12):
            s.render_raw(ASYNC)
        s.sync()
        assert staggered() == 0
    print(f"3-frame burst + sync, cornell-1k 1920x1080: {on:.3f} ms with the automatic stagger, {off:.3f} ms with CRT_STAGGER_US=0")
    assert on <= 1.25 * off + 0.05


def test_empty_inputs(nthreads):
    """No instances: every ray misses and samples the skybox (kernel_main.cl:198 loops zero times); zero query rays; a
    zero-byte upload; a frame of the minimum size."""
    sc = scenes.get("tiny")
    hip = _lib.hip()
    with driver.Session(16, 16, device=0) as s:                       # the smallest frame the reference accepts (Renderer.cpp:200)
        s.load_scene(sc)
        a = dict(s.arenas())
        a["instances"] = np.concatenate([a["instances"], np.zeros(5, _lib.INSTANCE_DTYPE)])   # what the device pool holds past the uploads
        orc = oracle_lib.Oracle(a, nthreads=nthreads)
        iv, ip, pos = s.camera()
        args, _, _ = s.trace_args()
        fp = C.POINTER(C.c_float)
        for n_inst in (0, 1, len(sc.instances), len(sc.instances) + 5):    # + 5: instances that were never uploaded (all-zero records)
            args.numMeshes = n_inst
            orc.s.numInstances = n_inst
            ref, st = orc.trace(orc.raygen(16, 16, iv, ip), pos, sc.sun_angle)
            for flags in (8, 4, 8 | 32):
                if flags & 32:
                    ref, st = orc.trace(orc.raygen(16, 16, iv, ip), pos, sc.sun_angle, shadows=True)
                assert hip.crt_render(C.byref(args), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp), flags) == 0
                assert np.array_equal(bits(s.read_output()), bits(ref)), (n_inst, flags)
                if flags & 8:
                    assert s.counters() == st
            if n_inst == 0:
                assert st["hits"] == 0 and st["misses"] == 256 and st["traversals"] == 0
        assert hip.crt_query_hits(None, None, 0, 0, None) == 0
        assert hip.crt_upload_triangles(None, 0, 0) == 0 and hip.crt_upload_instances(None, 0, 0) == 0


def test_readback_frames_arrive_in_pinned_host_memory(nthreads):
    """CRT_RENDER_READBACK: the frame (float4, or RGBA8 with UNORM8) is copied to pinned host memory behind its own
    kernels; crt_map_host_frame returns the copy of the most recent such frame, for synchronous frames and frames in flight."""
    sc = scenes.get("tiny")
    READBACK, ASYNC, UNORM8 = 128, 4, 64
    hip = _lib.hip()
    with driver.Session(200, 120, device=0) as s:
        s.load_scene(sc)
        orc = oracle_lib.Oracle(s.arenas(), nthreads=nthreads)
        iv, ip, pos = s.camera()
        ref, _ = orc.trace(orc.raygen(200, 120, iv, ip), pos, sc.sun_angle)
        ptr, n = C.c_void_p(), C.c_size_t()
        assert hip.crt_map_host_frame(C.byref(ptr), C.byref(n)) == -2          # nothing read back yet
        for flags in (READBACK, READBACK | ASYNC, READBACK | ASYNC, READBACK | ASYNC, READBACK | ASYNC):
            s.render_raw(flags)
            assert hip.crt_map_host_frame(C.byref(ptr), C.byref(n)) == 0 and n.value == 200 * 120 * 16
            got = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(120, 200, 4))
            assert np.array_equal(bits(got), bits(ref)), flags
        s.render_raw(READBACK | ASYNC | UNORM8)
        assert hip.crt_map_host_frame(C.byref(ptr), C.byref(n)) == 0 and n.value == 200 * 120 * 4
        got8 = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(120, 200, 4))
        assert np.array_equal(got8, orc.pack_unorm8(ref))
        s.resize(96, 64)
        assert hip.crt_map_host_frame(C.byref(ptr), C.byref(n)) == -2          # the old frame's geometry is gone


def test_animated_instances_stay_pipelined_and_ordered(nthreads):
    """Instance uploads are host-side and versioned: three frames with three different instance tables are submitted
    back to back (no waiting in between), each must show exactly the table it was submitted with."""
    sc = scenes.get("tiny")
    READBACK, ASYNC = 128, 4
    hip = _lib.hip()
    W, H = 200, 120
    with driver.Session(W, H, device=0) as s:
        s.load_scene(sc)
        base = s.arenas()
        iv, ip, pos = s.camera()
        args, _, _ = s.trace_args()
        fp = C.POINTER(C.c_float)
        for _ in range(4):                                   # warm the slots
            assert hip.crt_render(C.byref(args), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp), ASYNC) == 0
        tables = []
        for k in range(3):
            inst = base["instances"].copy()
            inst["inv"][1][3, :3] += np.float32(0.4 * (k + 1))          # inverse transform: translation row
            inst["inv"][0][3, 1] -= np.float32(0.3 * k)
            tables.append(inst)
            assert hip.crt_upload_instances(inst.ctypes.data, 0, len(inst)) == 0
            assert hip.crt_render(C.byref(args), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp), ASYNC | READBACK) == 0
        ptr, n = C.c_void_p(), C.c_size_t()
        frames = []
        for back in (2, 1, 0):
            assert hip.crt_map_host_frame_back(back, C.byref(ptr), C.byref(n)) == 0 and n.value == W * H * 16
            frames.append(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(H, W, 4)).copy())
        assert hip.crt_map_host_frame_back(3, C.byref(ptr), C.byref(n)) == -2
        for k in range(3):
            a = dict(base); a["instances"] = tables[k]
            orc = oracle_lib.Oracle(a, nthreads=nthreads)
            ref, _ = orc.trace(orc.raygen(W, H, iv, ip), pos, sc.sun_angle)
            assert np.array_equal(bits(frames[k]), bits(ref)), k
        assert not np.array_equal(bits(frames[0]), bits(frames[1])) and not np.array_equal(bits(frames[1]), bits(frames[2]))
