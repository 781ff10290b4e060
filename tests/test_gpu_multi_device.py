"""Several GPUs in ONE process behind the unchanged Renderer / C-ABI (crt_init_devices, Renderer::InitializeDevices):
replicated scene, 16-row bands dealt round-robin to the devices, every device's bands gathered into the first device's
frame by peer copies, Render() returning the whole frame (upstream drives one device, Renderer.cpp:134).

The GPU box has one MI355X, so the path is rehearsed by listing device 0 several times: N independent device states
(own pools, streams, frame slots, worker threads) that happen to share a GPU -- every line of the multi-device code runs,
only the peer copies are local. Where two or more GPUs are visible the same checks run on distinct devices."""
import ctypes as C

import numpy as np
import pytest

from clraytracer_amd import _lib, driver, scenes
import oracle_lib
from util import bits

pytestmark = pytest.mark.gpu


def visible_gpus():
    import torch
    return torch.cuda.device_count()


def single_frame(sc, w, h, flags=8):
    with driver.Session(w, h, device=0) as s:
        s.load_scene(sc)
        s.render_raw(flags)
        return s.read_output(), s.counters()


@pytest.mark.parametrize("ndev", [2, 3, 8])
def test_stitched_frame_through_renderer_equals_single_device(ndev):
    """Renderer::Render() + MapOutput() with N device states == the single-device frame, bit for bit; counters add up."""
    sc = scenes.get("tiny")
    w, h = 328, 200                                    # 25 tile rows: uneven band counts per device, partial last band
    ref, ref_cnt = single_frame(sc, w, h)
    with driver.Session(w, h, devices=[0] * ndev) as s:
        assert s.hip.crt_num_devices() == ndev
        s.load_scene(sc)
        s.render(postprocess=False)                    # the mirrored Renderer::Render (synchronous) ...
        got = s.output()                               # ... and MapOutput: the whole frame, not one device's bands
        assert np.array_equal(bits(got), bits(ref))
        s.render_raw(8)
        assert np.array_equal(bits(s.read_output()), bits(ref)) and s.counters() == ref_cnt
        # frames in flight: slots rotate on every device in step; the last frame read is complete
        for _ in range(7):
            s.render_raw(4)
        assert np.array_equal(bits(s.read_output()), bits(ref))
        # PostProcess and the RGBA8 target are per-pixel: every device processes its bands before the gather
        s.render(postprocess=True)
        post = s.output()
    with driver.Session(w, h, device=0) as s1:
        s1.load_scene(sc)
        s1.render(postprocess=True)
        assert np.array_equal(bits(s1.output()), bits(post))


def test_multi_device_matches_oracle_and_follows_scene_edits(nthreads):
    """cornell-1k at 960x540 on 4 device states against the oracle; then a resize, a moved instance and a camera change:
    every edit reaches every device."""
    sc = scenes.get("tiny")
    with driver.Session(960, 540, devices=[0, 0, 0, 0]) as s:
        s.load_scene(sc)
        orc = oracle_lib.Oracle(s.arenas(), nthreads=nthreads)

        def check():
            s.render_raw(8)
            iv, ip, pos = s.camera()
            a = s.arenas()
            o = oracle_lib.Oracle(a, nthreads=nthreads)
            ref, st = o.trace(o.raygen(s.width, s.height, iv, ip), pos, sc.sun_angle)
            got = s.read_output()
            bad = np.nonzero((bits(got) != bits(ref)).any(axis=(1, 2)))[0]
            assert len(bad) == 0, f"{len(bad)} rows differ: {bad[:24]}"
            assert s.counters() == st
        check()
        s.resize(640, 360)
        check()
        m = np.eye(4, dtype=np.float32); m[3, :3] = (1.5, 4.0, -3.0)
        p, keep = _lib.fptr(m)
        s.h.crth_set_mesh_matrix(1, p)
        s.render(postprocess=False)                    # Renderer::Render uploads the dirty instance range (Renderer.cpp:312-320) to every device
        check()
        s.set_camera((2.0, 9.0, 14.0), scenes._normalize((-0.1, -0.45, -1.0)))
        check()
        # shadow-ray extension and the device BVH builder run on every device too
        s.render_raw(8 | 32)
        iv, ip, pos = s.camera()
        o = oracle_lib.Oracle(s.arenas(), nthreads=nthreads)
        ref, st = o.trace(o.raygen(s.width, s.height, iv, ip), pos, sc.sun_angle, shadows=True)
        assert np.array_equal(bits(s.read_output()), bits(ref)) and s.counters() == st
    with driver.Session(320, 200, devices=[0, 0]) as s:
        s.load_scene(sc, device_bvh_build=True)
        s.render_raw(0)
        got = s.read_output()
    ref, _ = single_frame(sc, 320, 200)
    assert np.array_equal(bits(got), bits(ref))


def test_pipelined_readback_delivers_whole_frames():
    """CRT_RENDER_READBACK on a multi-device session: the pinned host copy holds the WHOLE gathered frame, for each of the
    frames in flight, while later frames are already running."""
    sc = scenes.get("tiny")
    w, h = 256, 144
    ref, _ = single_frame(sc, w, h, flags=0)
    hip = _lib.hip()
    with driver.Session(w, h, devices=[0, 0, 0]) as s:
        s.load_scene(sc)
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        for k in range(6):
            s.render_raw(4 | 128)
            assert hip.crt_map_host_frame(C.byref(ptr), C.byref(nbytes)) == 0 and nbytes.value == w * h * 16
            host = np.frombuffer((C.c_char * nbytes.value).from_address(ptr.value), np.float32).reshape(h, w, 4)
            assert np.array_equal(bits(host), bits(ref)), k
        s.sync()
        # the bands belong to the session; single-device diagnostics are refused
        assert hip.crt_set_row_bands(16, 0, 2) == -5 or hip.crt_set_row_bands(16, 0, 2) != 0
        a, iv, ip = s.trace_args()
        fp = C.POINTER(C.c_float)
        assert hip.crt_render(C.byref(a), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp), 2) != 0     # WRITE_RAYS


@pytest.mark.skipif(visible_gpus() < 2, reason="needs two visible GPUs (the rehearsal above covers the code path on one)")
def test_two_real_devices():
    sc = scenes.get("cornell-1k")
    ref, ref_cnt = single_frame(sc, 1920, 1080)
    with driver.Session(1920, 1080, devices=[0, 1]) as s:
        s.load_scene(sc)
        s.render(postprocess=False)
        assert np.array_equal(bits(s.output()), bits(ref))
        s.render_raw(8)
        assert s.counters() == ref_cnt
