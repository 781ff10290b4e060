"""The measurement aids added in round 3 must not change what is rendered, and must report sane values: the three-frames-in-one
dispatch diagnostic (CRT_RENDER_DIAG_MIX3), the in-flight clock probe, the per-frame burst times, the start-up stagger of frames in
flight (a one-wave timer kernel in front of the first frame a slot runs after the device was idle)."""
import ctypes as C

import numpy as np
import pytest

from clraytracer_amd import _lib, driver, scenes
from util import bits

pytestmark = pytest.mark.gpu
COUNT, ASYNC, MIX3, SHADOWS, POST, UNORM8 = 8, 4, 1024, 32, 1, 64


def test_mix3_dispatch_renders_the_same_frame_three_times():
    sc = scenes.get("tiny")
    with driver.Session(328, 200, device=0) as s:
        s.load_scene(sc)
        s.render_raw(COUNT); ref = s.read_output(); c1 = s.counters()
        s.render_raw(MIX3); got = s.read_output()
        assert np.array_equal(bits(got), bits(ref))
        s.render_raw(MIX3 | COUNT); c3 = s.counters()
        assert np.array_equal(bits(s.read_output()), bits(ref))
        assert all(c3[k] == 3 * c1[k] for k in ("rays", "primary", "secondary", "hits", "misses", "innerVisits", "triTests", "traversals", "pops")) and c3["maxStack"] == c1["maxStack"]
        # with the per-pixel epilogue and the shadow-ray instantiation too
        for f in (POST | UNORM8, SHADOWS):
            s.render_raw(f); want = s.read_output()
            s.render_raw(f | MIX3)
            assert np.array_equal(bits(s.read_output()), bits(want)), f
        a, iv, ip = s.trace_args()
        fp = C.POINTER(C.c_float)
        assert s.hip.crt_render(C.byref(a), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp), MIX3 | ASYNC) != 0      # synchronous only
    with driver.Session(328, 200, devices=[0, 0]) as s:
        s.load_scene(sc)
        a, iv, ip = s.trace_args()
        assert s.hip.crt_render(C.byref(a), iv.ctypes.data_as(C.POINTER(C.c_float)), ip.ctypes.data_as(C.POINTER(C.c_float)), MIX3) != 0   # one device only


def test_clock_probe_and_burst_times():
    sc = scenes.get("tiny")
    hip = _lib.hip()
    with driver.Session(640, 360, device=0) as s:
        s.load_scene(sc)
        for _ in range(12):
            s.render_raw(ASYNC)
        ghz = C.c_double(0.0)
        assert hip.crt_debug_measure_clock(200, C.byref(ghz)) == 0 and 0.5 < ghz.value < 2.6, ghz.value
        assert hip.crt_debug_measure_clock(0, C.byref(ghz)) != 0 and hip.crt_debug_measure_clock(200, None) != 0
        s.sync()
        assert hip.crt_frame_time_stats(None, 1) == 0
        for _ in range(9):
            s.render_raw(ASYNC)
        s.sync()
        n = C.c_size_t(0)
        t = np.zeros((256, 2), np.float64)
        assert hip.crt_debug_read_frame_times(t.ctypes.data, 256, C.byref(n)) == 0 and n.value == 9
        t = t[:9]
        assert (t[:, 1] > t[:, 0]).all() and (t[:, 0] >= 0).all() and t[:, 1].max() < 1000.0
        st = _lib.CrtFrameStats()
        assert hip.crt_frame_time_stats(C.byref(st), 0) == 0 and st.frames == 9
        assert abs(st.extentMs - t[:, 1].max()) < 1e-6 and 0 < st.firstFrameMs <= st.extentMs


@pytest.mark.parametrize("stagger", ["0", "150", None])
def test_start_up_stagger_changes_no_pixel(stagger, monkeypatch):
    """Bursts of frames in flight after an idle device, with the timer kernel in front of slots 1 and 2 (automatic, forced, off): the
    frames are the synchronous frame's, whichever slot ran them."""
    if stagger is None:
        monkeypatch.delenv("CRT_STAGGER_US", raising=False)
    else:
        monkeypatch.setenv("CRT_STAGGER_US", stagger)
    sc = scenes.get("tiny")
    with driver.Session(328, 200, device=0) as s:
        s.load_scene(sc)
        s.render_raw(0); ref = s.read_output()
        for burst in (1, 2, 3, 7):
            for _ in range(burst):
                s.render_raw(ASYNC)
            assert np.array_equal(bits(s.read_output()), bits(ref)), burst      # read_output waits: the next burst starts from idle again
