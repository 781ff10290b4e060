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


def _render(s, flags):
    a, iv, ip = s.trace_args()
    fp = C.POINTER(C.c_float)
    return s.hip.crt_render(C.byref(a), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp), flags)


def test_peer_access_state_is_reported():
    """crt_peer_access / crt_gather_path: how each device's bands reach the primary (a rehearsal session: the same GPU)."""
    sc = scenes.get("tiny")
    hip = _lib.hip()
    with driver.Session(256, 144, devices=[0, 0, 0]) as s:
        s.load_scene(sc)
        assert [hip.crt_peer_access(d) for d in range(3)] == [2, 2, 2]
        assert hip.crt_peer_access(3) < 0 and b"same-device" in hip.crt_gather_path()
    with driver.Session(256, 144, device=0) as s:
        assert hip.crt_peer_access(0) == 2 and b"one device" in hip.crt_gather_path()


@pytest.mark.parametrize("flags", [0, 4])
def test_a_failing_secondary_fails_the_frame_and_the_session_recovers(flags, monkeypatch):
    """If a secondary device's submission fails, crt_render returns the error WITHOUT queueing the primary's wait for that
    device's bands (it would otherwise wait on the slot's previous frame's event and present stale bands as a finished
    frame). The slot rotation advances on every device alike, so the following frames are whole and correct again."""
    sc = scenes.get("tiny")
    w, h = 328, 200
    ref, _ = single_frame(sc, w, h, flags=0)
    hip = _lib.hip()
    with driver.Session(w, h, devices=[0, 0, 0]) as s:
        s.load_scene(sc)
        for _ in range(4):
            assert _render(s, flags) == 0
        monkeypatch.delenv("CRT_DEBUG_HOOKS", raising=False)
        assert hip.crt_debug_inject_failure(1) == -5          # CRT_E_UNSUPPORTED: the hook is dead in a process that did not ask for it
        assert _render(s, flags) == 0
        monkeypatch.setenv("CRT_DEBUG_HOOKS", "1")
        for bad in (1, 2, 0):
            assert hip.crt_debug_inject_failure(bad) == 0
            assert _render(s, flags) != 0                      # the frame is abandoned, the error reported
            for _ in range(5):                                 # ... and every slot is usable again
                assert _render(s, flags) == 0
                assert np.array_equal(bits(s.read_output()), bits(ref))
        assert hip.crt_debug_inject_failure(7) != 0


def test_failed_resize_rolls_every_device_back(monkeypatch):
    """crt_resize on a session is all-or-nothing: when one device cannot reallocate, the devices that already did go back to
    the old size, the error is returned and the session keeps rendering the old frame size."""
    sc = scenes.get("tiny")
    w, h = 328, 200
    ref, _ = single_frame(sc, w, h, flags=0)
    hip = _lib.hip()
    with driver.Session(w, h, devices=[0, 0, 0]) as s:
        s.load_scene(sc)
        monkeypatch.setenv("CRT_DEBUG_HOOKS", "1")
        assert hip.crt_debug_inject_failure(2) == 0
        with pytest.raises(driver.CrtError):
            s.resize(640, 360)                                 # Renderer::OnWindowResize -> crt_resize fails on device 2
        s.h.crth_clear_error()
        assert _render(s, 0) == 0                              # still w x h on every device (and the old projection)
        assert np.array_equal(bits(s.read_output()), bits(ref))
        s.resize(640, 360)                                     # a later resize goes through
        assert _render(s, 0) == 0
        got = s.read_output()
    ref2, _ = single_frame(sc, 640, 360, flags=0)
    assert np.array_equal(bits(got), bits(ref2))


def test_devices_without_rows_stay_in_step():
    """A frame so short that some devices own no rows of it (16 rows = one band) must not let those devices fall out of the
    slot rotation: after pipelined short frames and a resize to a taller frame, every band of every later frame is there."""
    sc = scenes.get("tiny")
    with driver.Session(64, 16, devices=[0, 0, 0]) as s:
        s.load_scene(sc)
        for _ in range(4):                                     # devices 1 and 2 own nothing here
            assert _render(s, 4) == 0
        short = s.read_output()
        s.resize(328, 200)
        s.set_camera(sc.camera_pos, sc.camera_front)
        frames = []
        for _ in range(5):                                     # not a multiple of the slot count
            assert _render(s, 4) == 0
            frames.append(s.read_output())
    ref_short, _ = single_frame(sc, 64, 16, flags=0)
    ref, _ = single_frame(sc, 328, 200, flags=0)
    assert np.array_equal(bits(short), bits(ref_short))
    for k, f in enumerate(frames):
        assert np.array_equal(bits(f), bits(ref)), k


def test_synchronous_frame_overlaps_the_devices_shares():
    """Renderer::Render() semantics on N devices: the secondaries must not be waited for on the host before the primary's
    share is submitted (that made a synchronous frame last T_secondary + T_primary). Rehearsed on one GPU the shares
    compete for the same device, so the check is lenient: a 2-state synchronous frame of a tail-bound scene must take
    clearly less than two single-device frames (each half-frame keeps the full frame's longest rays)."""
    import time
    sc = scenes.get("multi-1M")
    w, h = 1920, 1080

    def sync_ms(devices):
        with driver.Session(w, h, devices=devices) if devices else driver.Session(w, h, device=0) as s:
            s.load_scene(sc)
            for _ in range(10):
                assert _render(s, 0) == 0
            t0 = time.perf_counter()
            for _ in range(40):
                assert _render(s, 0) == 0
            return (time.perf_counter() - t0) / 40 * 1e3
    one = sync_ms(None)
    two = sync_ms([0, 0])
    print(f"synchronous frame: one device state {one:.3f} ms, two states on the same GPU {two:.3f} ms")
    assert two < 1.6 * one, (one, two)


@pytest.mark.parametrize("variant", ["wavefront", "refill", "block", "ldstop"])
def test_compaction_kernels_behind_several_device_states(monkeypatch, variant):
    """The opt-in kernel structures (round 5: the wavefront form keeps frames in flight with a per-slot queue; refill / block count blocks
    where the default kernel counts tiles) also run one share of a banded frame each: three device states, uneven bands, synchronous
    and with frames in flight, against the single-device default kernel."""
    sc = scenes.get("tiny")
    w, h = 328, 200
    monkeypatch.delenv("CRT_KERNEL", raising=False)
    ref, ref_cnt = single_frame(sc, w, h)
    monkeypatch.setenv("CRT_KERNEL", variant)
    with driver.Session(w, h, devices=[0, 0, 0]) as s:
        s.load_scene(sc)
        prefix = {"wavefront": "crt_primary_kernel<", "refill": "crt_trace_refill_kernel<", "block": "crt_trace_block_kernel<", "ldstop": "crt_trace_ldstop_kernel<"}[variant]
        s.render_raw(8)
        assert s.last_kernel().startswith(prefix), s.last_kernel()
        assert np.array_equal(bits(s.read_output()), bits(ref)) and s.counters() == ref_cnt
        for _ in range(7):
            s.render_raw(4)
        assert s.last_kernel().startswith(prefix), s.last_kernel()
        assert np.array_equal(bits(s.read_output()), bits(ref))
        s.render_raw(0)
        assert np.array_equal(bits(s.read_output()), bits(ref))


@pytest.mark.parametrize("ndev", [2, 8])
def test_rgba8_frames_are_gathered_as_bytes(ndev, monkeypatch):
    """a CRT_RENDER_UNORM8 frame (upstream's render target is RGBA8, Renderer.cpp:63,192) travels to the first device as the
    4 bytes per pixel every device's Trace epilogue stores, not as float4 bands: the gathered byte frame, the float frame rebuilt from it on
    demand (x = byte / 255), the read-back and frames in flight all equal the single-device RGBA8 frame bit for bit; plain and FXAA frames
    keep the float gather; crt_debug_last_gather reports the bytes; CRT_GATHER_RGBA8=0 gives the old path and the same frames."""
    UNORM8, POST, ASYNC, READBACK, FXAA = 64, 1, 4, 128, 512
    sc = scenes.get("tiny")
    w, h = 328, 200
    monkeypatch.delenv("CRT_GATHER_RGBA8", raising=False)
    ref = {}
    with driver.Session(w, h, device=0) as s:
        s.load_scene(sc)
        assert s.last_gather() == (0, 0)
        for f in (0, UNORM8, UNORM8 | POST, UNORM8 | FXAA):
            s.render_raw(f)
            ref[f] = (s.read_output().copy(), s.read_output_rgba8().copy())
    assert not np.array_equal(ref[UNORM8][1], ref[UNORM8 | POST][1])
    rows_sent = sum(1 for y in range(h) if (y // 16) % ndev != 0)         # rows the secondaries own
    ptr, nbytes = C.c_void_p(), C.c_size_t()
    for env in (None, "0"):
        if env is not None:
            monkeypatch.setenv("CRT_GATHER_RGBA8", env)
        with driver.Session(w, h, devices=[0] * ndev) as s:
            s.load_scene(sc)
            for f in (UNORM8, UNORM8 | POST):
                s.render_raw(f)
                assert s.last_gather() == ((rows_sent * w * 4, 4) if env is None else (rows_sent * w * 16, 16)), (f, s.last_gather())
                assert np.array_equal(s.read_output_rgba8(), ref[f][1]), f             # the gathered bytes ARE the frame
                assert np.array_equal(bits(s.read_output()), bits(ref[f][0])), f       # float frame rebuilt on demand
                assert np.array_equal(s.read_output_rgba8(), ref[f][1]), f             # ... and the bytes once more, after the rebuild
                for _ in range(7):
                    s.render_raw(f | ASYNC)                                            # slots rotate on every device in step
                assert np.array_equal(s.read_output_rgba8(), ref[f][1]), f
                s.render_raw(f | READBACK)
                assert s.hip.crt_map_host_frame(C.byref(ptr), C.byref(nbytes)) == 0 and nbytes.value == w * h * 4
                host = np.frombuffer((C.c_char * nbytes.value).from_address(ptr.value), np.uint8).reshape(h, w, 4)
                assert np.array_equal(host, ref[f][1]), f
            # what follows an RGBA8 frame on the same slot: plain and filtered frames gather float4 bands as before
            for f in (0, UNORM8 | FXAA):
                s.render_raw(f)
                assert s.last_gather() == (rows_sent * w * 16, 16)
                assert np.array_equal(bits(s.read_output()), bits(ref[f][0])), f
            s.render_raw(UNORM8)
            assert np.array_equal(bits(s.read_output()), bits(ref[UNORM8][0]))
            # through the mirrored Renderer: SetUnorm8 + MapOutputRGBA8
            s.h.crth_set_unorm8(1)
            s.render(postprocess=True)
            got = _lib.as_array(s.h.crth_map_output_rgba8(), w * h * 4, np.uint8).reshape(h, w, 4)
            assert np.array_equal(got, ref[UNORM8 | POST][1])
            s.h.crth_set_unorm8(0)


def test_rgba8_gather_behind_the_wavefront_form(monkeypatch):
    """The wavefront form has no epilogue: its devices pack the bytes of the rows they own in a launch of their own (crt_pack_owned_kernel),
    never touching rows whose bytes other devices may already have delivered."""
    UNORM8, POST = 64, 1
    sc = scenes.get("tiny")
    w, h = 328, 200
    monkeypatch.delenv("CRT_KERNEL", raising=False)
    with driver.Session(w, h, device=0) as s:
        s.load_scene(sc)
        s.render_raw(UNORM8 | POST)
        ref = s.read_output_rgba8().copy()
    monkeypatch.setenv("CRT_KERNEL", "wavefront")
    with driver.Session(w, h, devices=[0, 0, 0]) as s:
        s.load_scene(sc)
        for _ in range(3):
            s.render_raw(UNORM8 | POST)
            assert s.last_kernel().startswith("crt_primary_kernel<") and s.last_gather()[1] == 4
            assert np.array_equal(s.read_output_rgba8(), ref)
