"""Opt-in kernel structures must be bit-identical to the default megakernel (own module: one device session at a time)."""
import numpy as np
import pytest

from clraytracer_amd import driver, scenes
from util import bits

pytestmark = pytest.mark.gpu
FLAG_COUNT = 8
# what crt_debug_last_kernel must report for a frame of each form (the names rocprofv3 prints, profiles/r0N{wavefront,refill,block}_summary.md)
KERNEL_OF = {"default": "crt_trace_kernel<", "wavefront": "crt_primary_kernel<", "refill": "crt_trace_refill_kernel<", "block": "crt_trace_block_kernel<",
             "ldstop": "crt_trace_ldstop_kernel<"}


@pytest.mark.parametrize("variant", ["wavefront", "refill", "block", "ldstop"])
def test_kernel_variants_are_bit_identical(nthreads, monkeypatch, variant):
    """The opt-in kernel structure (CRT_KERNEL=wavefront: one launch per bounce with ballot compaction in between) renders
    the same bits and counts the same work as the default megakernel. (Round 1's `persistent` and `lds` structures were
    retired in round 2; docs/DESIGN_HISTORY.md 4f keeps their measurements.)"""
    sc = scenes.get("tiny")
    POST, ASYNC, UNORM8, FXAA = 1, 4, 64, 512
    stages = (POST, UNORM8, POST | UNORM8, POST | UNORM8 | ASYNC, FXAA | UNORM8, FXAA | POST | UNORM8)
    monkeypatch.delenv("CRT_KERNEL", raising=False)
    with driver.Session(256, 144, device=0) as s:
        s.load_scene(sc)
        s.render_raw(FLAG_COUNT)
        assert s.last_kernel() == "crt_trace_kernel<1,0,0,0,0>"
        ref = s.read_output(); ref_cnt = s.counters()
        # the default kernel applies the RGBA8 target / PostProcess in its epilogue ...
        fused = []
        for f in stages:
            s.render_raw(f); fused.append(s.read_output().copy())
    monkeypatch.setenv("CRT_KERNEL", variant)
    with driver.Session(256, 144, device=0) as s:
        s.load_scene(sc)
        assert s.last_kernel() == ""                                       # nothing rendered yet
        s.render_raw(FLAG_COUNT)
        assert s.last_kernel().startswith(KERNEL_OF[variant]), s.last_kernel()   # not vacuous: the form really rendered the frame
        got = s.read_output(); cnt = s.counters()
        s.render_raw(0)
        got2 = s.read_output()
        # ... the variant runs them as launches of their own (crt_quantize_kernel, crt_postprocess_kernel): same bits
        for f, want in zip(stages, fused):
            s.render_raw(f)
            assert s.last_kernel().startswith(KERNEL_OF[variant]), (f, s.last_kernel())
            assert np.array_equal(bits(s.read_output()), bits(want)), f
    assert np.array_equal(bits(got), bits(ref)) and np.array_equal(bits(got2), bits(ref))
    assert cnt == ref_cnt
    assert not np.array_equal(bits(fused[0]), bits(ref)) and not np.array_equal(bits(fused[2]), bits(fused[0]))


@pytest.mark.parametrize("variant", ["wavefront", "refill", "block", "ldstop"])
def test_wavefront_compaction_at_config4_size_equals_the_oracle(nthreads, monkeypatch, variant):
    """BASELINE config 4 as written ("LDS stack + wavefront compaction on"): multi-1M, 1920x1080, rendered by the wavefront form
    (crt_primary_kernel -> ballot compaction -> crt_bounce_kernel; the bounce loop of kernel_main.cl:187 split into launches)
    and compared with the ORACLE -- frame bits and every work counter -- and with the oracle-written known answer
    tests/golden/full_frames.json; synchronous, with frames in flight, and with the shadow-ray-free flags the variant supports."""
    import hashlib, json, os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full_frames.json")))["multi-1M"]
    import oracle_lib
    sc = scenes.get("multi-1M")
    monkeypatch.setenv("CRT_KERNEL", variant)
    with driver.Session(1920, 1080, device=0) as s:
        s.load_scene(sc)
        orc = oracle_lib.Oracle(s.arenas(), nthreads=nthreads)
        iv, ip, pos = s.camera()
        ref, st = orc.trace(orc.raygen(1920, 1080, iv, ip), pos, sc.sun_angle)
        s.render_raw(FLAG_COUNT)
        assert s.last_kernel().startswith(KERNEL_OF[variant]), s.last_kernel()
        got = s.read_output(); cnt = s.counters()
        assert cnt == st                                                   # rays, hits, pops, inner visits, triangle tests, max stack ...
        assert np.array_equal(bits(got[..., :3]), bits(ref[..., :3]))
        assert hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest() == gold["frame_sha256"]
        assert {k: cnt[k] for k in gold["counters"]} == gold["counters"]
        s.render_raw(0)
        assert np.array_equal(bits(s.read_output()), bits(got))
        for _ in range(5):
            s.render_raw(4)                                                # frames in flight
        assert s.last_kernel().startswith(KERNEL_OF[variant]), s.last_kernel()
        assert np.array_equal(bits(s.read_output()), bits(got))
        print(f"{variant} multi-1M 1920x1080: {st['rays']} rays, {st['secondary']} compacted bounce rays, frame and counters equal to the oracle")


@pytest.mark.parametrize("variant", ["wavefront", "refill", "block", "ldstop"])
def test_a_form_refuses_what_it_cannot_render(monkeypatch, variant):
    """ONE rule (VERDICT r5 #1b): a frame the selected form cannot render -- shadow rays, refraction, the diagnostic mix, the stamped
    launch (wavefront), more than 64 instances (refill / block), a forced instance tree -- returns CRT_E_UNSUPPORTED (-5) and renders
    NOTHING: no silent fall-back to the default kernel, the last frame and its kernel name stay what they were."""
    import ctypes as C
    from clraytracer_amd import _lib
    SHADOWS, REFRACT, STAMPS, MIX3, POST = 32, 256, 16, 1024, 1
    sc = scenes.get("tiny")
    monkeypatch.setenv("CRT_KERNEL", variant)
    with driver.Session(256, 144, device=0) as s:
        s.load_scene(sc)
        s.render_raw(0)
        name = s.last_kernel(); frame = s.read_output().copy()
        assert name.startswith(KERNEL_OF[variant])
        a, iv, ip = s.trace_args()
        fp = C.POINTER(C.c_float)
        refused = [SHADOWS, REFRACT, SHADOWS | REFRACT, SHADOWS | POST, SHADOWS | 4, MIX3] + ([STAMPS] if variant in ("wavefront", "ldstop") else [])
        for f in refused:
            assert s.hip.crt_render(C.byref(a), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp), f) == _lib.CRT_E_UNSUPPORTED, f
            assert s.last_kernel() == name
        assert np.array_equal(bits(s.read_output()), bits(frame))
        if variant in ("refill", "block"):
            s.render_raw(STAMPS)                                            # refill / block have stamped instantiations of their own
            assert s.last_kernel().startswith(KERNEL_OF[variant]) and np.array_equal(bits(s.read_output()), bits(frame))
            a.numMeshes = 65                                                # one 64-bit candidate mask per lane
            assert s.hip.crt_render(C.byref(a), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp), 0) == _lib.CRT_E_UNSUPPORTED
    monkeypatch.setenv("CRT_TLAS", "1")                                      # the instance tree belongs to the default kernel
    with driver.Session(256, 144, device=0) as s:
        s.load_scene(sc)
        a, iv, ip = s.trace_args()
        assert s.hip.crt_render(C.byref(a), iv.ctypes.data_as(C.POINTER(C.c_float)), ip.ctypes.data_as(C.POINTER(C.c_float)), 0) == _lib.CRT_E_UNSUPPORTED


def test_ldstop_tree_top_table_on_one_and_on_many_meshes(nthreads, monkeypatch):
    """CRT_KERNEL=ldstop (north_star's "hot BVH tiles staged in LDS", crt_ldstop.h): the 252-record table is split over the scene's meshes --
    one mesh gets 7-8 levels of its tree (cornell-1k), two meshes 126 records each (tiny), and with more instances than one candidate mask holds
    (130, the chunked linear loop) the form still renders the default kernel's frame and counts the default kernel's work."""
    for name, extra in (("cornell-1k", 0), ("tiny", 0), ("tiny", 128)):
        sc = scenes.get(name)
        frames = {}
        for form in ("default", "ldstop"):
            monkeypatch.setenv("CRT_KERNEL", form)
            with driver.Session(320, 200, device=0) as s:
                s.load_scene(sc)
                if extra:
                    s.h.crth_begin_instances()
                    for k in range(len(sc.instances) + extra):
                        m = scenes._trs(0.5 + 0.1 * (k % 4), (0.2, 1.0, 0.3), 0.41 * k, (float((k % 13) - 6) * 5.0, float((k // 13) - 5) * 5.0, -float(k % 5) * 3.0))
                        pm, keep = _lib_fptr(m)
                        s.h.crth_register_instance(k % 2, 0xFFFF, pm)
                    s.h.crth_end_instances()
                    s.set_camera((0.0, 0.0, 60.0), scenes._normalize((0.0, 0.0, -1.0)))
                s.render_raw(FLAG_COUNT)
                assert s.last_kernel().startswith(KERNEL_OF[form])
                frames[form] = (s.read_output().copy(), s.counters())
                for _ in range(4):
                    s.render_raw(4)
                assert np.array_equal(bits(s.read_output()), bits(frames[form][0]))
        assert np.array_equal(bits(frames["ldstop"][0]), bits(frames["default"][0])), (name, extra)
        assert frames["ldstop"][1] == frames["default"][1] and frames["default"][1]["hits"] > 0, (name, extra)


def _lib_fptr(m):
    from clraytracer_amd import _lib
    return _lib.fptr(m)


def test_unknown_kernel_name_fails_init(monkeypatch):
    """CRT_KERNEL is read by crt_init; a value that names no form is a typo, not a wish for the default kernel (before round 6 it
    silently selected the default, so a variant test could pass without its variant)."""
    monkeypatch.setenv("CRT_KERNEL", "wavefrnt")
    with pytest.raises(driver.CrtError):
        driver.Session(64, 64, device=0)
    for ok in ("", "default"):
        monkeypatch.setenv("CRT_KERNEL", ok)
        with driver.Session(64, 64, device=0) as s:
            s.load_scene(scenes.get("tiny"))
            s.render_raw(32)
            assert s.last_kernel() == "crt_trace_kernel<0,0,1,0,0>"


def test_stamped_launch_renders_the_same_frame():
    """CRT_RENDER_STAMPS (the diagnostic instantiation behind tools/wave_timeline.py: per-wave start/end stamps, 6 waves per
    SIMD) must render the frame the plain launch renders -- also through the per-pixel epilogue (PostProcess, RGBA8 target),
    which the stamped instantiation applies itself -- and hand back one sane stamp record per wave."""
    import ctypes as C
    sc = scenes.get("tiny")
    STAMPS, POST, UNORM8 = 16, 1, 64
    with driver.Session(256, 144, device=0) as s:
        s.load_scene(sc)
        for extra in (0, POST, POST | UNORM8):
            s.render_raw(extra); want = s.read_output().copy()
            s.render_raw(STAMPS | extra)
            assert s.last_kernel() == "crt_trace_kernel<0,1,0,0,0>"
            assert np.array_equal(bits(s.read_output()), bits(want)), extra
        n = C.c_size_t(0)
        assert s.hip.crt_debug_read_stamps(None, 0, C.byref(n)) == 0 and n.value >= (256 // 8) * (144 // 8)
        st = np.zeros((n.value, 8), np.uint64)
        assert s.hip.crt_debug_read_stamps(st.ctypes.data, n.value, C.byref(n)) == 0
        ran = st[st[:, 1] > 0]                                # waves of the list padding never start
        assert len(ran) >= (256 // 8) * (144 // 8) and (ran[:, 1] >= ran[:, 0]).all() and (ran[:, 2] > 0).all()
