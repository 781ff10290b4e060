"""Opt-in kernel structures must be bit-identical to the default megakernel (own module: one device session at a time)."""
import numpy as np
import pytest

from clraytracer_amd import driver, scenes
from util import bits

pytestmark = pytest.mark.gpu
FLAG_COUNT = 8


@pytest.mark.parametrize("variant", ["wavefront"])
def test_kernel_variants_are_bit_identical(nthreads, monkeypatch, variant):
    """The opt-in kernel structure (CRT_KERNEL=wavefront: one launch per bounce with ballot compaction in between) renders
    the same bits and counts the same work as the default megakernel. (Round 1's `persistent` and `lds` structures were
    retired in round 2; DESIGN.md keeps their measurements.)"""
    sc = scenes.get("tiny")
    POST, ASYNC, UNORM8, FXAA = 1, 4, 64, 512
    stages = (POST, UNORM8, POST | UNORM8, POST | UNORM8 | ASYNC, FXAA | UNORM8, FXAA | POST | UNORM8)
    monkeypatch.delenv("CRT_KERNEL", raising=False)
    with driver.Session(256, 144, device=0) as s:
        s.load_scene(sc)
        s.render_raw(FLAG_COUNT)
        ref = s.read_output(); ref_cnt = s.counters()
        # the default kernel applies the RGBA8 target / PostProcess in its epilogue ...
        fused = []
        for f in stages:
            s.render_raw(f); fused.append(s.read_output().copy())
    monkeypatch.setenv("CRT_KERNEL", variant)
    with driver.Session(256, 144, device=0) as s:
        s.load_scene(sc)
        s.render_raw(FLAG_COUNT)
        got = s.read_output(); cnt = s.counters()
        s.render_raw(0)
        got2 = s.read_output()
        # ... the variant runs them as launches of their own (crt_quantize_kernel, crt_postprocess_kernel): same bits
        for f, want in zip(stages, fused):
            s.render_raw(f)
            assert np.array_equal(bits(s.read_output()), bits(want)), f
    assert np.array_equal(bits(got), bits(ref)) and np.array_equal(bits(got2), bits(ref))
    assert cnt == ref_cnt
    assert not np.array_equal(bits(fused[0]), bits(ref)) and not np.array_equal(bits(fused[2]), bits(fused[0]))


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
            assert np.array_equal(bits(s.read_output()), bits(want)), extra
        n = C.c_size_t(0)
        assert s.hip.crt_debug_read_stamps(None, 0, C.byref(n)) == 0 and n.value >= (256 // 8) * (144 // 8)
        st = np.zeros((n.value, 8), np.uint64)
        assert s.hip.crt_debug_read_stamps(st.ctypes.data, n.value, C.byref(n)) == 0
        ran = st[st[:, 1] > 0]                                # waves of the list padding never start
        assert len(ran) >= (256 // 8) * (144 // 8) and (ran[:, 1] >= ran[:, 0]).all() and (ran[:, 2] > 0).all()
