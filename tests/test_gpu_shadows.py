"""CRT_RENDER_SHADOWS (extension; upstream's shadow ray is a commented-out TODO, kernel_main.cl:256-258): the HIP path
must match the oracle's definition (orc_trace_ex, shadows=1) bit for bit, counters included, and must leave frames
rendered without the flag untouched."""
import numpy as np
import pytest

from clraytracer_amd import driver, scenes
import oracle_lib
from util import bits

pytestmark = pytest.mark.gpu
SHADOWS, COUNTERS, ASYNC = 32, 8, 4


@pytest.mark.parametrize("name,size", [("tiny", (200, 120)), ("cornell-1k", (320, 184)), ("sponza-class-250k", (480, 270))])
def test_shadow_frames_match_oracle(name, size, nthreads):
    sc = scenes.get(name)
    w, h = size
    with driver.Session(w, h, device=0) as s:
        s.load_scene(sc)
        orc = oracle_lib.Oracle(s.arenas(), nthreads=nthreads)
        iv, ip, pos = s.camera()
        rays = orc.raygen(w, h, iv, ip)
        plain, st0 = orc.trace(rays, pos, sc.sun_angle)
        ref, st = orc.trace(rays, pos, sc.sun_angle, shadows=True)
        assert st["shadowRays"] > 0 and st0["shadowRays"] == 0
        assert st["rays"] == st0["rays"] + st["shadowRays"] and st["shadowHits"] <= st["shadowRays"]
        for _ in range(3):                                   # identity lists, then feedback lists
            s.render_raw(SHADOWS)
            assert np.array_equal(bits(s.read_output()), bits(ref))
        s.render_raw(SHADOWS | COUNTERS)
        assert np.array_equal(bits(s.read_output()), bits(ref))
        assert s.counters() == st
        s.render_raw(SHADOWS | ASYNC); s.render_raw(SHADOWS | ASYNC); s.render_raw(SHADOWS | ASYNC)
        assert np.array_equal(bits(s.read_output()), bits(ref))
        s.render_raw(COUNTERS)                               # the flag off: upstream semantics, unchanged
        assert np.array_equal(bits(s.read_output()), bits(plain)) and s.counters() == st0
        if st["shadowHits"]:
            assert not np.array_equal(bits(ref), bits(plain))
        # the mirrored Renderer: SetShadows / SetPipelined
        s.render(postprocess=False, shadows=True, pipelined=True); s.render(postprocess=False, shadows=True, pipelined=True)
        assert np.array_equal(bits(s.output()), bits(ref))
        s.render(postprocess=False)
        assert np.array_equal(bits(s.output()), bits(plain))


def test_shadows_many_instances(nthreads):
    """More than one candidate chunk (401 instances) and early exit across instances: counters still exact."""
    sc = scenes.get("tiny")
    with driver.Session(160, 96, device=0) as s:
        s.load_scene(sc)
        base = len(sc.instances)
        s.h.crth_begin_instances()
        for k in range(base, 130):
            m = scenes._trs(0.4 + 0.05 * (k % 7), (0.2, 1.0, 0.1), 0.3 * k, ((k % 13) - 6.0, 2.0 + (k % 5), -3.0 - (k % 11)))
            from clraytracer_amd import _lib
            p, keep = _lib.fptr(m)
            s.h.crth_register_instance(k % 2, 0xFFFF, p)
        s.h.crth_end_instances()
        s.render(postprocess=False)                          # uploads the new instances
        orc = oracle_lib.Oracle(s.arenas(), nthreads=nthreads)
        iv, ip, pos = s.camera()
        ref, st = orc.trace(orc.raygen(160, 96, iv, ip), pos, sc.sun_angle, shadows=True)
        s.render_raw(SHADOWS | COUNTERS)
        assert np.array_equal(bits(s.read_output()), bits(ref)) and s.counters() == st
        assert st["shadowHits"] > 0


@pytest.mark.parametrize("name", ["tiny", "sponza-class-250k"])
def test_unorm8_render_target_like_upstream(name, nthreads):
    """CRT_RENDER_UNORM8 (hazard H8): the frame upstream actually displays -- Trace quantised into the RGBA8 texture,
    PostProcess reading that back, the result quantised again -- against the oracle composition. The quantised Trace
    stage is bit-exact; after PostProcess (powf: <= 2e-5 apart) a byte may differ by one code at a rounding boundary."""
    sc = scenes.get(name)
    UNORM8, POST = 64, 1
    with driver.Session(320, 184, device=0) as s:
        s.load_scene(sc)
        orc = oracle_lib.Oracle(s.arenas(), nthreads=nthreads)
        iv, ip, pos = s.camera()
        pre, _ = orc.trace(orc.raygen(320, 184, iv, ip), pos, sc.sun_angle)
        q = orc.quantize_unorm8(pre)
        s.render_raw(UNORM8)
        assert np.array_equal(bits(s.read_output()), bits(q))
        rgba = np.zeros((184, 320, 4), np.uint8)
        from clraytracer_amd import _lib
        assert _lib.hip().crt_read_output_rgba8(rgba.ctypes.data, rgba.size) == 0
        assert np.array_equal(rgba, orc.pack_unorm8(pre))
        want = orc.pack_unorm8(orc.postprocess(q))
        s.render_raw(UNORM8 | POST)
        assert _lib.hip().crt_read_output_rgba8(rgba.ctypes.data, rgba.size) == 0
        d = np.abs(rgba.astype(np.int16) - want.astype(np.int16))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3, (d.max(), (d > 0).mean())
        # the mirrored Renderer: upstream's Render() = UNORM8 target + PostProcess
        s.h.crth_set_unorm8(1)
        s.render(postprocess=True)
        got = _lib.as_array(s.h.crth_map_output_rgba8(), 320 * 184 * 4, np.uint8).reshape(184, 320, 4)
        s.h.crth_set_unorm8(0)
        assert np.array_equal(got, rgba)
