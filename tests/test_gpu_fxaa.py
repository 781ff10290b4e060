"""CRT_RENDER_FXAA / Renderer::SetFXAA: upstream's FXAA (kernel_main.cl:289-340, dead code upstream: call commented out at
kernel_main.cl:349) as the first PostProcess stage. Semantics are the oracle's (orc_fxaa, pinned against an independent numpy
restatement in tests/test_fxaa.py); the HIP kernel has to match it bit for bit -- the filter is + - * / floor min max only."""
import ctypes as C

import numpy as np
import pytest

from clraytracer_amd import _lib, driver, scenes
import oracle_lib
from util import bits

pytestmark = pytest.mark.gpu

POST, ASYNC, UNORM8, READBACK, FXAA = 1, 4, 64, 128, 512


@pytest.mark.parametrize("name,w,h", [("tiny", 200, 120), ("cornell-1k", 333, 187), ("tiny", 16, 16)])
def test_fxaa_stage_bit_exact_and_chain(name, w, h, nthreads):
    sc = scenes.get(name)
    with driver.Session(w, h, device=0) as s:
        s.load_scene(sc)
        orc = oracle_lib.Oracle(s.arenas(), nthreads=nthreads)
        s.render_raw(0)
        raw = s.read_output()
        s.render_raw(FXAA)
        got = s.read_output()
        ref = orc.fxaa(raw)
        assert np.array_equal(bits(got), bits(ref))
        assert not np.array_equal(bits(got), bits(raw))
        # the chain upstream sketches: FXAA -> Saturation -> Reinhard -> Gamma -> Vignette (powf: tolerance as in test_postprocess)
        s.render_raw(FXAA | POST)
        full = s.read_output()
        want = orc.postprocess(ref)
        assert np.array_equal(np.isnan(full), np.isnan(want))
        m = np.isfinite(want)
        assert np.abs(full[m] - want[m]).max() < 2e-5
        # through upstream's RGBA8 target (hazard H8): store+load before the filter reads, store after it
        s.render_raw(FXAA | UNORM8)
        assert np.array_equal(bits(s.read_output()), bits(orc.quantize_unorm8(orc.fxaa(orc.quantize_unorm8(raw)))))
        # mirrored Renderer
        s.render(postprocess=False, fxaa=True)
        assert np.array_equal(bits(s.output()), bits(ref))
        s.render(postprocess=False)
        assert np.array_equal(bits(s.output()), bits(raw))


def test_fxaa_with_frames_in_flight_resize_and_readback():
    sc = scenes.get("tiny")
    hip = _lib.hip()
    with driver.Session(256, 144, device=0) as s:
        s.load_scene(sc)
        s.render_raw(0)
        ref = oracle_lib.fxaa(s.read_output())
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        for k in range(7):                                 # every slot allocates its own unfiltered copy
            s.render_raw(ASYNC | FXAA | READBACK)
            assert hip.crt_map_host_frame(C.byref(ptr), C.byref(nbytes)) == 0
            host = np.frombuffer((C.c_char * nbytes.value).from_address(ptr.value), np.float32).reshape(144, 256, 4)
            assert np.array_equal(bits(host), bits(ref)), k
        assert np.array_equal(bits(s.read_output()), bits(ref))
        s.resize(320, 200)                                 # drops the copies; the next frame allocates the new size
        s.render_raw(0)
        ref2 = oracle_lib.fxaa(s.read_output())
        s.render_raw(ASYNC | FXAA); s.render_raw(ASYNC | FXAA)
        assert np.array_equal(bits(s.read_output()), bits(ref2))
        # a share of the rows cannot be filtered without the neighbouring bands: refused, and nothing breaks
        s.set_row_bands(16, 1, 2)
        a, iv, ip = s.trace_args()
        fp = C.POINTER(C.c_float)
        assert hip.crt_render(C.byref(a), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp), FXAA) == -5    # CRT_E_UNSUPPORTED
        s.set_row_bands(16, 0, 1)
        s.render_raw(FXAA)
        assert np.array_equal(bits(s.read_output()), bits(ref2))


@pytest.mark.parametrize("ndev", [2, 5])
def test_fxaa_in_a_multi_device_session_filters_the_gathered_frame(ndev):
    """Bands from every device are gathered raw; the first device filters (and post-processes) the whole frame."""
    sc = scenes.get("tiny")
    w, h = 328, 200
    with driver.Session(w, h, device=0) as s1:
        s1.load_scene(sc)
        s1.render_raw(FXAA); one = s1.read_output()
        s1.render_raw(FXAA | POST | UNORM8); one_post = s1.read_output()
    with driver.Session(w, h, devices=[0] * ndev) as s:
        s.load_scene(sc)
        s.render_raw(FXAA)
        assert np.array_equal(bits(s.read_output()), bits(one))
        for _ in range(5):
            s.render_raw(ASYNC | FXAA | POST | UNORM8)
        assert np.array_equal(bits(s.read_output()), bits(one_post))
        s.render(postprocess=False, fxaa=True)
        assert np.array_equal(bits(s.output()), bits(one))
