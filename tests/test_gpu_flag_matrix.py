"""Every combination of crt_render flags that may be combined, on one small scene: the frame must be the composition the flags
stand for -- Trace (plain / shadow rays / refraction, kernel_main.cl:164-275 and the oracle's extensions) -> RGBA8 store+load
(hazard H8) -> FXAA -> PostProcess -> RGBA8 store -- whichever launch structure the combination selects (counting, stamped,
pipelined, read-back; stages fused into Trace's or the filter's epilogue, or run as launches of their own)."""
import ctypes as C
import itertools

import numpy as np
import pytest

from clraytracer_amd import _lib, driver, scenes
import oracle_lib
from util import bits

pytestmark = pytest.mark.gpu

POST, ASYNC, COUNT, STAMPS, SHADOWS, UNORM8, READBACK, REFRACT, FXAA = 1, 4, 8, 16, 32, 64, 128, 256, 512


@pytest.mark.parametrize("kind", ["one-device", "three-device-states", "wavefront-variant"])
def test_all_flag_combinations(nthreads, monkeypatch, kind):
    sc = scenes.get("tiny")
    W, H = 112, 72
    hip = _lib.hip()
    if kind == "wavefront-variant":
        monkeypatch.setenv("CRT_KERNEL", "wavefront")          # per-bounce launches; every per-pixel stage is a launch of its own
    else:
        monkeypatch.delenv("CRT_KERNEL", raising=False)
    plain_only = kind == "wavefront-variant"                    # shadow rays / refraction / stamps belong to the default kernel
    with driver.Session(W, H, **({"devices": [0, 0, 0]} if kind == "three-device-states" else {"device": 0})) as s:
        s.load_scene(sc)
        orc = oracle_lib.Oracle(s.arenas(), nthreads=nthreads)
        iv, ip, pos = s.camera()
        rays = orc.raygen(W, H, iv, ip)
        base = {(sh, rf): orc.trace(rays, pos, sc.sun_angle, shadows=bool(sh), refraction=bool(rf)) for sh in (0, 1) for rf in (0, 1)}
        assert not np.array_equal(bits(base[0, 0][0]), bits(base[1, 0][0])) and not np.array_equal(bits(base[0, 0][0]), bits(base[0, 1][0]))
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        checked = 0
        for sh, rf, cnt, asy, rb, stamps in itertools.product((0, 1), repeat=6):
            if stamps and (sh or rf or cnt or asy or kind != "one-device"):
                continue                                       # the stamped launch is a diagnostic of the plain synchronous frame on one device
            if plain_only and (sh or rf):
                continue
            for stage in range(8):
                post, unorm, fxaa = stage & 1, stage & 2, stage & 4
                flags = (SHADOWS * sh) | (REFRACT * rf) | (COUNT * cnt) | (ASYNC * asy) | (READBACK * rb) | (STAMPS * stamps) \
                    | (POST if post else 0) | (UNORM8 if unorm else 0) | (FXAA if fxaa else 0)
                want, st = base[sh, rf]
                if unorm:
                    want = orc.quantize_unorm8(want)
                if fxaa:
                    want = orc.fxaa(want)
                if post:
                    want = orc.postprocess(want)
                if unorm and (post or fxaa):
                    want = orc.quantize_unorm8(want)
                s.render_raw(flags)
                # the kernel that rendered it is the one the combination stands for (crt_debug_last_kernel)
                want_kernel = ("crt_primary_kernel<%d>+crt_wavefront_scan_kernel+crt_bounce_kernel<%d>" % (cnt, cnt)) if plain_only else \
                    ("crt_trace_kernel<0,1,0,0,0>" if stamps else "crt_trace_kernel<%d,0,%d,0,%d>" % (cnt, sh, rf))
                assert s.last_kernel() == want_kernel, (flags, s.last_kernel())
                got = s.read_output()
                if post:                                       # powf: 2e-5; through the RGBA8 store that can move a value by one code
                    tol = (1.0 / 255.0 + 1e-6) if unorm else 2e-5
                    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
                    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.nanmax(d) <= tol, (flags, float(np.nanmax(d)))
                    assert (d > 2e-5).sum() <= 0.002 * d.size, (flags, int((d > 2e-5).sum()))
                else:
                    assert np.array_equal(bits(got), bits(want)), flags
                if cnt:
                    assert s.counters() == st, flags
                if rb:
                    assert hip.crt_map_host_frame(C.byref(ptr), C.byref(nbytes)) == 0, flags
                    if unorm:                                  # RGBA8 bytes of the same frame
                        host = np.frombuffer((C.c_char * nbytes.value).from_address(ptr.value), np.uint8).reshape(H, W, 4)
                        assert nbytes.value == W * H * 4 and np.array_equal(host, np.rint(np.nan_to_num(got) * 255.0).astype(np.uint8)), flags
                    else:
                        host = np.frombuffer((C.c_char * nbytes.value).from_address(ptr.value), np.float32).reshape(H, W, 4)
                        assert nbytes.value == W * H * 16 and np.array_equal(bits(host), bits(got)), flags
                checked += 1
        assert checked == {"one-device": (32 + 2) * 8, "three-device-states": 32 * 8, "wavefront-variant": 8 * 8}[kind]
