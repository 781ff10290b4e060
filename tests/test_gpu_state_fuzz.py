"""Stateful fuzz of the boundary: seeded random sequences of the calls a host makes between frames -- resize, camera moves,
instance moves through the mirrored Renderer (dirty-range upload), row-band changes, queries, reads -- interleaved with frames
of random flag combinations, synchronous and in flight. After every frame that is read, the pixels (and counters, when
counted) must be what the oracle computes for the state the session is in NOW: nothing stale may survive a transition
(frame slots, second buffers, launch lists, instance tables, band ownership)."""
import numpy as np
import pytest

from clraytracer_amd import _lib, driver, scenes
import oracle_lib
from util import bits

pytestmark = pytest.mark.gpu

POST, ASYNC, COUNT, SHADOWS, UNORM8, READBACK, FXAA = 1, 4, 8, 32, 64, 128, 512
SIZES = [(112, 72), (96, 96), (203, 77), (64, 48)]


def expected(orc_cache, s, sc, nthreads, flags, sun):
    iv, ip, pos = s.camera()
    key = (s.width, s.height, iv.tobytes(), ip.tobytes(), s.arenas()["instances"].tobytes(), bool(flags & SHADOWS), sun)
    if key not in orc_cache:
        orc = oracle_lib.Oracle(s.arenas(), nthreads=nthreads)
        orc_cache[key] = orc.trace(orc.raygen(s.width, s.height, iv, ip), pos, sun, shadows=bool(flags & SHADOWS))
    want, st = orc_cache[key]
    o = oracle_lib
    if flags & UNORM8:
        want = o.Oracle.quantize_unorm8(None, want)
    if flags & FXAA:
        want = o.fxaa(want)
    if flags & POST:
        want = o.Oracle.postprocess(None, want)
    if (flags & UNORM8) and (flags & (POST | FXAA)):
        want = o.Oracle.quantize_unorm8(None, want)
    return want, st


@pytest.mark.parametrize("seed,ndev", [(11, 1), (12, 1), (13, 1), (14, 1), (15, 1), (16, 1), (21, 2), (22, 3), (23, 2)])
def test_random_call_sequences(seed, ndev, nthreads):
    rng = np.random.default_rng(seed)
    sc = scenes.get("tiny")
    cache = {}
    with driver.Session(*SIZES[0], **({"device": 0} if ndev == 1 else {"devices": [0] * ndev})) as s:
        s.load_scene(sc)
        banded, band_rank = False, 0
        frames = 0
        for step in range(150):
            op = rng.choice(["frame", "frame", "frame", "resize", "camera", "instance", "bands", "query"])
            if ndev > 1 and op in ("bands", "query"):
                op = "frame"                                   # the bands belong to a multi-device session; queries are single-device diagnostics
            if op == "resize":
                w, h = SIZES[int(rng.integers(len(SIZES)))]
                s.resize(w, h)
            elif op == "camera":
                pos = (float(rng.uniform(-3, 3)), float(rng.uniform(4, 10)), float(rng.uniform(10, 18)))
                s.set_camera(pos, scenes._normalize((-pos[0] * 0.1, -0.4, -1.0)))
            elif op == "instance":
                m = sc.instances[1].matrix.copy(); m[3, :3] += rng.uniform(-1.5, 1.5, 3).astype(np.float32)
                p, keep = _lib.fptr(m)
                s.h.crth_set_mesh_matrix(1, p)
                s.render(postprocess=False)                    # the mirrored Renderer uploads the dirty range (Renderer.cpp:312-320)
            elif op == "bands":
                banded = not banded
                band_rank = int(rng.integers(2)) if banded else 0
                s.set_row_bands(16, band_rank, 2 if banded else 1)
            elif op == "query":
                o = np.tile(np.asarray(s.camera()[2], np.float32), (64, 1)); d = rng.normal(size=(64, 3)).astype(np.float32)
                d /= np.linalg.norm(d, axis=1, keepdims=True)
                got = s.query_hits(o, d)
                ref, st = oracle_lib.Oracle(s.arenas(), nthreads=nthreads).closest_hits(o, d)
                assert got.tobytes() == ref.tobytes() and s.counters() == st, (seed, step)
            else:
                flags = 0
                for f, p in ((POST, 0.3), (UNORM8, 0.3), (FXAA, 0.25), (SHADOWS, 0.3), (COUNT, 0.3), (ASYNC, 0.5), (READBACK, 0.2)):
                    if rng.random() < p:
                        flags |= f
                if banded:
                    flags &= ~FXAA                             # a share of the rows cannot be filtered (refused: CRT_E_UNSUPPORTED)
                sun = float(sc.sun_angle)
                for _ in range(int(rng.integers(1, 4))):       # the same frame once or a few times (slot rotation when ASYNC)
                    s.render_raw(flags, sun_angle=sun)
                got = s.read_output()
                want, st = expected(cache, s, sc, nthreads, flags, sun)
                if banded:
                    # only the rows this rank owns are rendered; the others hold whatever an earlier frame left there
                    own = np.array([_lib.hip().crt_row_owner(y, 16, 2) == band_rank for y in range(s.height)])
                    assert own.sum() == s.owned_rows() and 0 < own.sum() < s.height
                    got, want = got[own], want[own]
                if flags & POST:
                    tol = (1.0 / 255.0 + 1e-6) if flags & UNORM8 else 2e-5
                    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
                    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.nanmax(d) <= tol, (seed, step, flags)
                else:
                    assert np.array_equal(bits(got), bits(want)), (seed, step, flags)
                if (flags & COUNT) and not banded:             # a rank's counters cover its rows only
                    assert s.counters() == st, (seed, step, flags)
                frames += 1
        assert frames >= 20
