"""Committed golden fixtures (tests/golden/, produced by the oracle via make_golden.py) pin the oracle against
regressions on CPU and give the GPU path known answers that need no oracle run."""
import os

import numpy as np
import pytest

from clraytracer_amd import driver, scenes
import oracle_lib
from util import bits

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["tiny", "cornell-1k"]


def load(name):
    g = np.load(os.path.join(GOLDEN, "frames_%s.npz" % name), allow_pickle=False)
    return g, scenes.get(name)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(name, nthreads):
    g, sc = load(name)
    h, w, _ = g["rays"].shape
    with driver.Session(w, h, host_only=True) as s:
        s.load_scene(sc)
        iv, ip, pos = s.camera()
        orc = oracle_lib.Oracle(s.arenas(), nthreads=nthreads)
    # camera matrices of the mirrored Camera (polynomial Sin/Cos, pinned rsqrt) are part of the fixture
    assert np.array_equal(bits(iv), bits(g["inv_view"])) and np.array_equal(bits(ip), bits(g["inv_proj"]))
    rays = orc.raygen(w, h, iv, ip)
    assert np.array_equal(bits(rays), bits(g["rays"]))
    pre, st = orc.trace(rays, pos, float(g["sun_angle"]))
    assert np.array_equal(bits(pre), bits(g["pre"]))
    assert [st[k] for k in g["stat_keys"].tolist()] == g["stats"].tolist()
    post = orc.postprocess(pre)
    assert np.array_equal(np.isnan(post), np.isnan(g["post"]))
    m = np.isfinite(g["post"])
    assert np.array_equal(bits(post[m]), bits(g["post"][m]))
    hits, hst = orc.closest_hits(g["ray_o"], g["ray_d"])
    assert hits.tobytes() == g["hits"].tobytes()
    assert [hst[k] for k in g["stat_keys"].tolist()] == g["hit_stats"].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_reproduces_golden(name):
    g, sc = load(name)
    h, w, _ = g["rays"].shape
    with driver.Session(w, h, device=0) as s:
        s.load_scene(sc)
        s.render_raw(2 | 8)                     # WRITE_RAYS | COUNTERS
        assert np.array_equal(bits(s.read_rays()), bits(g["rays"]))
        pre = s.read_output()
        d = np.abs(pre[..., :3].astype(np.float64) - g["pre"][..., :3].astype(np.float64))
        assert np.sqrt(np.mean(d * d)) < 1e-4 and (d.max(-1) > 1e-5).sum() <= 1
        cnt = s.counters()
        assert [cnt[k] for k in g["stat_keys"].tolist()] == g["stats"].tolist()
        s.render_raw(1)                         # POSTPROCESS
        post = s.read_output()
        m = np.isfinite(g["post"])
        assert np.array_equal(np.isnan(post), np.isnan(g["post"])) and np.abs(post[m] - g["post"][m]).max() < 2e-5
        hits = s.query_hits(g["ray_o"], g["ray_d"])
        ref = g["hits"]
        assert np.array_equal(hits["instance"], ref["instance"]) and np.array_equal(hits["tri"], ref["tri"])
        for f in ("t", "u", "v"):
            assert np.array_equal(bits(hits[f]), bits(ref[f]))
