"""Parity from many viewpoints: seeded random cameras -- outside, grazing and INSIDE the geometry (hazard H1: boxes that
contain the ray origin are rejected, so interior views lose most of the scene, and must lose exactly the same) -- on two
scenes, frames and counters against the oracle bit for bit, through synchronous frames, frames in flight and shadows."""
import numpy as np
import pytest

from clraytracer_amd import driver, scenes
import oracle_lib
from util import bits

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,extent", [("sponza-class-250k", 45.0), ("multi-1M", 14.0)])
def test_random_viewpoints(name, extent, nthreads):
    sc = scenes.get(name)
    rng = np.random.RandomState(99)
    W, H = 160, 96
    with driver.Session(W, H, device=0) as s:
        s.load_scene(sc)
        orc = oracle_lib.Oracle(s.arenas(), nthreads=nthreads)
        hits_total = 0
        for k in range(14):
            if k % 3 == 0:      # inside the scene's bounding volume
                pos = rng.uniform(-0.4 * extent, 0.4 * extent, 3); pos[1] = rng.uniform(0.0, 0.3 * extent)
            elif k % 3 == 1:    # far outside, looking at the middle
                pos = rng.normal(size=3); pos = pos / np.linalg.norm(pos) * extent * rng.uniform(1.5, 3.0); pos[1] = abs(pos[1])
            else:               # near the surface, grazing
                pos = rng.uniform(-extent, extent, 3); pos[1] = rng.uniform(0.5, 3.0)
            target = rng.uniform(-0.3 * extent, 0.3 * extent, 3)
            front = target - pos
            if np.linalg.norm(front) < 1e-3:
                front = np.array([0.0, 0.0, -1.0])
            s.set_camera(tuple(float(x) for x in pos), scenes._normalize(tuple(float(x) for x in front)))
            iv, ip, p = s.camera()
            rays = orc.raygen(W, H, iv, ip)
            sun = float(rng.uniform(0.0, 6.28))
            ref, st = orc.trace(rays, p, sun)
            s.render_raw(8, sun_angle=sun)
            assert np.array_equal(bits(s.read_output()), bits(ref)), (name, k)
            assert s.counters() == st, (name, k)
            s.render_raw(4, sun_angle=sun); s.render_raw(4, sun_angle=sun)
            assert np.array_equal(bits(s.read_output()), bits(ref)), (name, k, "async")
            if k % 4 == 0:
                refs, sts = orc.trace(rays, p, sun, shadows=True)
                s.render_raw(8 | 32, sun_angle=sun)
                assert np.array_equal(bits(s.read_output()), bits(refs)) and s.counters() == sts, (name, k, "shadows")
            hits_total += st["hits"]
        assert hits_total > 20000
