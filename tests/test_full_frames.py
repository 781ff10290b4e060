"""BASELINE.json's full-size frames against committed known answers (tests/golden/full_frames.json, written by
tests/golden/make_full_frames.py from the CPU oracle): SHA-256 of the 1920x1080 float frame, of the shadow-extension
frame, of the RGBA8 bytes, and the work counters, for the four synthetic scenes. The CPU test pins the oracle (and the
scene generators / importer / BVH builder feeding it); the GPU test pins the HIP path without running the oracle."""
import hashlib
import json
import os

import numpy as np
import pytest

from clraytracer_amd import _lib, driver, scenes
import oracle_lib

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full_frames.json")))
NAMES = list(GOLD)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_full_size_answers(name, nthreads):
    g = GOLD[name]
    sc = scenes.get(name)
    with driver.Session(g["width"], g["height"], host_only=True) as s:
        s.load_scene(sc)
        a = s.arenas()
        assert len(a["tris"]) == g["triangles"]
        orc = oracle_lib.Oracle(a, nthreads=nthreads)
        iv, ip, pos = s.camera()
        rays = orc.raygen(g["width"], g["height"], iv, ip)
        assert sha(rays) == g["rays_sha256"]
        pre, st = orc.trace(rays, pos, sc.sun_angle)
        assert sha(pre) == g["frame_sha256"] and {k: st[k] for k in g["counters"]} == g["counters"]
        assert sha(orc.pack_unorm8(pre)) == g["rgba8_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_path_reproduces_full_size_answers(name):
    g = GOLD[name]
    sc = scenes.get(name)
    hip = _lib.hip()
    with driver.Session(g["width"], g["height"], device=0) as s:
        s.load_scene(sc, device_bvh_build=(name == "sponza-class-250k"))       # one scene through crt_build_bvh
        s.render_raw(2)                                                        # WRITE_RAYS
        assert sha(s.read_rays()) == g["rays_sha256"]
        s.render_raw(8)
        assert sha(s.read_output()) == g["frame_sha256"] and {k: s.counters()[k] for k in g["counters"]} == g["counters"]
        rgba = np.zeros((g["height"], g["width"], 4), np.uint8)
        assert hip.crt_read_output_rgba8(rgba.ctypes.data, rgba.size) == 0 and sha(rgba) == g["rgba8_sha256"]
        for _ in range(4):
            s.render_raw(4)                                                    # frames in flight
        assert sha(s.read_output()) == g["frame_sha256"]
        s.render_raw(8 | 32)
        assert sha(s.read_output()) == g["shadow_frame_sha256"] and {k: s.counters()[k] for k in g["shadow_counters"]} == g["shadow_counters"]
