"""The conservative instance cull at the edge of, and beyond, the range it is proven for.

crt_device.h (above sphere_culls) derives the fp32 error of the object-space transform + slab tests against the
double-precision world-space sphere and the range of ray origins |o| <= O_i for which the cull's slack
(1.0201 r^2 + 4e-6 |oc|^2) dominates it; crt_instances.h (rebuild_instance_master) stores O_i per instance, never culls an instance whose O_i is below
the reach of bounce-ray origins, and runs a frame / query whose origins lie beyond the smallest O_i without the cull.
Here: instance scales 1e-3 ... 1e3, Frobenius condition numbers to ~300, instances up to 3e5 units from the world
origin, ray origins out to 1e6 units and to 0.95 O_i, rays that graze the bounding spheres and the corners of the boxes
they are built around. Hit records bit-exact and work counters equal to the oracle (which has no cull: a wrongly culled
instance whose child box the ray does pass shows up in innerVisits even when no triangle is hit).
Reference: the per-ray instance loop without any cull, kernel_main.cl:198-217.
"""
import ctypes as C

import numpy as np
import pytest

from clraytracer_amd import _lib, driver, scenes
import oracle_lib
from util import bits, rmse

pytestmark = pytest.mark.gpu


def _scene(instances, name):
    base = scenes.get("tiny")
    return scenes.Scene(name, base.dir, base.skybox, base.meshes, instances, (0.0, 2.0, 30.0), (0.0, 0.0, -1.0))


def _nonuniform(sx, sy, sz, axis, angle, t):
    m = scenes._trs(1.0, axis, angle, (0, 0, 0)).astype(np.float64)
    m[:3, :3] = np.diag([sx, sy, sz]) @ m[:3, :3]
    m[3, :3] = t
    return m.astype(np.float32)


SETS = {
    # smallest O_i first: the 1e-3-scale instance (w ~ 3e-3) may be culled for origins up to ~40 units only
    "scales-1e-3-to-1e3": [scenes.Instance(0, 0xFFFF, scenes._trs(1e-3, (0.2, 1.0, 0.1), 0.7, (0.01, 0.02, -0.03))),
                           scenes.Instance(1, 0xFFFF, scenes._trs(1.0, (1.0, 0.3, 0.2), 1.9, (20.0, 5.0, -30.0))),
                           scenes.Instance(0, 0xFFFF, scenes._trs(1e3, (0.1, 0.2, 1.0), 2.6, (5e4, 1e4, -8e4))),
                           scenes.Instance(1, 0xFFFF, _nonuniform(1.0, 30.0, 0.2, (0.3, 0.5, 1.0), 0.4, (100.0, 0.0, 0.0)))],
    # condition numbers ~150 and ~300, and an instance so far out that it is never culled (O_i < 0)
    "ill-conditioned": [scenes.Instance(0, 0xFFFF, _nonuniform(1.0, 30.0, 0.2, (0.3, 0.5, 1.0), 0.4, (100.0, 0.0, 0.0))),
                        scenes.Instance(1, 0xFFFF, _nonuniform(60.0, 0.5, 1.0, (1.0, 0.1, 0.4), 2.2, (-300.0, 40.0, 10.0))),
                        scenes.Instance(0, 0xFFFF, scenes._trs(1.0, (0.0, 1.0, 0.0), 0.3, (3e5, 0.0, 0.0))),
                        scenes.Instance(1, 0xFFFF, scenes._trs(8.0, (0.5, 1.0, 0.0), 1.1, (0.0, -50.0, 200.0)))],
    # large instances only: the cull stays on for cameras 1e5 ... 1e6 units away
    "huge": [scenes.Instance(0, 0xFFFF, scenes._trs(1e3, (0.1, 0.2, 1.0), 2.6, (5e4, 1e4, -8e4))),
             scenes.Instance(1, 0xFFFF, scenes._trs(2.5e3, (1.0, 0.2, 0.3), 0.9, (-2e5, 3e4, 1e5))),
             scenes.Instance(0, 0xFFFF, scenes._trs(4e2, (0.0, 1.0, 0.2), 4.0, (1e4, -2e4, 3e4)))],
}


def _cull_range(s, n):
    lim = np.zeros(n, np.float32)
    scene_lim = C.c_float(); reach = C.c_float(); frames = C.c_uint64()
    _lib.check(s.hip.crt_get_cull_range(lim.ctypes.data_as(C.POINTER(C.c_float)), n, C.byref(scene_lim), C.byref(reach), C.byref(frames)), "crt_get_cull_range")
    return lim, float(scene_lim.value), float(reach.value), int(frames.value)


def _culled(s):
    v = C.c_uint64()
    _lib.check(s.hip.crt_get_culled_visits(C.byref(v)), "crt_get_culled_visits")
    return int(v.value)


def _child_boxes(a, mesh):
    root = a["nodes"][a["roots"][mesh]]
    assert root["triCount"] == 0
    return a["nodes"][root["leftFirst"]], a["nodes"][root["leftFirst"] + 1]


def _grazing_rays(a, rng, origin_norm, per_instance):
    """Rays from origins `origin_norm` from the world origin whose LINES pass each instance's bounding sphere at 0.9 ... 1.1 radii,
    and rays aimed at (and just past) the world images of the corners of the root's child boxes, where the sphere touches the box."""
    os_, ds_ = [], []
    for inst in a["instances"]:
        fwd = np.linalg.inv(inst["inv"].astype(np.float64))
        k0, k1 = _child_boxes(a, int(inst["meshIndex"]))
        lo = np.minimum(k0["min"], k1["min"]).astype(np.float64); hi = np.maximum(k0["max"], k1["max"]).astype(np.float64)
        corners = np.array([[(hi if (k >> b) & 1 else lo)[b] for b in range(3)] for k in range(8)])
        cw = np.concatenate([corners, np.ones((8, 1))], 1) @ fwd
        cw = cw[:, :3]
        c = (np.append((lo + hi) / 2, 1.0) @ fwd)[:3]
        r = np.linalg.norm(cw - c, axis=1).max()
        n = per_instance
        o = rng.normal(size=(n, 3)); o *= (origin_norm * rng.uniform(0.3, 1.0, n) / np.linalg.norm(o, axis=1))[:, None]
        # every other origin 1.5 ... 30 radii from the instance instead, where the allowed range reaches that far: upstream's rays end at
        # t = 99999 (MathAndSTL.cl:123), so only origins within 1e5 units of an instance can hit it at all
        near = rng.normal(size=(n, 3)); near = c + near * (r * rng.uniform(1.5, max(2.0, min(30.0, 9e4 / r)), n) / np.linalg.norm(near, axis=1))[:, None]
        use = (np.arange(n) % 2 == 0) & (np.linalg.norm(near, axis=1) <= origin_norm)
        o[use] = near[use]
        v = c - o
        t = np.cross(v, rng.normal(size=(n, 3))); t /= np.linalg.norm(t, axis=1, keepdims=True)      # unit, perpendicular to the line of sight
        rho = r * rng.uniform(0.9, 1.1, n)
        rho[: n // 4] = r * (1.0 + rng.uniform(-2e-2, 2e-2, n // 4))                                 # a quarter within 2 % of tangency
        tgt = c + t * rho[:, None]
        # second family: at the box corners, offset by 1e-7 ... 1e-2 radii in a random direction
        kk = rng.randint(0, 8, n // 2)
        off = rng.normal(size=(n // 2, 3)); off *= (r * 10.0 ** rng.uniform(-7, -2, n // 2) / np.linalg.norm(off, axis=1))[:, None]
        tgt[: n // 2] = np.where(rng.uniform(size=(n // 2, 1)) < 0.5, tgt[: n // 2], cw[kk] + off)
        # ... and every fifth ray at the middle of the instance, to see geometry through all that arithmetic
        mid = np.arange(n) % 5 == 4
        tgt[mid] = c + rng.normal(size=(int(mid.sum()), 3)) * (0.25 * r)
        d = tgt - o
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        os_.append(o); ds_.append(d)
    return np.concatenate(os_).astype(np.float32), np.concatenate(ds_).astype(np.float32)


@pytest.fixture(scope="module", params=sorted(SETS))
def sess(request, nthreads):
    sc = _scene(SETS[request.param], "cull-" + request.param)
    s = driver.Session(256, 144, device=0)
    s.load_scene(sc)
    orc = oracle_lib.Oracle(s.arenas(), nthreads=nthreads)
    yield request.param, sc, s, orc
    s.close()


def _same_records(s, orc, o, d):
    gpu = s.query_hits(o, d)
    ref, st = orc.closest_hits(o, d)
    assert np.array_equal(gpu["instance"], ref["instance"]) and np.array_equal(gpu["tri"], ref["tri"])
    for f in ("t", "u", "v"):
        assert np.array_equal(bits(gpu[f]), bits(ref[f])), f
    assert s.counters() == st
    return ref, st


def test_limits_are_what_the_derivation_says(sess):
    name, sc, s, orc = sess
    a = s.arenas()
    lim, scene_lim, reach, _ = _cull_range(s, len(a["instances"]))
    print(name, "O_i =", lim, "scene limit", scene_lim, "bounce reach", reach)
    U = 2.0 ** -24; g3 = 3 * U / (1 - 3 * U); g4 = 4 * U / (1 - 4 * U)
    for i, inst in enumerate(a["instances"]):
        inv = inst["inv"].astype(np.float64); fwd = np.linalg.inv(inv)
        kappa = np.linalg.norm(inv[:3, :3]) * np.linalg.norm(fwd[:3, :3]); tau = np.linalg.norm(inv[3, :3]) * np.linalg.norm(fwd[:3, :3])
        k0, k1 = _child_boxes(a, int(inst["meshIndex"]))
        lo = np.minimum(k0["min"], k1["min"]).astype(np.float64); hi = np.maximum(k0["max"], k1["max"]).astype(np.float64)
        corners = np.array([[(hi if (k >> b) & 1 else lo)[b] for b in range(3)] for k in range(8)])
        cw = (np.concatenate([corners, np.ones((8, 1))], 1) @ fwd)[:, :3]
        w = np.linalg.norm(cw - (np.append((lo + hi) / 2, 1.0) @ fwd)[:3], axis=1).max() * (1 + 1e-4)
        c1 = (1 + 3 ** 0.5) * g3 * kappa
        expect = (w * ((1.02 * (1 - c1 * c1 / 2.8e-6)) ** 0.5 - 1 - c1) - g4 * tau) / (g4 * kappa)
        if expect < reach:
            assert lim[i] == 0.0, (i, expect, lim[i])
        else:
            assert abs(lim[i] - expect) <= 2e-3 * expect, (i, expect, lim[i])
    cullable = lim[lim > 0]
    assert len(cullable) and abs(scene_lim - cullable.min()) <= 1e-5 * scene_lim
    if name == "ill-conditioned":
        assert lim[2] == 0.0                      # 3e5 units out at unit scale: no origin range left
    if name == "huge":
        assert scene_lim > 2e6                    # cameras at 1e5 ... 1e6 keep the cull


def test_grazing_rays_inside_the_proven_range(sess):
    name, sc, s, orc = sess
    a = s.arenas()
    lim, scene_lim, reach, frames0 = _cull_range(s, len(a["instances"]))
    rng = np.random.RandomState(11)
    total_culled = 0
    for frac in (0.95, 0.3):
        o, d = _grazing_rays(a, rng, frac * scene_lim, 6000)
        assert np.linalg.norm(o.astype(np.float64), axis=1).max() <= scene_lim
        ref, st = _same_records(s, orc, o, d)
        total_culled += _culled(s)
        print(name, f"origins to {frac} x {scene_lim:.4g}: {len(o)} rays, {(ref['instance'] >= 0).sum()} hits, innerVisits {st['innerVisits']}, culled visits {_culled(s)}")
        assert (ref["instance"] >= 0).sum() > 300      # the rays do reach geometry
    assert total_culled > 0                       # the cull was on and did answer visits
    assert _cull_range(s, 0)[3] == frames0        # no launch had to drop it


def test_origins_beyond_the_range_run_without_the_cull(sess):
    name, sc, s, orc = sess
    a = s.arenas()
    lim, scene_lim, reach, frames0 = _cull_range(s, len(a["instances"]))
    rng = np.random.RandomState(12)
    for norm in (2.0 * scene_lim, 1e5, 1e6, 100.0 * scene_lim):
        if norm <= scene_lim:
            continue
        o, d = _grazing_rays(a, rng, norm, 3000)
        o[0] = o[0] / np.linalg.norm(o[0]) * np.float32(norm * 1.0001)      # at least one origin really is outside
        _same_records(s, orc, o, d)
        assert _culled(s) == 0
    assert _cull_range(s, 0)[3] > frames0


def test_frames_from_far_cameras(sess):
    """Whole frames (both bounces, so bounce-ray origins too): cameras at 0.9 O and at 1e5 / 1e6 units, looking at an instance."""
    name, sc, s, orc = sess
    a = s.arenas()
    lim, scene_lim, reach, _ = _cull_range(s, len(a["instances"]))
    rng = np.random.RandomState(13)
    for norm in (0.9 * scene_lim, 1e5, 1e6):
        for k, inst in enumerate(a["instances"]):
            fwd = np.linalg.inv(inst["inv"].astype(np.float64))
            c = fwd[3, :3]
            p = rng.normal(size=3); p *= norm / np.linalg.norm(p)
            front = c - p
            if not np.linalg.norm(front) > 0:
                continue
            front /= np.linalg.norm(front)
            s.set_camera(p.astype(np.float32), front.astype(np.float32))
            s.render_raw(8)
            gpu = s.read_output(); cnt = s.counters()
            iv, ip, pos = s.camera()
            rays = orc.raygen(s.width, s.height, iv, ip)
            ref, st = orc.trace(rays, pos, sc.sun_angle)
            assert cnt == st, (name, norm, k)
            dlt = np.abs(gpu[..., :3].astype(np.float64) - ref[..., :3].astype(np.float64)).max(-1)      # tolerance of test_gpu_configs.py
            assert rmse(gpu, ref) < 1e-4 and int((dlt > 1e-5).sum()) <= int(1e-5 * dlt.size) + 1, (name, norm, k)
            if np.linalg.norm(pos.astype(np.float64)) > scene_lim:
                assert _culled(s) == 0


def test_fuzz_tool_runs_clean_on_a_few_seeds():
    """tools/fuzz_cull.py (random instance matrices, origins inside / at / beyond the proven range; profiles/r04_fuzz_cull.txt) stays runnable."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_cull.py"), "5000", "6"], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0 and "all hit records and counters equal the oracle's" in p.stdout, p.stdout[-2000:]
