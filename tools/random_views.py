#!/usr/bin/env python3
"""Full-size parity sweep, run once on the GPU box: every scene at 1920x1080 from seeded random viewpoints -- outside, grazing
and inside the geometry (hazard H1) -- HIP frame and work counters against the oracle, plain and with the shadow-ray
extension. Prints one line per scene and a total; exits non-zero on any counter difference or on more differing pixels than
the skybox-texel tolerance. Under CRT_KERNEL=wavefront|refill|block: no shadow rays (a form refuses them with
CRT_E_UNSUPPORTED) and every frame's kernel name (crt_debug_last_kernel) is checked against the form.
    python tools/random_views.py [views_per_scene]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from clraytracer_amd import driver, scenes  # noqa: E402
import oracle_lib  # noqa: E402

views = int(sys.argv[1]) if len(sys.argv) > 1 else 12
W, H = 1920, 1080
SCENES = [("cornell-1k", 4.0), ("sponza-class-250k", 45.0), ("multi-1M", 14.0), ("multi-1M-dense", 14.0), ("sponza-sibenik", 20.0), ("nanosuit-demo", 12.0)]
threads = min(64, len(os.sched_getaffinity(0)))
form = os.environ.get("CRT_KERNEL") or "default"
prefix = {"default": "crt_trace_kernel<", "wavefront": "crt_primary_kernel<", "refill": "crt_trace_refill_kernel<", "block": "crt_trace_block_kernel<", "ldstop": "crt_trace_ldstop_kernel<"}[form]
grand = {"frames": 0, "rays": 0, "pixels_differing": 0}
t_all = time.time()
for name, extent in SCENES:
    sc = scenes.get(name)
    rng = np.random.RandomState(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
    with driver.Session(W, H, device=0) as s:
        s.load_scene(sc)
        orc = oracle_lib.Oracle(s.arenas(), nthreads=threads)
        rays_total, diff_total, hits_total, t0 = 0, 0, 0, time.time()
        for k in range(views):
            if k % 3 == 0:
                pos = rng.uniform(-0.4 * extent, 0.4 * extent, 3); pos[1] = rng.uniform(0.0, 0.3 * extent)
            elif k % 3 == 1:
                pos = rng.normal(size=3); pos = pos / np.linalg.norm(pos) * extent * rng.uniform(1.5, 3.0); pos[1] = abs(pos[1])
            else:
                pos = rng.uniform(-extent, extent, 3); pos[1] = rng.uniform(0.5, 3.0)
            front = rng.uniform(-0.3 * extent, 0.3 * extent, 3) - pos
            if np.linalg.norm(front) < 1e-3:
                front = np.array([0.0, 0.0, -1.0])
            s.set_camera(tuple(float(x) for x in pos), scenes._normalize(tuple(float(x) for x in front)))
            iv, ip, p = s.camera()
            rays = orc.raygen(W, H, iv, ip)
            sun = float(rng.uniform(0.0, 6.28))
            shadows = k % 2 == 1 and form == "default"
            ref, st = orc.trace(rays, p, sun, shadows=shadows)
            s.render_raw(8 | (32 if shadows else 0), sun_angle=sun)
            if not s.last_kernel().startswith(prefix):
                print(f"{name} view {k}: rendered by {s.last_kernel()}, not by the {form} form"); sys.exit(1)
            got = s.read_output()
            nd = int((got.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())
            if s.counters() != st or nd > max(2, int(1e-5 * W * H)):
                print(f"{name} view {k}: counters equal {s.counters() == st}, {nd} pixels differ"); sys.exit(1)
            s.render_raw(4 | (32 if shadows else 0), sun_angle=sun); s.render_raw(4 | (32 if shadows else 0), sun_angle=sun)
            if not np.array_equal(s.read_output().view(np.uint32), got.view(np.uint32)):
                print(f"{name} view {k}: frames in flight differ from the synchronous frame"); sys.exit(1)
            rays_total += st["rays"]; diff_total += nd; hits_total += st["hits"]
        print(f"{name}: {views} views at {W}x{H} ({'odd ones with shadow rays' if form == 'default' else 'no shadow rays: the form refuses them'}; kernel form {form}): {rays_total} rays, {hits_total} hits, counters exact, "
              f"{diff_total} pixels differing in total (skybox texel flips), {time.time() - t0:.0f} s", flush=True)
        grand["frames"] += views; grand["rays"] += rays_total; grand["pixels_differing"] += diff_total
print(f"total, kernel form {form} (every frame rendered by {prefix}...>): {grand['frames']} full-size frames, {grand['rays']} rays, {grand['pixels_differing']} differing pixels, {time.time() - t_all:.0f} s on {threads} host threads")
