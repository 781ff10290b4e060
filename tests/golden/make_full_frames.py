#!/usr/bin/env python3
"""Known answers at BASELINE.json's full sizes, from the CPU ORACLE (never from the HIP path): SHA-256 of the 1920x1080
pre-PostProcess float frame, of the shadow-extension frame and of the RGBA8 (hazard H8) bytes, plus the work counters,
for the synthetic scenes and the two scenes made of the reference's shipped assets -> tests/golden/full_frames.json. Run from the repo root (about a minute on 8 cores):
    python tests/golden/make_full_frames.py
The scenes are generated from seeds (clraytracer_amd/scenes.py) or loaded from assets/, so the fixture is
machine-independent."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

from clraytracer_amd import driver, scenes  # noqa: E402
import oracle_lib  # noqa: E402

COUNTERS = ["rays", "primary", "secondary", "hits", "misses", "traversals", "pops", "innerVisits", "triTests", "capHits",
            "stackOverflows", "maxStack", "shadowRays", "shadowHits"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    out = {}
    w, h = 1920, 1080
    for name in ("tiny", "cornell-1k", "sponza-class-250k", "multi-1M", "multi-1M-dense", "sponza-sibenik", "nanosuit-demo"):
        sc = scenes.get(name)
        with driver.Session(w, h, host_only=True) as s:
            s.load_scene(sc)
            orc = oracle_lib.Oracle(s.arenas(), nthreads=os.cpu_count() or 1)
            iv, ip, pos = s.camera()
            rays = orc.raygen(w, h, iv, ip)
            pre, st = orc.trace(rays, pos, sc.sun_angle)
            shad, sts = orc.trace(rays, pos, sc.sun_angle, shadows=True)
            out[name] = {"width": w, "height": h, "triangles": int(len(s.arenas()["tris"])),
                         "rays_sha256": sha(rays), "frame_sha256": sha(pre), "rgba8_sha256": sha(orc.pack_unorm8(pre)),
                         "shadow_frame_sha256": sha(shad),
                         "counters": {k: int(st[k]) for k in COUNTERS}, "shadow_counters": {k: int(sts[k]) for k in COUNTERS}}
            print(name, out[name]["counters"]["rays"], out[name]["frame_sha256"][:16])
    json.dump(out, open(os.path.join(HERE, "full_frames.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
