#!/usr/bin/env python3
"""Regenerates the committed golden fixtures from the CPU ORACLE (never from the HIP path).

The reference has no tests or golden vectors of its own and cannot be run in this image (SURVEY.md
8c), so these fixtures pin the oracle's behaviour against regressions and give the GPU tests
size-independent known answers. Run from the repo root:  python tests/golden/make_golden.py
    bvh_<scene>.json      node/triangle hashes of the oracle BVH build on the importer's output
    frames_<scene>.npz    64x36 pre- and post-PostProcess frames, matrices, 4096 seeded rays + hit records
"""
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
from util import seeded_rays  # noqa: E402


def main():
    for name in ("tiny", "cornell-1k"):
        sc = scenes.get(name)
        w, h = 64, 36
        with driver.Session(w, h, host_only=True) as s:
            s.load_scene(sc)
            a = s.arenas()
            iv, ip, pos = s.camera()
        # independent rebuild with the ORACLE builder from the importer's triangles: sort the host's
        # triangles back is impossible, so import again without building
        with driver.Session(w, h, host_only=True) as s2:
            s2.h.crth_prepare_meshes()
            s2.h.crth_import_texture(sc.skybox.encode())
            counts = []
            for p in sc.meshes:
                m = s2.h.crth_import_mesh(p.encode())
                info = np.zeros(4, np.uint32); s2.h.crth_mesh_info(m, info.ctypes.data); counts.append(int(info[0]))
            raw = s2.arenas()["tris"]
        otris, onodes, oroots, used = oracle_lib.build_bvh(raw, counts)
        json.dump({"scene": name, "num_nodes": int(used), "roots": [int(r) for r in oroots],
                   "nodes_sha256": hashlib.sha256(onodes.tobytes()).hexdigest(),
                   "tris_sha256": hashlib.sha256(otris.tobytes()).hexdigest()},
                  open(os.path.join(HERE, "bvh_%s.json" % name), "w"), indent=1)
        a = dict(a); a["tris"] = otris; a["nodes"] = onodes; a["roots"] = oroots
        orc = oracle_lib.Oracle(a, nthreads=4)
        rays = orc.raygen(w, h, iv, ip)
        pre, st = orc.trace(rays, pos, sc.sun_angle)
        post = orc.postprocess(pre)
        o, d = seeded_rays(a, pos, 4096, seed=2024)
        hits, hst = orc.closest_hits(o, d)
        np.savez_compressed(os.path.join(HERE, "frames_%s.npz" % name), inv_view=iv, inv_proj=ip, cam_pos=pos,
                            sun_angle=np.float32(sc.sun_angle), rays=rays, pre=pre, post=post,
                            stats=np.array([st[k] for k in sorted(st)], np.int64), stat_keys=np.array(sorted(st)),
                            ray_o=o, ray_d=d, hits=hits, hit_stats=np.array([hst[k] for k in sorted(hst)], np.int64))
        print(name, used, "nodes;", st)


if __name__ == "__main__":
    main()
