#!/usr/bin/env python3
"""Many-seed differential check of crt_build_bvh (device BuildBVH) against the oracle's recursive builder, on the GPU box:
    python tools/fuzz_bvh_build.py [first_seed] [count]
Every seed draws 1..4 meshes with sizes across the builder's three node classes (1 .. 60,000 triangles) and a geometry kind per mesh
(tests/test_gpu_bvh_build.py's special_tris: random, grid-snapped, signed zeros, two clusters, one centroid, sorted, reversed, failed
partitions), plus -- on some seeds -- coordinates snapped so coarsely that most centroids coincide. Triangles, nodes, roots and the node
count must equal the oracle's byte for byte. Prints one summary line; exits non-zero on the first difference."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_bvh_build as tb  # noqa: E402
from clraytracer_amd import _lib, driver  # noqa: E402
import oracle_lib  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
KINDS = ["random", "grid", "signed-zeros", "two-clusters", "same-centroid", "sorted", "reversed", "degenerate-right", "degenerate-left"]
SIZES = [1, 2, 3, 8, 9, 17, 64, 65, 128, 129, 700, 1024, 1025, 2048, 2049, 3000, 5000, 9000, 20000, 60000]
t0 = time.time()
tot = {"builds": 0, "meshes": 0, "tris": 0, "nodes": 0, "leaf3": 0}
with driver.Session(64, 48, device=0) as s:
    hip = _lib.hip()
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        counts = [int(rng.choice(SIZES)) for _ in range(int(rng.integers(1, 5)))]
        parts = []
        for j, n in enumerate(counts):
            kind = str(rng.choice(KINDS, p=[0.3, 0.2, 0.1, 0.1, 0.05, 0.05, 0.05, 0.075, 0.075]))
            t = tb.special_tris(kind if n >= 2 else "random", n, 1000 * seed + j)
            if rng.random() < 0.2:                                  # coarse snap: many equal centroids, many exact ties with split planes
                q = np.float32(rng.choice([4.0, 8.0, 16.0]))
                for k in ("v0", "v1", "v2"):
                    t[k] = np.round(t[k] / q) * q
            parts.append(t)
        tris = np.ascontiguousarray(np.concatenate(parts))
        ot, on, oroots, ou = oracle_lib.build_bvh(tris, counts)
        dt_, dn, dr, du, _ = tb.device_build(hip, tris, counts)
        if du != ou or not np.array_equal(dr, oroots) or dt_.tobytes() != ot.tobytes() or dn.tobytes() != on.tobytes():
            print(f"seed {seed}: device build differs from the oracle's (mesh sizes {counts})"); sys.exit(1)
        tot["builds"] += 1; tot["meshes"] += len(counts); tot["tris"] += len(tris); tot["nodes"] += int(ou)
        if (seed - first + 1) % 200 == 0:
            print(f"  ... {seed - first + 1} builds, all equal so far, {time.time() - t0:.0f} s", flush=True)
print(f"BVH build fuzz seeds {first}..{first + count - 1}: {tot['builds']} builds, {tot['meshes']} meshes, {tot['tris']} triangles -> {tot['nodes']} nodes: "
      f"triangle order, node bytes, roots and node counts all equal the oracle's; {time.time() - t0:.0f} s")
