#!/usr/bin/env python3
"""Many-seed run of tests/test_gpu_fuzz.py's differential check (HIP path vs oracle on pathological geometry), on the GPU box:
    python tools/fuzz_many.py [first_seed] [count]
For every seed a random recipe (mesh sizes 1..8000 triangles, grid 2..33, 2..8 instances of random kinds incl. singular ones) is
rendered from two cameras and as four lattice frames (tests/test_gpu_fuzz.py::lattice_views: the axis-/grid-aligned special rays as whole
frames), and -- default kernel only -- queried with 4096 axis-/grid-aligned rays; hit records, counters and frames must equal the oracle's.
Under CRT_KERNEL=wavefront|refill|block the sweep uses only what the form renders (no shadow rays: a form refuses them with
CRT_E_UNSUPPORTED) and skips the query half (crt_query_hits has a kernel of its own and never runs a compaction form); every frame's
kernel name (crt_debug_last_kernel) is checked against the form. Prints one summary line; exits non-zero on the first difference."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathlib  # noqa: E402
import test_gpu_fuzz as fz  # noqa: E402
from clraytracer_amd import driver, scenes  # noqa: E402
import oracle_lib  # noqa: E402
from util import bits  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
kinds_all = ["plain", "mirrored", "tiny", "huge", "flat", "zero"]
form = os.environ.get("CRT_KERNEL") or "default"
prefix = fz.FORMS[form]
t0 = time.time()
tot = {"rays": 0, "nan_t": 0, "cap": 0, "hits": 0, "frames": 0, "frame_rays": 0, "skybox_flips": 0}
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    sizes = [int(rng.choice([1, 2, 3, 7, 40, 129, 700, 3000, 8000])) for _ in range(int(rng.integers(1, 4)))]
    grid = int(rng.choice([2, 3, 5, 9, 17, 33]))
    kinds = [str(rng.choice(kinds_all, p=[0.5, 0.1, 0.1, 0.1, 0.1, 0.1])) for _ in range(int(rng.integers(2, 9)))]
    with tempfile.TemporaryDirectory() as tmp, np.errstate(all="ignore"):
        sc = fz.build_scene(pathlib.Path(tmp), rng, seed, sizes, grid, kinds)
        with driver.Session(208, 120, device=0) as s:
            s.load_scene(sc)
            orc = oracle_lib.Oracle(s.arenas(), nthreads=min(16, os.cpu_count() or 1))
            o, d = fz.special_rays(rng, 4096)
            if form == "default":
                got = s.query_hits(o, d); cnt = s.counters()
                ref, st = orc.closest_hits(o, d)
                if got.tobytes() != ref.tobytes() or cnt != st:
                    print(f"seed {seed}: hit records / counters differ (sizes {sizes}, grid {grid}, kinds {kinds})"); sys.exit(1)
                tot["rays"] += len(o); tot["nan_t"] += int(np.isnan(ref["t"]).sum()); tot["cap"] += st["capHits"]; tot["hits"] += st["hits"]
            flags = 8 | (32 if (seed % 2 and form == "default") else 0)
            try:
                frames, rays, flips, cap = fz.check_frames(s, orc, sc, rng, prefix, ((0.5, 1.0, 9.0), (0.0, 0.0, 0.25)), flags=flags)
            except AssertionError as e:
                print(f"seed {seed}: {e} (sizes {sizes}, grid {grid}, kinds {kinds})"); sys.exit(1)
            tot["frames"] += frames; tot["frame_rays"] += rays; tot["skybox_flips"] += flips; tot["cap"] += cap
    if (seed - first + 1) % 500 == 0:
        print(f"  ... {seed - first + 1} scenes, all equal so far, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz seeds {first}..{first + count - 1}, kernel form {form} (every frame rendered by {prefix}...>): {count} scenes, "
      + (f"{tot['rays']} query rays ({tot['hits']} hits, {tot['nan_t']} with NaN t), " if form == "default" else "no query half (crt_query_hits never runs a form), ")
      + f"{tot['frames']} frames = 2 cameras + 4 lattice views per scene ({tot['frame_rays']} rays; "
      + ("odd seeds with shadow rays" if form == "default" else "no shadow rays: the form refuses them") + f"), {tot['cap']} rays stopped by the 250-pop cap: all hit records, "
      f"counters and frames equal the oracle's; {tot['skybox_flips']} skybox-texel flips tolerated; {time.time() - t0:.0f} s")
