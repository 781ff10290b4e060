#!/usr/bin/env python3
"""Differential fuzz of the conservative instance cull against the oracle (which has none), on the GPU box:
    python tools/fuzz_cull.py [first_seed] [count] [range_scale]
Every seed: 2..8 instances of the `tiny` scene's two meshes with log-uniform scales 1e-3..1e3, non-uniform factors up to 100 (Frobenius
condition numbers to ~300), random rotations, translations log-uniform up to 1e6 units; rays that graze the bounding spheres and the
corners of the boxes they are built around (tests/test_gpu_cull_bound.py's generator) from origins up to 1.0 x the scene's proven
range O (crt_get_cull_range), and up to 3.0 x O (those batches must run without the cull). Hit records and all work counters must equal
the oracle's. range_scale > 1 (needs CRT_DEBUG_HOOKS=1; sets CRT_DEBUG_CULL_RANGE_SCALE) stretches every O_i by that factor and only
REPORTS differences: how far beyond the derived worst-case bound the cull stays exact in practice."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
if scale != 1.0:
    os.environ["CRT_DEBUG_HOOKS"] = "1"; os.environ["CRT_DEBUG_CULL_RANGE_SCALE"] = repr(scale)
import test_gpu_cull_bound as cb  # noqa: E402
from clraytracer_amd import driver, scenes  # noqa: E402
import oracle_lib  # noqa: E402
from util import bits  # noqa: E402

t0 = time.time()
tot = {"scenes": 0, "rays": 0, "hits": 0, "culled": 0, "nocull_batches": 0, "never_culled_instances": 0, "instances": 0, "diff_scenes": 0, "diff_records": 0, "diff_counters": 0}
for seed in range(first, first + count):
    rng = np.random.RandomState(seed)
    insts = []
    for _ in range(int(rng.randint(2, 9))):
        s = 10.0 ** rng.uniform(-3, 3)
        f = 10.0 ** rng.uniform(-1, 1, 3) if rng.uniform() < 0.5 else np.ones(3)           # non-uniform: ratios up to 100
        t = rng.normal(size=3); t *= 10.0 ** rng.uniform(-2, 6) / np.linalg.norm(t)
        insts.append(scenes.Instance(int(rng.randint(0, 2)), 0xFFFF, cb._nonuniform(s * f[0], s * f[1], s * f[2], rng.normal(size=3), rng.uniform(0, 6.28), t)))
    sc = cb._scene(insts, f"cullfuzz-{seed}")
    with driver.Session(64, 48, device=0) as ss:
        ss.load_scene(sc)
        a = ss.arenas()
        orc = oracle_lib.Oracle(a, nthreads=min(16, os.cpu_count() or 1))
        lim, scene_lim, reach, _ = cb._cull_range(ss, len(a["instances"]))
        tot["instances"] += len(lim); tot["never_culled_instances"] += int((lim == 0).sum())
        if not (lim > 0).any():
            scene_lim = 100.0                                                                 # nothing cullable: any origins will do
        bad_scene = False
        for lo_f, hi_f in ((0.05, 1.0), (1.0, 3.0)):
            o, d = cb._grazing_rays(a, rng, hi_f * scene_lim, 1500)
            # _grazing_rays spreads origins over 0.3..1.0 of the bound: stretch the lower part down to lo_f
            gpu = ss.query_hits(o, d); cnt = ss.counters()
            ref, st = orc.closest_hits(o, d)
            nrec = int((gpu["instance"] != ref["instance"]).sum() + (gpu["tri"] != ref["tri"]).sum() + sum(int((bits(gpu[f]) != bits(ref[f])).sum()) for f in ("t", "u", "v")))
            tot["rays"] += len(o); tot["hits"] += int((ref["instance"] >= 0).sum()); tot["culled"] += cb._culled(ss)
            beyond = hi_f > 1.0 and (lim > 0).any() and np.linalg.norm(o.astype(np.float64), axis=1).max() > scene_lim
            if beyond:
                tot["nocull_batches"] += 1
                assert cb._culled(ss) == 0, (seed, "a batch with origins beyond the range ran with the cull")
            if nrec or cnt != st:
                bad_scene = True; tot["diff_records"] += nrec; tot["diff_counters"] += int(cnt != st)
                if scale == 1.0 or beyond:
                    print(f"seed {seed}: DIFFERENCE {'beyond the range (no cull!)' if beyond else 'inside the proven range'}: {nrec} record fields, counters equal: {cnt == st}; O_i = {lim}, scene limit {scene_lim}")
                    sys.exit(1)
        tot["diff_scenes"] += int(bad_scene)
    tot["scenes"] += 1
    if tot["scenes"] % 100 == 0:
        print(f"  ... {tot['scenes']} scenes, {tot['diff_scenes']} with differences, {time.time() - t0:.0f} s", flush=True)
what = "all hit records and counters equal the oracle's" if scale == 1.0 else f"every O_i stretched x{scale:g}: {tot['diff_scenes']} scenes with differences ({tot['diff_records']} record fields, {tot['diff_counters']} counter sets)"
print(f"cull fuzz seeds {first}..{first + count - 1}: {tot['scenes']} scenes, {tot['instances']} instances ({tot['never_culled_instances']} never culled: no origin range), "
      f"{tot['rays']} rays ({tot['hits']} hits), {tot['culled']} instance visits answered by the cull, {tot['nocull_batches']} batches beyond the range ran without it: {what}; {time.time() - t0:.0f} s")
