#!/usr/bin/env python3
"""Round-3 experiment, kept for the record: -DCRT_EXP_TOPCOUNT no longer exists in the tree (tools/experiments/r03_ab_arms.patch brings the
counting arm back against round 3's crt_device.h); results: profiles/r03_top_count.txt.
Experiment (GPU box, library built with -DCRT_EXP_TOPCOUNT, see tools/ab_build.sh): how many of the trace kernel's vector-path
inner-node fetches go to the top K levels of a mesh's tree? The node array is renumbered so that every mesh's top-K child pairs come
first (tools/top_layout.py; same tree, same frame) and the counting instantiation reports, per frame: lane-level vector-path fetches,
those to a top record, and those in a wave-level step where EVERY active lane wanted a top record (the steps an LDS-resident copy of
the top records would take off the vector-memory path altogether).
    CRT_EXP_TOP_PAIRS=<T> is set per K by this script before the session starts (one process per K)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import numpy as np
    from clraytracer_amd import _lib, driver, scenes
    from top_layout import reorder_global
    name, K = sys.argv[2], int(sys.argv[3])
    sc = scenes.get(name)
    # T first (host only), then the real session with the environment set
    with driver.Session(64, 48, host_only=True) as s:
        s.load_scene(sc); a = s.arenas(); nodes, roots = a["nodes"].copy(), a["roots"].copy()
    n2, r2, T = reorder_global(nodes, roots, K)
    os.environ["CRT_EXP_TOP_PAIRS"] = str(T)
    with driver.Session(1920, 1080, device=0) as s:
        s.load_scene(sc)
        hip = _lib.hip()
        assert hip.crt_upload_bvh_nodes(n2.ctypes.data, 0, n2.nbytes) == 0
        assert hip.crt_upload_bvh_roots(r2.ctypes.data, 0, len(r2)) == 0
        s.render_raw(8)                                  # CRT_RENDER_COUNTERS
        c = s.counters()
        print(f"{name} K={K} T={T} records ({T * 64 / 1024:.0f} KiB): inner visits {c['innerVisits']}, vector-path lane fetches {c['shadowRays']}, "
              f"to a top record {c['stackOverflows']} ({100.0 * c['stackOverflows'] / max(1, c['shadowRays']):.1f} %), in all-top wave steps {c['shadowHits']} "
              f"({100.0 * c['shadowHits'] / max(1, c['shadowRays']):.1f} %)", flush=True)
    sys.exit(0)
name = sys.argv[1] if len(sys.argv) > 1 else "multi-1M"
for K in [int(x) for x in sys.argv[2:]] or [4, 5, 6, 8]:
    subprocess.run([sys.executable, os.path.abspath(__file__), "--one", name, str(K)], check=True)
