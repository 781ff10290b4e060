#!/usr/bin/env python3
"""Experiment (GPU box): does the ORDER of the BVH node records matter to the trace kernel? The tree (structure, boxes, child order) is
the reference's; only node indices change: per mesh the top K levels are renumbered breadth-first (pair by pair), the rest keeps the
builder's depth-first order behind them. The frame must stay bit-identical (golden hash); frames in flight and synchronous rates are timed.
    python tools/layout_experiment.py [scene] [K ...]"""
import ctypes as C, hashlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clraytracer_amd import _lib, driver, scenes

name = sys.argv[1] if len(sys.argv) > 1 else "multi-1M"
Ks = [int(x) for x in sys.argv[2:]] or [0, 6, 10, 14, 99]
sc = scenes.get(name)


def reorder(nodes, roots, counts_nodes, K):
    """nodes: reference-layout array; every mesh m owns nodes[start_m : start_m + cnt_m] with its root first."""
    out = nodes.copy()
    for root, cnt in zip(roots, counts_nodes):
        root = int(root)
        if cnt < 3 or K == 0:
            continue
        lf = nodes["leftFirst"]; tc = nodes["triCount"]
        # breadth-first over inner nodes for K levels: order of their child PAIRS
        order = []                                  # old pair starts, new order
        frontier = [root]
        for _ in range(K):
            nxt = []
            for n in frontier:
                if tc[n] == 0:
                    l = int(lf[n]); order.append(l); nxt += [l, l + 1]
            frontier = nxt
            if not frontier:
                break
        top = set(order)
        # remaining pairs in their old (depth-first) order
        rest = [l for l in range(root + 1, root + cnt, 2) if l not in top]
        new_of = {}
        pos = root + 1
        for l in order + rest:
            new_of[l] = pos; pos += 2
        assert pos == root + cnt
        for l, nl in new_of.items():
            out[nl] = nodes[l]; out[nl + 1] = nodes[l + 1]
        # child links
        seg = out[root:root + cnt]
        inner = seg["triCount"] == 0
        old = seg["leftFirst"][inner]
        seg["leftFirst"][inner] = np.array([new_of[int(l)] for l in old], np.uint32)
        out[root:root + cnt] = seg
    return out


with driver.Session(1920, 1080, device=0) as s:
    s.load_scene(sc)
    hip = _lib.hip()
    a = s.arenas()
    nodes0, roots = a["nodes"].copy(), a["roots"].copy()
    nused = len(nodes0)
    starts = list(map(int, roots)) + [nused]
    counts_nodes = [starts[i + 1] - starts[i] for i in range(len(roots))]
    targs, iv, ip = s.trace_args(); fp = C.POINTER(C.c_float)
    args = (C.byref(targs), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp))
    ref_hash = None
    for K in Ks:
        nodes = reorder(nodes0, roots, counts_nodes, K)
        assert hip.crt_upload_bvh_nodes(nodes.ctypes.data, 0, nodes.nbytes) == 0
        assert hip.crt_upload_bvh_roots(roots.ctypes.data, 0, len(roots)) == 0
        for _ in range(15):
            assert hip.crt_render(*args, 0) == 0
        h = hashlib.sha256(s.read_output().tobytes()).hexdigest()
        ref_hash = ref_hash or h
        res = []
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(100):
                hip.crt_render(*args, 0)
            ts = (time.perf_counter() - t0) / 100
            for _ in range(10):
                hip.crt_render(*args, 4)
            hip.crt_sync()
            t0 = time.perf_counter()
            for _ in range(200):
                hip.crt_render(*args, 4)
            hip.crt_sync()
            tf = (time.perf_counter() - t0) / 200
            res.append((ts, tf))
        ts = min(r[0] for r in res); tf = min(r[1] for r in res)
        print(f"{name} K={K:3d}: synchronous {ts * 1e3:.4f} ms/frame, in flight {tf * 1e3:.4f} ms/frame, frame {'identical' if h == ref_hash else 'DIFFERENT'}", flush=True)
