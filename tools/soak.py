#!/usr/bin/env python3
"""Soak check on the GPU box: ~3000 frames of multi-1M 1920x1080, mostly in flight with synchronous ones mixed in; the frame is hashed every ~50 frames against tests/golden/full_frames.json (no oracle involved)."""
import ctypes as C, hashlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clraytracer_amd import _lib, driver, scenes
g = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/full_frames.json")))["multi-1M"]
sc = scenes.get("multi-1M")
with driver.Session(1920, 1080, device=0) as s:
    s.load_scene(sc)
    hip = _lib.hip()
    a, iv, ip = s.trace_args(); fp = C.POINTER(C.c_float)
    args = (C.byref(a), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp))
    bad = 0; t0 = time.time()
    for round_ in range(60):
        for k in range(50 + round_ % 3):
            hip.crt_render(*args, 4 if (k % 7) else 0)      # mostly frames in flight, some synchronous
        h = hashlib.sha256(s.read_output().tobytes()).hexdigest()
        bad += h != g["frame_sha256"]
    print("soak: 60 rounds x ~51 frames,", bad, "mismatching hashes,", round(time.time() - t0, 1), "s")
    assert bad == 0
    # the per-pixel stages mixed in (epilogue, FXAA ping-pong, RGBA8 read-back): every combination must keep giving the frame
    # it gave the first time, and the plain frame must still be the golden one afterwards
    combos = (1, 64, 1 | 64, 512, 512 | 1, 512 | 1 | 64, 1 | 64 | 128, 512 | 128)
    first = {}
    for round_ in range(12):
        for f in combos:
            for k in range(5):
                hip.crt_render(*args, f | (4 if k else 0))
            hsh = hashlib.sha256(s.read_output().tobytes()).hexdigest()
            bad += first.setdefault(f, hsh) != hsh
    hip.crt_render(*args, 0)
    bad += hashlib.sha256(s.read_output().tobytes()).hexdigest() != g["frame_sha256"]
    print("soak: 12 rounds x", len(combos), "stage combinations x 5 frames,", bad, "mismatches,", round(time.time() - t0, 1), "s")
    assert bad == 0 and len(set(first.values())) >= 6
