#!/usr/bin/env python3
"""Renders every bench scene on the GPU the way upstream displays it -- Trace into the RGBA8 target, PostProcess -- and writes
640x360 PNGs (the bench cameras) to gpurun_out/frames/: what the numbers in profiles/ are frames OF. Run on the GPU box.
    python tools/render_gallery.py [--fxaa]"""
import os
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clraytracer_amd import driver, scenes  # noqa: E402


def write_png(path, rgb8):
    h, w, _ = rgb8.shape
    raw = b"".join(b"\x00" + rgb8[y].tobytes() for y in range(h))

    def chunk(t, d):
        c = struct.pack(">I", len(d)) + t + d
        return c + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


fxaa = "--fxaa" in sys.argv
out = os.path.join(ROOT, "gpurun_out", "frames")
os.makedirs(out, exist_ok=True)
W, H = 640, 360
for name in ("cornell-1k", "sponza-class-250k", "multi-1M", "multi-1M-dense", "sponza-sibenik", "nanosuit-demo"):
    with driver.Session(W, H, device=0) as s:
        s.load_scene(scenes.get(name))
        s.render_raw(1 | 64 | (512 if fxaa else 0))          # PostProcess | RGBA8 target (| FXAA)
        img = s.read_output()                                # float4, already through the RGBA8 store: k / 255 exactly
        rgb8 = np.rint(np.clip(np.nan_to_num(img[..., :3]), 0.0, 1.0) * 255.0).astype(np.uint8)
        path = os.path.join(out, f"{name}{'_fxaa' if fxaa else ''}.png")
        write_png(path, rgb8)
        print(name, rgb8.shape, os.path.getsize(path), "bytes")
