#!/usr/bin/env python3
"""CPU preview of a scene through the oracle (no GPU): writes a PNG and prints the primary hit fraction and the work per ray.
    python tools/preview.py <scene> [--w 320 --h 180] [--pos x y z] [--front x y z] [-o /tmp/preview.png]
Test/tooling only (uses oracle/): for choosing cameras and eyeballing imported assets."""
import argparse, os, struct, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from clraytracer_amd import driver, scenes
import oracle_lib


def write_png(path, rgb8):
    h, w, _ = rgb8.shape
    raw = b"".join(b"\x00" + rgb8[y].tobytes() for y in range(h))
    def chunk(t, d):
        c = struct.pack(">I", len(d)) + t + d
        return c + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene"); ap.add_argument("--w", type=int, default=320); ap.add_argument("--h", type=int, default=180)
    ap.add_argument("--pos", type=float, nargs=3); ap.add_argument("--front", type=float, nargs=3)
    ap.add_argument("-o", default="/tmp/preview.png"); ap.add_argument("--post", action="store_true")
    a = ap.parse_args()
    sc = scenes.get(a.scene)
    with driver.Session(a.w, a.h, host_only=True) as s:
        s.load_scene(sc)
        if a.pos or a.front:
            s.set_camera(a.pos or sc.camera_pos, scenes._normalize(a.front or sc.camera_front))
        iv, ip, pos = s.camera()
        orc = oracle_lib.Oracle(s.arenas())
        img, st = orc.trace(orc.raygen(a.w, a.h, iv, ip), pos, sc.sun_angle)
        if a.post:
            img = orc.postprocess(img)
    rgb = np.clip(np.nan_to_num(img[..., :3]) * 255.0 + 0.5, 0, 255).astype(np.uint8)
    write_png(a.o, rgb[::-1])       # row 0 of the frame is the bottom of the view (kernel_main.cl:280-281, GL texture origin)
    print(f"{sc.name}: primary hit fraction {1 - (st['misses'] - (st['secondary'] - (st['hits'] - (st['primary'] - (st['misses'] - 0)))))/ max(1, st['primary']):.3f} (approx)")
    print({k: st[k] for k in ("rays", "primary", "secondary", "hits", "misses", "innerVisits", "triTests", "capHits", "maxStack")})
    print(f"secondary/primary = {st['secondary'] / st['primary']:.3f} (= fraction of primary rays that hit); visits/ray {st['innerVisits'] / st['rays']:.1f}, tri tests/ray {st['triTests'] / st['rays']:.2f} -> {a.o}")


if __name__ == "__main__":
    main()
