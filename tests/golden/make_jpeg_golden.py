#!/usr/bin/env python3
"""Known answers for the JPEG texture decoder, produced by the REFERENCE's own decoder: every JPEG under the reference's
CLRayTracer/Assets is decoded with the reference's vendored stb_image.h (v2.27) compiled as it lies into
oracle/_ref/libstb_image_ref.so (oracle/Makefile `ref`), exactly as ResourceManager.cpp:193 calls it --
stbi_load(path, &w, &h, &channels, 3) -- and the SHA-256 of the RGB8 bytes is recorded in tests/golden/jpeg_stb.json.
Runs only where /root/reference is mounted (the build container):
    python tests/golden/make_jpeg_golden.py
The subset of those files that the asset scenes need is committed under assets/Assets (data fixtures), so
tests/test_jpeg.py can check the product's decoder against these hashes anywhere."""
import ctypes as C
import glob
import hashlib
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_ASSETS = "/root/reference/CLRayTracer/Assets"
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libstb_image_ref.so")


def main():
    L = C.CDLL(REF_SO)
    L.stbi_load.restype = C.POINTER(C.c_ubyte)
    L.stbi_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    L.stbi_image_free.argtypes = [C.c_void_p]
    out = {}
    files = sorted(f for f in glob.glob(os.path.join(REF_ASSETS, "**", "*"), recursive=True) if f.lower().endswith((".jpg", ".jpeg")))
    for f in files:
        w, h, c = C.c_int(), C.c_int(), C.c_int()
        p = L.stbi_load(f.encode(), C.byref(w), C.byref(h), C.byref(c), 3)
        assert p, f
        rel = os.path.relpath(f, REF_ASSETS)
        out[rel] = {"width": w.value, "height": h.value, "channels_in_file": c.value, "file_bytes": os.path.getsize(f),
                    "file_sha256": hashlib.sha256(open(f, "rb").read()).hexdigest(),
                    "rgb8_sha256": hashlib.sha256(C.string_at(p, w.value * h.value * 3)).hexdigest()}
        L.stbi_image_free(p)
    json.dump(out, open(os.path.join(HERE, "jpeg_stb.json"), "w"), indent=1, sort_keys=True)
    print(len(out), "files")


if __name__ == "__main__":
    main()
