"""JPEG texture import (reference: ResourceManager.cpp:180-222, `stbi_load(path, &w, &h, &channels, 3)` from the vendored
stb_image v2.27). The product's own decoder (clraytracer_amd/host/JpegDecode.cpp) must produce the same RGB8 bytes.

What pins what:
* tests/golden/jpeg_stb.json holds, for every JPEG the reference ships, the SHA-256 of the bytes the REFERENCE's decoder
  produces (tests/golden/make_jpeg_golden.py runs the reference's stb_image.h, compiled as it lies into oracle/_ref/).
* The JPEGs the asset scenes use are committed under assets/ at the repository root (data) -> checked everywhere, CPU only.
* Where /root/reference is mounted, all 47 shipped JPEGs are checked, and where oracle/_ref/libstb_image_ref.so exists the
  two decoders are also compared live, byte for byte, including on damaged streams (same pixels whenever both decode).
Coverage of the shipped set: baseline and progressive, 4:4:4 and 4:2:0, grey-scale, restart intervals, Adobe APP14,
odd sizes (155x23 ... 8192x4096)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from clraytracer_amd import _lib, driver

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "jpeg_stb.json")))
FIXTURES = os.path.join(os.path.dirname(HERE), "assets", "Assets")
REF_ASSETS = "/root/reference/CLRayTracer/Assets"
REF_SO = os.path.join(_lib.ROOT, "oracle", "_ref", "libstb_image_ref.so")


def decode(data):
    """-> (rgb bytes, (w, h, channels_in_file, progressive)) or (None, error string)"""
    h = _lib.host()
    info = (C.c_int * 4)()
    err = C.c_char_p()
    need = h.crth_jpeg_decode(data, len(data), None, 0, info, C.byref(err))
    if need == 0:
        return None, (err.value or b"").decode()
    buf = C.create_string_buffer(need)
    assert h.crth_jpeg_decode(data, len(data), buf, need, info, C.byref(err)) == need
    return buf.raw, tuple(info)


def locate(rel):
    for base in (FIXTURES, REF_ASSETS):
        p = os.path.join(base, rel)
        if os.path.exists(p):
            return p
    return None


@pytest.mark.parametrize("rel", sorted(GOLD))
def test_decoder_matches_reference_stb_image_hashes(rel):
    path = locate(rel)
    if path is None:
        pytest.skip("not among the committed fixtures and the reference tree is not mounted")
    g = GOLD[rel]
    data = open(path, "rb").read()
    assert hashlib.sha256(data).hexdigest() == g["file_sha256"]
    rgb, info = decode(data)
    assert rgb is not None, info
    assert info[:3] == (g["width"], g["height"], g["channels_in_file"])
    assert len(rgb) == g["width"] * g["height"] * 3
    assert hashlib.sha256(rgb).hexdigest() == g["rgb8_sha256"]


def test_fixture_set_covers_the_format_variants():
    """The committed subset alone exercises baseline + progressive, 4:2:0 + 4:4:4, grey-scale and non-multiple-of-16 sizes."""
    seen = set()
    for rel, g in GOLD.items():
        p = os.path.join(FIXTURES, rel)
        if not os.path.exists(p):
            continue
        rgb, info = decode(open(p, "rb").read())
        seen.add(("progressive" if info[3] else "baseline", info[2]))
        if info[0] % 16 or info[1] % 16:
            seen.add("ragged")
    assert {("baseline", 3), ("progressive", 3), ("progressive", 1), "ragged"} <= seen


def _ref():
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref/libstb_image_ref.so not built (needs /root/reference at build time)")
    L = C.CDLL(REF_SO)
    L.stbi_load_from_memory.restype = C.POINTER(C.c_ubyte)
    L.stbi_load_from_memory.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    L.stbi_image_free.argtypes = [C.c_void_p]

    def ref_decode(data):
        w, h, c = C.c_int(), C.c_int(), C.c_int()
        p = L.stbi_load_from_memory(data, len(data), C.byref(w), C.byref(h), C.byref(c), 3)
        if not p:
            return None
        out = C.string_at(p, w.value * h.value * 3)
        L.stbi_image_free(p)
        return out, (w.value, h.value, c.value)
    return ref_decode


def test_live_against_reference_build_including_damaged_streams():
    """Both decoders on the same bytes: intact files must agree exactly; truncated and bit-flipped files must never crash
    the product's decoder, must be refused when the header is gone, and -- wherever both decoders still produce an image
    of the same size -- the damage is confined to the entropy-coded data they both interpret the same way up to the first
    error, so the leading rows agree."""
    ref_decode = _ref()
    rng = np.random.RandomState(11)
    names = [r for r in sorted(GOLD) if os.path.exists(os.path.join(FIXTURES, r))]
    assert len(names) >= 20
    checked = damaged_same = 0
    for rel in names:
        data = open(os.path.join(FIXTURES, rel), "rb").read()
        a, b = decode(data), ref_decode(data)
        assert a[0] == b[0] and a[1][:3] == b[1]
        checked += 1
        if len(data) > 400_000:
            continue
        for cut in (3, 20, len(data) // 3, len(data) - 2):
            rgb, info = decode(data[:cut])                       # must return, whatever it returns
            if cut <= 20:
                assert rgb is None
        for _ in range(3):
            d = bytearray(data)
            pos = int(rng.randint(len(d) // 2, len(d) - 2))       # inside the entropy-coded data of the last scan(s)
            d[pos] ^= 1 << int(rng.randint(0, 8))
            mine, theirs = decode(bytes(d)), ref_decode(bytes(d))
            if mine[0] is not None and theirs is not None and len(mine[0]) == len(theirs[0]):
                w = theirs[1][0]
                head = 8 * w * 3                                   # the first MCU row precedes any damage in the second half
                if mine[1][3] == 0:                                # baseline: rows decode in file order
                    assert mine[0][:head] == theirs[0][:head]
                damaged_same += mine[0] == theirs[0]
    assert checked >= 20
    print(f"{checked} fixtures identical to the reference build; {damaged_same} damaged variants still identical")


def test_rejects_non_jpeg_and_unsupported():
    assert decode(b"")[0] is None and decode(b"\x89PNG\r\n\x1a\n" + b"\0" * 64)[0] is None
    # 12-bit precision / arithmetic coding / lossless frames are refused like stb_image refuses them
    sof = lambda marker, prec: b"\xff\xd8" + b"\xff" + bytes([marker]) + b"\x00\x0b" + bytes([prec]) + b"\x00\x10\x00\x10\x01\x01\x11\x00" + b"\xff\xd9"
    assert decode(sof(0xC0, 12))[0] is None
    assert decode(sof(0xC9, 8))[0] is None and decode(sof(0xC3, 8))[0] is None


def test_import_texture_reads_jpeg_with_windows_path_semantics(tmp_path):
    """ResourceManager::ImportTexture on a JPEG: texel arena bytes == decoder output; the path is found under the asset
    root although the MTL spells it with a different case (upstream's sponza.mtl says 01_ST_KP.JPG, the file is 01_St_kp.JPG)."""
    root = os.path.join(os.path.dirname(HERE), "assets")
    g = GOLD["sponza/01_St_kp.JPG"]
    with driver.Session(64, 48, host_only=True) as s:
        h = s.h
        h.crth_set_asset_root(root.encode())
        h.crth_prepare_meshes()
        t = h.crth_import_texture(b"Assets/sponza/01_ST_KP.JPG")
        assert h.crth_last_error() == 0 and t == 2
        tex = _lib.as_array(h.crth_textures(), 32, _lib.TEXTURE_DTYPE)[2]
        assert (tex["width"], tex["height"], tex["offset"]) == (g["width"], g["height"], 2)
        texels = _lib.as_array(h.crth_texels(), h.crth_texel_bytes(), np.uint8)
        assert hashlib.sha256(texels[6:].tobytes()).hexdigest() == g["rgb8_sha256"]
        assert h.crth_import_texture(b"Assets/sponza/does_not_exist.jpg") == 0 and h.crth_last_error() != 0


def test_asset_scene_imports_every_texture_upstream_would():
    """sponza.clm through ImportMesh: 20 materials, one ImportTexture per map_Kd (19, duplicates included, as upstream:
    ResourceManager.cpp:262-266 does not de-duplicate) -> texture indices 2..20 after the two defaults."""
    root = os.path.join(os.path.dirname(HERE), "assets")
    with driver.Session(64, 48, host_only=True) as s:
        h = s.h
        h.crth_set_asset_root(root.encode())
        h.crth_prepare_meshes()
        h.crth_import_mesh(os.path.join(root, "Assets", "sponza", "sponza.obj").encode())
        assert h.crth_last_error() == 0
        assert h.crth_num_textures() == 21 and h.crth_num_materials() == 21
        mats = _lib.as_array(h.crth_materials(), h.crth_num_materials(), _lib.MATERIAL_DTYPE)
        assert sorted(int(x) for x in mats["albedo"][1:] if x) == list(range(2, 21))
        tex = _lib.as_array(h.crth_textures(), 32, _lib.TEXTURE_DTYPE)
        # KAMEN.JPG is imported three times (sp_podLZx, sp_zid, zid_vani): same size, different offsets
        sizes = [(int(t["width"]), int(t["height"])) for t in tex[2:21]]
        assert sizes.count((640, 477)) >= 4
        assert int(tex[20]["offset"]) * 3 + int(tex[20]["width"]) * int(tex[20]["height"]) * 3 == h.crth_texel_bytes()
