"""The `.clm` mesh cache (AssetManager.cpp:291-381) of the mirrored importer and its QuickLZ 1.5.0 level-1 stream.
The product's decoder is written from the format; where the reference tree is present its own quicklz.c -- compiled
as it lies into oracle/_ref/ (oracle/Makefile) -- provides real compressed streams and the decompression to compare with."""
import ctypes as C
import os
import shutil

import numpy as np
import pytest

from clraytracer_amd import _lib, driver, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libquicklz_ref.so")
REF_ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "Assets")   # the caches upstream ships, committed as data fixtures


def qlz_decompress(data, out_len):
    H = _lib.host()
    src = np.frombuffer(data, np.uint8)
    dst = np.zeros(out_len, np.uint8)
    n = H.crth_qlz_decompress(src.ctypes.data, len(src), dst.ctypes.data, len(dst))
    return n, dst


def test_stored_block_round_trip():
    H = _lib.host()
    rng = np.random.RandomState(1)
    for n in (1, 8, 300, 80000):
        raw = rng.randint(0, 256, n).astype(np.uint8)
        buf = np.zeros(n + 9, np.uint8)
        assert H.crth_qlz_store(raw.ctypes.data, n, buf.ctypes.data) == n + 9
        assert buf[0] == 0x46 and int.from_bytes(buf[1:5].tobytes(), "little") == n + 9 and int.from_bytes(buf[5:9].tobytes(), "little") == n
        got_n, got = qlz_decompress(buf.tobytes(), n)
        assert got_n == n and np.array_equal(got, raw)
    # malformed input is refused, never read or written out of bounds
    assert qlz_decompress(b"\x47\x10\x00\x00\x00\x40\x00\x00\x00" + b"\xff" * 7, 64)[0] == 0     # compressed flag, garbage body
    assert qlz_decompress(b"\x46\x09\x00\x00\x00\x40\x00\x00\x00", 64)[0] == 0                   # stored, body missing
    assert qlz_decompress(b"\x46", 64)[0] == 0


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libquicklz_ref.so is only built where the reference tree is mounted")
def test_decoder_against_the_reference_codec():
    R = C.CDLL(REF_SO)
    R.qlz_get_setting.restype = C.c_int
    assert R.qlz_get_setting(0) == 1 and R.qlz_get_setting(3) == 0       # level 1, no streaming buffer: what upstream ships
    R.qlz_compress.restype = C.c_size_t
    R.qlz_compress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    rng = np.random.RandomState(7)
    sc = scenes.get("tiny")
    with driver.Session(64, 48, host_only=True) as s:
        s.load_scene(sc)
        tris = s.arenas()["tris"].copy()
    samples = [tris.tobytes(),                                               # what a .clm holds: highly repetitive 80-B records
               bytes(200000), b"abc" * 70000, (b"0123456789" * 10 + b"x") * 3000,
               rng.randint(0, 4, 150000).astype(np.uint8).tobytes(),          # low entropy
               rng.randint(0, 256, 100000).astype(np.uint8).tobytes(),        # incompressible -> stored block
               np.repeat(rng.randint(0, 256, 3000).astype(np.uint8), rng.randint(1, 300, 3000)).tobytes(),
               b"a", b"ab" * 5, bytes(range(256)) * 3]
    for raw in samples:
        state = C.create_string_buffer(R.qlz_get_setting(1))
        comp = C.create_string_buffer(len(raw) + 400)
        n = R.qlz_compress(raw, comp, len(raw), state)
        assert n > 0
        got_n, got = qlz_decompress(comp.raw[:n], len(raw))
        assert got_n == len(raw) and got.tobytes() == raw, (len(raw), n)
        # truncated streams are refused
        if n > 20:
            assert qlz_decompress(comp.raw[:n // 2], len(raw))[0] in (0,)


@pytest.mark.parametrize("rel,min_tris", [("sphere.clm", 80), ("sponza/sponza.clm", 66447), ("sibenik/sibenik.clm", None), ("nanosuit/nanosuit.clm", None)])
def test_loads_upstream_mesh_caches(rel, min_tris, tmp_path):
    """The caches upstream ships (QuickLZ-compressed above 1000 triangles) load through the mirrored importer: they
    decode to exactly numTris * 80 bytes of finite triangles and a BVH can be built over them."""
    src = os.path.join(REF_ASSETS, rel)
    stem = os.path.splitext(os.path.basename(rel))[0]
    shutil.copy(src, tmp_path / (stem + ".clm"))                             # no .obj next to it: only the cache exists
    with driver.Session(64, 48, host_only=True) as s:
        h = s.h
        h.crth_prepare_meshes()
        import struct
        want_tris, want_mats = struct.unpack_from("<ii", open(src, "rb").read(12), 4)
        handle = h.crth_import_mesh(str(tmp_path / (stem + ".obj")).encode())
        # (no asset root is set here, so the JPEG textures the MTL text names are not found: those imports fail and fall
        #  back to the default texture -- the mesh itself must be there; tests/test_jpeg.py covers the textured import)
        info = np.zeros(4, np.uint32)
        h.crth_mesh_info(handle, info.ctypes.data)
        assert info[0] == want_tris and info[3] == want_mats and (min_tris is None or info[0] == min_tris)
        tris = _lib.as_array(h.crth_triangles(), h.crth_num_triangles(), _lib.TRI_DTYPE)
        assert len(tris) == info[0]
        for k in ("v0", "v1", "v2"):
            assert np.isfinite(tris[k]).all() and np.abs(tris[k]).max() < 1e6
        print(f"{rel}: {info[0]} triangles, {info[3]} materials")


@pytest.mark.parametrize("level", [2, 3])          # 320 triangles: raw in the file; 1280: behind upstream's 1000-triangle threshold -> QuickLZ block
def test_cache_written_on_import_and_preferred_afterwards(level, tmp_path):
    obj = scenes._write_mesh(str(tmp_path), "ico", scenes._icosphere(level, 1.5), [((0.8, 0.6, 0.4), None), ((0.2, 0.9, 0.4), None)])
    clm = obj[:-4] + ".clm"
    with driver.Session(64, 48, host_only=True) as s:
        s.h.crth_prepare_meshes()
        s.h.crth_import_mesh(obj.encode())
        assert s.h.crth_last_error() == 0 and os.path.exists(clm)
        first = _lib.as_array(s.h.crth_triangles(), s.h.crth_num_triangles(), _lib.TRI_DTYPE).copy()
        n_mat = s.h.crth_num_materials()
    assert len(first) == 20 * 4 ** level
    import struct
    blob = open(clm, "rb").read()
    version, nt, nm = struct.unpack_from("<Iii", blob, 0)
    msz, = struct.unpack_from("<I", blob, 12 + 24 * nm)
    body = len(blob) - (12 + 24 * nm + 4 + msz)
    assert version == 0 and nt == len(first) and nm == 2
    if nt < 1000:
        assert body == nt * 80                                              # raw triangles below upstream's threshold
    else:                                                                   # u64 size + the QuickLZ stream upstream's qlz_compress would write
        comp, = struct.unpack_from("<Q", blob, len(blob) - body)
        assert comp == body - 8 and comp < nt * 80 and blob[len(blob) - body + 8] == 0x47
        assert blob[len(blob) - body + 8:] == qlz_compress(first.tobytes())
    os.utime(clm, (os.path.getmtime(obj) + 5, os.path.getmtime(obj) + 5))
    os.rename(obj, obj + ".hidden")                                          # only the cache can serve the import now
    with driver.Session(64, 48, host_only=True) as s:
        s.h.crth_prepare_meshes()
        s.h.crth_import_mesh(obj.encode())
        assert s.h.crth_last_error() == 0
        again = _lib.as_array(s.h.crth_triangles(), s.h.crth_num_triangles(), _lib.TRI_DTYPE)
        assert again.tobytes() == first.tobytes() and s.h.crth_num_materials() == n_mat
    # a corrupt cache is ignored when the OBJ is there, an error when it is not
    os.rename(obj + ".hidden", obj)
    with open(clm, "r+b") as f:
        f.seek(40); f.write(b"\xff" * 64)
    os.utime(clm, (os.path.getmtime(obj) + 5, os.path.getmtime(obj) + 5))
    with driver.Session(64, 48, host_only=True) as s:
        s.h.crth_prepare_meshes()
        s.h.crth_import_mesh(obj.encode())
        assert s.h.crth_last_error() == 0
        assert _lib.as_array(s.h.crth_triangles(), s.h.crth_num_triangles(), _lib.TRI_DTYPE).tobytes() == first.tobytes()


FIXTURE_CLMS = ["sponza/sponza.clm", "sibenik/sibenik.clm", "nanosuit/nanosuit.clm"]
FIXTURE_ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "Assets")


def _clm_stream(blob):
    """-> (numTris, offset of the QuickLZ stream, its length) of a .clm with >= 1000 triangles (AssetManager.cpp:294-361)."""
    import struct
    version, nt, nm = struct.unpack_from("<Iii", blob, 0)
    assert version == 0 and nt >= 1000
    at = 12 + 24 * nm
    msz, = struct.unpack_from("<I", blob, at)
    at += 4 + msz
    comp, = struct.unpack_from("<Q", blob, at)
    return nt, at + 8, comp


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libquicklz_ref.so is only built where the reference tree is mounted")
@pytest.mark.parametrize("rel", FIXTURE_CLMS)
def test_shipped_caches_decode_like_the_reference_codec(rel):
    """The QuickLZ streams inside the caches upstream ships -- the only reference-held data of the mesh path -- decoded by
    the product's decoder and by the reference's own quicklz.c (compiled as it lies): identical bytes."""
    R = C.CDLL(REF_SO)
    R.qlz_get_setting.restype = C.c_int
    R.qlz_decompress.restype = C.c_size_t
    R.qlz_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    R.qlz_size_decompressed.restype = C.c_size_t
    R.qlz_size_decompressed.argtypes = [C.c_void_p]
    blob = open(os.path.join(FIXTURE_ASSETS, rel), "rb").read()
    nt, at, comp = _clm_stream(blob)
    stream = blob[at:at + comp]
    assert len(stream) == comp and R.qlz_size_decompressed(stream) == nt * 80
    state = C.create_string_buffer(R.qlz_get_setting(2))
    ref = C.create_string_buffer(nt * 80)
    assert R.qlz_decompress(stream, ref, state) == nt * 80
    n, got = qlz_decompress(stream, nt * 80)
    assert n == nt * 80 and got.tobytes() == ref.raw
    tris = np.frombuffer(ref.raw, _lib.TRI_DTYPE)
    assert np.isfinite(tris["v0"]).all() and np.isfinite(tris["v1"]).all() and np.isfinite(tris["v2"]).all()


def test_cache_with_a_lying_length_field_is_refused(tmp_path):
    """The 64-bit compressed-size field is untrusted: a value near 2^64 must not wrap the bounds check."""
    import struct
    blob = bytearray(open(os.path.join(FIXTURE_ASSETS, "nanosuit", "nanosuit.clm"), "rb").read())
    nt, at, comp = _clm_stream(bytes(blob))
    for lie in (2 ** 64 - 1, 2 ** 64 - (at - 4), len(blob), comp + 1):
        bad = bytearray(blob)
        struct.pack_into("<Q", bad, at - 8, lie)
        p = tmp_path / "liar.clm"
        p.write_bytes(bad)
        with driver.Session(64, 48, host_only=True) as s:
            s.h.crth_prepare_meshes()
            s.h.crth_import_mesh(str(tmp_path / "liar.obj").encode())
            assert s.h.crth_last_error() != 0 and s.h.crth_num_triangles() == 0


def qlz_compress(raw):
    H = _lib.host()
    src = np.frombuffer(raw, np.uint8)
    dst = np.zeros(len(raw) + 400, np.uint8)
    n = H.crth_qlz_compress(src.ctypes.data, len(src), dst.ctypes.data)
    return dst[:n].tobytes()


@pytest.mark.parametrize("rel", FIXTURE_CLMS)
def test_compressor_reproduces_the_streams_upstream_ships(rel):
    """Golden vectors held by the reference itself: the QuickLZ streams inside its shipped .clm caches were written by
    upstream's qlz_compress. Decode one, compress the triangles again with the product's compressor: the same bytes, so a
    cache written here is the file upstream would have written. Needs no reference tree (the caches are data fixtures)."""
    blob = open(os.path.join(FIXTURE_ASSETS, rel), "rb").read()
    nt, at, comp = _clm_stream(blob)
    stream = blob[at:at + comp]
    n, tris = qlz_decompress(stream, nt * 80)
    assert n == nt * 80
    again = qlz_compress(tris.tobytes())
    assert len(again) == comp and again == stream


def _compressor_samples():
    rng = np.random.RandomState(11)
    runs = np.repeat(rng.randint(0, 256, 3000).astype(np.uint8), rng.randint(1, 300, 3000)).tobytes()
    yield from [bytes(200000), b"abc" * 70000, (b"0123456789" * 10 + b"x") * 3000, runs,
                rng.randint(0, 4, 150000).astype(np.uint8).tobytes(),            # low entropy
                rng.randint(0, 256, 100000).astype(np.uint8).tobytes(),          # incompressible -> abandoned, stored block
                rng.randint(0, 256, 50000).astype(np.uint8).tobytes() + bytes(50000),   # gives up or not at the half-way check
                bytes(50000) + rng.randint(0, 256, 50000).astype(np.uint8).tobytes(),
                bytes(range(256)) * 3, b"aaaaaaaaaaab" * 40, b"a" * 11, b"a" * 215, b"a" * 216, b"ab" * 108, b"abcdefghijk"]
    for n in (11, 12, 17, 31, 64, 215, 216, 217, 1000, 4099, 65536):             # every size class incl. the short-header boundary
        yield rng.randint(0, 3, n).astype(np.uint8).tobytes()
        yield (rng.randint(0, 256, 7).astype(np.uint8).tobytes() * (n // 7 + 1))[:n]


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libquicklz_ref.so is only built where the reference tree is mounted")
def test_compressor_against_the_reference_codec():
    """Byte-for-byte against the reference's qlz_compress (quicklz.c compiled as it lies) on repetitive, low-entropy, run-length,
    incompressible and boundary-sized inputs, and on real triangle records."""
    R = C.CDLL(REF_SO)
    R.qlz_get_setting.restype = C.c_int
    R.qlz_compress.restype = C.c_size_t
    R.qlz_compress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    with driver.Session(64, 48, host_only=True) as s:
        s.load_scene(scenes.get("tiny"))
        tris = s.arenas()["tris"].copy()
    kinds = set()
    for raw in [tris.tobytes()] + list(_compressor_samples()):
        state = C.create_string_buffer(R.qlz_get_setting(1))
        comp = C.create_string_buffer(len(raw) + 400)
        n = R.qlz_compress(raw, comp, len(raw), state)
        mine = qlz_compress(raw)
        assert len(mine) == n and mine == comp.raw[:n], (len(raw), n, len(mine))
        kinds.add(mine[0])
        got_n, got = qlz_decompress(mine, len(raw))
        assert got_n == len(raw) and got.tobytes() == raw
    assert kinds == {0x47, 0x46, 0x45, 0x44} or kinds >= {0x47, 0x46, 0x45}     # long/short header, compressed/stored


def test_compressor_round_trips_without_the_reference():
    for raw in _compressor_samples():
        mine = qlz_compress(raw)
        got_n, got = qlz_decompress(mine, len(raw))
        assert got_n == len(raw) and got.tobytes() == raw, len(raw)
    assert qlz_compress(b"") == b""


@pytest.mark.parametrize("rel", ["sphere.clm"] + FIXTURE_CLMS)
def test_mtl_parser_reproduces_the_material_records_upstream_cached(rel, tmp_path, capfd):
    """More golden data the reference holds: every shipped cache stores the `ObjMaterial` records upstream's importer parsed
    out of the MTL file and the MTL text itself as the parser left it (a NUL written over the newline behind every name and
    texture path; AssetManager.cpp:118-160,300-304). Put the newlines back, import a one-triangle OBJ that names that MTL
    through the mirrored importer, and the cache written here must carry the same material records, the same text length
    and the same terminated text, byte for byte: names, packed colours, half-float shininess / opacity, path offsets."""
    import struct
    blob = open(os.path.join(FIXTURE_ASSETS, rel), "rb").read()
    version, nt, nm = struct.unpack_from("<Iii", blob, 0)
    msz, = struct.unpack_from("<I", blob, 12 + 24 * nm)
    text = bytearray(blob[16 + 24 * nm:16 + 24 * nm + msz])
    assert nm >= 1 and text.count(0) >= nm
    crlf = b"\r\n" in text                                                     # nanosuit's MTL was cached with CRLF line ends
    for i in range(len(text)):
        if text[i] == 0:
            text[i] = 13 if (crlf and i + 1 < len(text) and text[i + 1] == 10) else 10
    stem = os.path.splitext(os.path.basename(rel))[0]
    (tmp_path / (stem + ".mtl")).write_bytes(bytes(text))
    (tmp_path / (stem + ".obj")).write_text(f"mtllib {stem}.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvn 0 0 1\nf 1/1/1 2/1/1 3/1/1\n")
    with driver.Session(64, 48, host_only=True) as s:
        s.h.crth_prepare_meshes()
        s.h.crth_import_mesh(str(tmp_path / (stem + ".obj")).encode())           # textures are not there: those imports fall back, the mesh import stands
    capfd.readouterr()
    mine = (tmp_path / (stem + ".clm")).read_bytes()
    head = 16 + 24 * nm + msz
    assert struct.unpack_from("<Iii", mine, 0) == (0, 1, nm)
    assert mine[8:head] == blob[8:head]                                          # numMaterials, ObjMaterial[], mtl size, mtl text
    assert len(mine) == head + 80                                                # one raw triangle behind it


def test_decoder_survives_damaged_streams():
    """2000 seeded mutations (bit flips, truncations, spliced garbage, lying size fields) of valid streams: the decoder returns
    a length or 0 and never reads or writes outside its buffers (run under ASan + UBSan by tools/sanitize_host.sh)."""
    rng = np.random.RandomState(5)
    valid = [qlz_compress(raw) for raw in list(_compressor_samples())[:12]]
    sizes = [len(raw) for raw in list(_compressor_samples())[:12]]
    ok = 0
    for k in range(2000):
        i = int(rng.randint(len(valid)))
        b = bytearray(valid[i])
        kind = k % 5
        if kind == 0:
            for _ in range(int(rng.randint(1, 6))):
                b[int(rng.randint(len(b)))] ^= 1 << int(rng.randint(8))
        elif kind == 1:
            b = b[:int(rng.randint(0, len(b)))]
        elif kind == 2:
            at = int(rng.randint(len(b)))
            b[at:at + 8] = rng.randint(0, 256, 8).astype(np.uint8).tobytes()
        elif kind == 3 and len(b) > 9:
            b[1:9] = rng.randint(0, 256, 8).astype(np.uint8).tobytes()      # compressed / decompressed size fields
        else:
            b = bytearray(rng.randint(0, 256, int(rng.randint(1, 64))).astype(np.uint8).tobytes())
        cap = sizes[i] if k % 3 else int(rng.randint(0, sizes[i] + 1))        # also output buffers that are too small
        n, out = qlz_decompress(bytes(b), cap)
        assert n == 0 or n <= cap
        ok += n != 0
    assert ok < 2000
