"""OBJ/MTL subset importer (AssetManager.cpp:90-289) and the mirrored ResourceManager bookkeeping."""
import os

import numpy as np
import pytest

from clraytracer_amd import _lib, driver, scenes

H = _lib.host()


def write(path, text):
    with open(path, "w") as f:
        f.write(text)


OBJ = """# comment line
mtllib quad.mtl
o quad
v -1.000000 0.000000 -1.500000
v 1.000000 0.000000 -1.500000
v 1.000000 2.250000 -1.500000
v -1.000000 2.250000 -1.500000
vt 0.000000 0.000000
vt 1.000000 0.000000
vt 1.000000 0.750000
vt 0.000000 0.750000
vn 0.000000 0.000000 1.000000
vn 0.577350 0.577350 0.577350
s off
usemtl red
f 1/1/1 2/2/1 3/3/2
usemtl blue
f 1/1/1 3/3/2 4/4/1
"""
MTL = """# materials
newmtl red
Ns 75.000000
d 0.250000
Kd 1.000000 0.500000 0.000000
Ks 0.100000 0.200000 0.300000
newmtl blue
Kd 0.000000 0.000000 1.000000
map_Kd tex.ppm
"""


@pytest.fixture()
def quad(tmp_path):
    write(tmp_path / "quad.obj", OBJ)
    write(tmp_path / "quad.mtl", MTL)
    scenes.write_ppm(str(tmp_path / "tex.ppm"), np.arange(4 * 2 * 3, dtype=np.uint8).reshape(2, 4, 3))
    scenes.write_ppm(str(tmp_path / "sky.ppm"), np.full((2, 2, 3), 200, np.uint8))
    return tmp_path


def test_import_quad(quad):
    with driver.Session(64, 48, host_only=True) as s:
        H.crth_prepare_meshes()
        assert H.crth_import_texture(str(quad / "sky.ppm").encode()) == 2
        assert H.crth_import_mesh(str(quad / "quad.obj").encode()) == 0
        assert H.crth_last_error() == 0
        a = s.arenas()
        t = a["tris"]
        assert len(t) == 2
        assert t["v0"][0].tolist() == [-1.0, 0.0, -1.5] and t["v2"][0].tolist() == [1.0, 2.25, -1.5]
        assert t["v2"][1].tolist() == [-1.0, 2.25, -1.5]
        # uv: v is flipped to 1 - v (AssetManager.cpp:271), stored as half
        uv = t["uv"].view(np.float16).astype(np.float32).reshape(2, 3, 2)
        assert uv[0].tolist() == [[0.0, 1.0], [1.0, 1.0], [1.0, 0.25]]
        n = t["n"].view(np.float16).astype(np.float32).reshape(2, 3, 3)
        assert n[0][0].tolist() == [0.0, 0.0, 1.0]
        assert abs(n[0][2][0] - 0.57735) < 1e-3
        assert t["mat"].tolist() == [0, 1]                     # slots in MTL order through the name hash
        # materials: slot 0 is PrepareMeshes' default, the mesh's own start at 1 (ResourceManager.cpp:246)
        info = np.zeros(4, np.uint32); H.crth_mesh_info(0, info.ctypes.data)
        assert info.tolist() == [2, 0, 1, 2]
        m = a["materials"]
        assert m["color"][1] == (255 | (127 << 8) | (0 << 16))  # PackColorRGBU32 truncates (Math.hpp:237)
        assert m["specularColor"][1] == (25 | (51 << 8) | (76 << 16))
        assert m["color"][2] == (255 << 16) and m["specularColor"][2] == 0xFFFFFFFF
        assert m["shininess"][1] == H.crth_float_to_half(75.0 / 50.0) and m["roughness"][1] == H.crth_float_to_half(0.25)
        assert m["shininess"][2] == H.crth_float_to_half(2.2) and m["roughness"][2] == H.crth_float_to_half(0.6)
        assert m["albedo"][1] == 0 and m["albedo"][2] == 3       # tex.ppm imported as texture 3
        tex = a["textures"]
        assert (tex["width"][3], tex["height"][3], tex["offset"][3]) == (4, 2, 2 + 4)   # after white, black, 2x2 sky
        assert tex["offset"][2] == 2 and tex["offset"][1] == 3   # upstream's black-texture byte offset quirk
        px = a["texels"]
        assert px[:6].tolist() == [255, 255, 255, 0, 0, 0]
        assert px[6 + 12:].tolist() == list(range(24))


def test_import_errors_do_not_exit(quad, capfd):
    with driver.Session(64, 48, host_only=True):
        assert H.crth_import_mesh(str(quad / "missing.obj").encode()) == 0
        assert H.crth_last_error() != 0
    write(quad / "bad.obj", "v 0 0 0\nvt 0 0\nvn 0 0 1\nf 1/1/1 2/1/1 1/1/1\n")
    with driver.Session(64, 48, host_only=True):
        H.crth_import_mesh(str(quad / "bad.obj").encode())
        assert H.crth_last_error() != 0                          # face index out of range
    write(quad / "empty.obj", "# nothing\n")
    with driver.Session(64, 48, host_only=True):
        H.crth_import_mesh(str(quad / "empty.obj").encode())
        assert H.crth_last_error() != 0
    with driver.Session(64, 48, host_only=True):
        H.crth_import_texture(str(quad / "quad.obj").encode())  # not a PPM
        assert H.crth_last_error() != 0


def test_writer_importer_round_trip(tmp_path):
    rng = np.random.RandomState(4)
    nv, nt = 300, 500
    pos = rng.uniform(-64, 64, (nv, 3)).astype(np.float32)
    uv = rng.uniform(0, 4, (nv, 2)).astype(np.float32)
    nrm = rng.normal(size=(nv, 3)); nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    tri = rng.randint(0, nv, (nt, 3)).astype(np.int32)
    mat = np.sort(rng.randint(0, 3, nt)).astype(np.int32)
    m = scenes.Mesh(pos, uv, nrm, tri, mat)
    scenes._write_mesh(str(tmp_path), "rt", m, [((0.5, 0.5, 0.5), None)] * 3)
    scenes.write_ppm(str(tmp_path / "sky.ppm"), np.zeros((2, 2, 3), np.uint8))
    with driver.Session(64, 48, host_only=True) as s:
        H.crth_prepare_meshes()
        H.crth_import_texture(str(tmp_path / "sky.ppm").encode())
        H.crth_import_mesh(str(tmp_path / "rt.obj").encode())
        assert H.crth_last_error() == 0
        t = s.arenas()["tris"]
    # what the text round trip must give: %.6f text parsed in double, narrowed to float
    q = lambda a: np.array([float("%.6f" % x) for x in a.ravel()], np.float64).astype(np.float32).reshape(a.shape)
    assert np.array_equal(t["v0"], q(pos)[tri[:, 0]]) and np.array_equal(t["v1"], q(pos)[tri[:, 1]]) and np.array_equal(t["v2"], q(pos)[tri[:, 2]])
    assert np.array_equal(t["mat"], mat.astype(np.uint16))
    f2h = np.vectorize(lambda v: H.crth_float_to_half(float(v)), otypes=[np.uint16])
    quv = q(uv)
    exp_uv = np.stack([f2h(quv[tri[:, k], 0]) if c == 0 else f2h(np.float32(1.0) - quv[tri[:, k], 1]) for k in range(3) for c in range(2)], 1)
    assert np.array_equal(t["uv"], exp_uv)
    exp_n = np.stack([f2h(q(nrm)[tri[:, k], c]) for k in range(3) for c in range(3)], 1)
    assert np.array_equal(t["n"], exp_n)


def test_scene_load_bookkeeping():
    sc = scenes.get("tiny")
    with driver.Session(64, 48, host_only=True) as s:
        s.load_scene(sc)
        a = s.arenas()
        assert len(a["roots"]) == 2 and len(a["instances"]) == 3
        assert a["num_textures"] == 2 + 1 + 2                   # white, black, sky, one map per mesh
        assert a["textures"]["offset"][2] == 2                  # skybox is texture 2 at texel offset 2 (hazard H9)
        inst = a["instances"]
        assert inst["meshIndex"].tolist() == [0, 1, 0]
        # DefaultMaterial resolves to the mesh's own materialStart (Renderer.cpp:231-233)
        assert inst["materialStart"].tolist() == [1, 3, 1]
        for k, i in enumerate(sc.instances):
            assert np.allclose(i.matrix.astype(np.float64) @ inst["inv"][k].astype(np.float64), np.eye(4), atol=1e-5)


def test_triangle_arena_capacity_is_an_error_not_an_overrun(tmp_path):
    """MAX_TRIANGLES (1.2 M, ResourceManager.cpp:34) bounds the host arena: importing past it must fail with a code while
    parsing (upstream would write past the arena first and exit later), from the OBJ and from a `.clm` cache alike."""
    import numpy as np
    from clraytracer_amd import _lib, driver, scenes
    big = scenes._icosphere(7, 1.0)                       # 327,680 triangles
    obj = scenes._write_mesh(str(tmp_path), "big", big, [((0.5, 0.5, 0.5), None)])
    with driver.Session(64, 48, host_only=True) as s:
        h = s.h
        h.crth_prepare_meshes()
        for k in range(3):                                # 983,040 triangles in
            h.crth_import_mesh(obj.encode())
            assert h.crth_last_error() == 0, k
        assert h.crth_num_triangles() == 3 * 327680
        h.crth_import_mesh(obj.encode())                  # the fourth would end at 1,310,720 > 1,200,000 (served from the .clm cache)
        assert h.crth_last_error() == -3 or h.crth_last_error() == -2
        assert h.crth_num_triangles() == 3 * 327680 and h.crth_num_meshes() == 3
    os.remove(obj[:-4] + ".clm")
    with driver.Session(64, 48, host_only=True) as s:
        h = s.h
        h.crth_set_mesh_cache(0)                          # the same through the OBJ parser
        h.crth_prepare_meshes()
        for k in range(3):
            h.crth_import_mesh(obj.encode())
        h.crth_import_mesh(obj.encode())
        assert h.crth_last_error() in (-2, -3) and h.crth_num_meshes() == 3
        h.crth_set_mesh_cache(1)


def test_importer_survives_damaged_files(quad, tmp_path, capfd):
    """600 seeded mutations of a valid OBJ / MTL / .clm triple (byte flips, truncations, inserted garbage, huge indices and
    counts): every import returns a mesh or an error code -- nothing exits, hangs, or touches memory outside its buffers
    (run under ASan + UBSan by tools/sanitize_host.sh). Upstream's parser trusts its input; this one is a file-facing
    boundary."""
    import os
    import shutil
    rng = np.random.RandomState(9)
    obj = (quad / "quad.obj").read_bytes(); mtl = (quad / "quad.mtl").read_bytes()
    # a valid cache of the same mesh, written by the importer itself
    with driver.Session(64, 48, host_only=True):
        H.crth_prepare_meshes()
        assert H.crth_import_mesh(str(quad / "quad.obj").encode()) == 0
    clm = (quad / "quad.clm").read_bytes()

    def mutate(b):
        b = bytearray(b)
        kind = int(rng.randint(6))
        if kind == 0 and len(b):
            for _ in range(int(rng.randint(1, 8))):
                b[int(rng.randint(len(b)))] = int(rng.randint(256))
        elif kind == 1:
            b = b[:int(rng.randint(0, len(b) + 1))]
        elif kind == 2:
            at = int(rng.randint(len(b) + 1)); b[at:at] = rng.randint(0, 256, int(rng.randint(1, 40))).astype(np.uint8).tobytes()
        elif kind == 3:
            at = int(rng.randint(len(b) + 1)); b[at:at] = rng.choice([b"f 99999999/1/1 2/2/2 3/3/3\n", b"f -5/-5/-5 1/1/1 2/2/2\n", b"v 1e39 nan inf\n", b"usemtl nosuch\n",
                                                                     b"newmtl " + b"x" * 300 + b"\n", b"map_Kd " + b"y" * 600 + b"\n", b"f 1 2 3\n", b"f 1//1 2//2 3//3 4//4 5//5\n"])
        elif kind == 4:
            b = b.replace(b"\n", b"\r\n")
        else:
            b = b.replace(b" ", b"\t", int(rng.randint(1, 5)))
        return bytes(b)

    outcomes = {"ok": 0, "error": 0}
    for k in range(600):
        d = tmp_path / f"m{k % 7}"
        shutil.rmtree(d, ignore_errors=True); d.mkdir()
        which = k % 3
        (d / "quad.obj").write_bytes(mutate(obj) if which == 0 else obj)
        (d / "quad.mtl").write_bytes(mutate(mtl) if which == 1 else mtl)
        shutil.copy(quad / "tex.ppm", d / "tex.ppm")
        if which == 2:                                          # a damaged cache that looks fresher than the OBJ
            (d / "quad.clm").write_bytes(mutate(clm))
            t = os.path.getmtime(d / "quad.obj") + 5
            os.utime(d / "quad.clm", (t, t))
            if k % 2:
                os.remove(d / "quad.obj")                       # ... with or without the OBJ to fall back to
        with driver.Session(64, 48, host_only=True):
            H.crth_prepare_meshes()
            h = H.crth_import_mesh(str(d / "quad.obj").encode())
            err = H.crth_last_error()
            assert (err == 0) or (err < 0)
            outcomes["ok" if err == 0 else "error"] += 1
            if err == 0:
                info = np.zeros(4, np.uint32); H.crth_mesh_info(h, info.ctypes.data)
                assert info[0] <= 100000                        # whatever was parsed is bounded by the file
    capfd.readouterr()
    assert outcomes["ok"] > 50 and outcomes["error"] > 50, outcomes
