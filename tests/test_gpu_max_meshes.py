"""Upstream's mesh limit (ResourceManager.cpp:40: 128 BVH roots): 128 different meshes, one instance each, through the host and the device BuildBVH, the
instance tree and the linear sphere loop, and the LDS-staged tree tops (whose 252-record table then holds ONE record per mesh) -- frames and counters
against the oracle; the 129th ImportMesh is refused with an error code instead of upstream's exit(0)."""
import numpy as np
import pytest

from clraytracer_amd import _lib, driver, scenes
import oracle_lib
from util import bits

pytestmark = pytest.mark.gpu
N = 128


def many_mesh_scene(tmp_path):
    rng = np.random.default_rng(11)
    paths, insts = [], []
    for k in range(N):
        kind = k % 3
        if kind == 0:
            mesh = scenes._box((-0.5, -0.5, -0.5), (0.5 + 0.01 * k, 0.5, 0.5))
        else:
            mesh = scenes._icosphere(kind, radius=0.5 + 0.002 * k)
        paths.append(scenes._write_mesh(str(tmp_path), f"m{k:03d}", mesh, [((0.3 + 0.005 * k, 0.8 - 0.004 * k, 0.5), None)]))
        m = scenes._trs(0.8 + 0.4 * rng.random(), rng.normal(size=3), rng.uniform(0, 6.28), (float((k % 16) - 7.5) * 2.2, float((k // 16) - 3.5) * 2.2, -float(k % 5)))
        insts.append(scenes.Instance(k, 0xFFFF, m))
    sky = str(tmp_path / "sky.ppm")
    scenes.write_ppm(sky, scenes._skybox(64, 32))
    return scenes.Scene("meshes128", str(tmp_path), sky, paths, insts, (0.0, 0.0, 22.0), scenes._normalize((0.0, 0.0, -1.0)))


@pytest.mark.parametrize("mode", ["host-build", "device-build", "linear-loop", "ldstop"])
def test_128_meshes(tmp_path, nthreads, monkeypatch, mode):
    sc = many_mesh_scene(tmp_path)
    monkeypatch.delenv("CRT_KERNEL", raising=False); monkeypatch.delenv("CRT_TLAS", raising=False)
    if mode == "linear-loop":
        monkeypatch.setenv("CRT_TLAS", "0")
    if mode == "ldstop":
        monkeypatch.setenv("CRT_KERNEL", "ldstop")
    W, H = 480, 272
    with driver.Session(W, H, device=0) as s:
        s.load_scene(sc, device_bvh_build=(mode == "device-build"))
        assert s.h.crth_num_meshes() == N
        a = s.arenas()
        assert len(a["roots"]) == N and len(a["instances"]) == N
        orc = oracle_lib.Oracle(a, nthreads=nthreads)
        iv, ip, pos = s.camera()
        want, st = orc.trace(orc.raygen(W, H, iv, ip), pos, sc.sun_angle)
        assert st["hits"] > 2000 and st["secondary"] > 2000          # every mesh is a few dozen pixels: 128 instance entries per ray either way
        s.render_raw(8)
        kernel = s.last_kernel()
        assert kernel == {"ldstop": "crt_trace_ldstop_kernel<1>", "linear-loop": "crt_trace_kernel<1,0,0,0,0>"}.get(mode, "crt_trace_kernel<1,0,0,1,0>"), kernel
        assert s.counters() == st
        assert ((bits(s.read_output()) != bits(want)).any(axis=2)).sum() <= 2
        plain = s.read_output().copy()
        for _ in range(4):
            s.render_raw(4)
        assert np.array_equal(bits(s.read_output()), bits(plain))
        if mode == "host-build":
            # one mesh too many: an error code, the session stays usable
            extra = scenes._write_mesh(str(tmp_path), "one_too_many", scenes._box((0, 0, 0), (1, 1, 1)), [((0.5, 0.5, 0.5), None)])
            assert s.h.crth_import_mesh(extra.encode()) == 0 and s.h.crth_last_error() == _lib.CRT_E_OUT_OF_RANGE
            s.h.crth_clear_error()
            s.render_raw(0)
            assert np.array_equal(bits(s.read_output()), bits(plain))
