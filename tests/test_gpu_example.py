"""The C++ example (examples/headless_main.cpp: the reference's EngineMain/Engine_Start sequence over the mirrored
Renderer/ResourceManager API) renders the same frame as the Python driver, PostProcess included."""
import os
import subprocess

import numpy as np
import pytest

from clraytracer_amd import driver, scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_headless_example_matches_driver(tmp_path):
    exe = os.path.join(ROOT, "examples", "crt_headless")
    assert os.path.exists(exe), "run make"
    sc = scenes.get("cornell-1k")
    w, h = 320, 200
    out = str(tmp_path / "frame.ppm")
    cmd = [exe, sc.skybox, out, str(w), str(h), "3"] + ["%r" % float(v) for v in sc.camera_pos] + ["%r" % float(v) for v in sc.camera_front] + sc.meshes
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert res.returncode == 0, res.stdout
    assert "ms GPU time per frame" in res.stdout
    data = open(out, "rb").read()
    header = b"P6\n%d %d\n255\n" % (w, h)
    assert data.startswith(header)
    img = np.frombuffer(data[len(header):], np.uint8).reshape(h, w, 3)[::-1]
    with driver.Session(w, h, device=0) as s:
        s.load_scene(sc)
        s.render(postprocess=True)
        ref = s.output()[..., :3]
    ref8 = (np.clip(np.nan_to_num(ref, nan=0.0), 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
    assert np.abs(img.astype(int) - ref8.astype(int)).max() <= 1      # float -> text -> float camera arguments round-trip exactly
    assert (img != ref8).mean() < 1e-3


def test_cpp_example_renders_upstream_sponza_on_two_device_states(tmp_path):
    """The same program on upstream's own Sponza cache with its JPEG textures (CRT_ASSET_ROOT) and on two device states
    (CRT_DEVICES=0,0: Renderer::InitializeDevices): identical to the Python driver's single-device frame."""
    exe = os.path.join(ROOT, "examples", "crt_headless")
    assets = os.path.join(ROOT, "assets")
    w, h = 480, 272
    out = str(tmp_path / "sponza.ppm")
    pos, front = (-3.0, 19.5, 3.5), scenes._normalize((0.25, -1.0, -0.55))
    mesh = os.path.join(assets, "Assets", "sponza", "sponza.obj")
    sky = os.path.join(assets, "Assets", "earthmap.jpg")
    cmd = [exe, sky, out, str(w), str(h), "2"] + ["%r" % float(v) for v in pos] + ["%r" % float(v) for v in front] + [mesh]
    env = dict(os.environ, CRT_ASSET_ROOT=assets, CRT_DEVICES="0,0")
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120, env=env)
    assert res.returncode == 0, res.stdout
    data = open(out, "rb").read()
    header = b"P6\n%d %d\n255\n" % (w, h)
    img = np.frombuffer(data[len(header):], np.uint8).reshape(h, w, 3)[::-1]
    sc = scenes.Scene("sponza-only", assets, sky, [mesh], [scenes.Instance(0, 0xFFFF, np.eye(4, dtype=np.float32))], pos, front, asset_root=assets)
    with driver.Session(w, h, device=0) as s:
        s.load_scene(sc)
        s.render(postprocess=True)
        ref = s.output()[..., :3]
    ref8 = (np.clip(np.nan_to_num(ref, nan=0.0), 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
    assert np.abs(img.astype(int) - ref8.astype(int)).max() <= 1 and (img != ref8).mean() < 1e-3
    assert len(np.unique(img.reshape(-1, 3), axis=0)) > 500            # textured, not a flat default-white frame
