"""The C-ABI libraries load on a box without a GPU and export every symbol the headers declare."""
import ctypes as C
import os
import re

import pytest

from clraytracer_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, text)))


def test_hip_library_exports_every_declared_symbol():
    api, dbg = declared("crt_api.h", "crt_"), declared("crt_debug.h", "crt_")
    assert len(api) >= 24 and not [n for n in api if n.startswith("crt_debug_")]      # the drop-in surface holds no diagnostics
    assert dbg and all(n.startswith("crt_debug_") for n in dbg)
    names = sorted(api + dbg)
    lib = C.CDLL(_lib.HIP_SO)
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.HIP_API) == names  # the ctypes table binds exactly the header


def test_host_library_exports_every_declared_symbol():
    names = declared("crt_host.h", "crth_")
    assert len(names) >= 50
    _lib.hip()
    lib = C.CDLL(_lib.HOST_SO)
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.HOST_API) == names


def test_struct_sizes_match_reference():
    assert _lib.TRI_DTYPE.itemsize == 80        # ResourceManager.hpp:69
    assert _lib.NODE_DTYPE.itemsize == 32
    assert _lib.MATERIAL_DTYPE.itemsize == 16
    assert _lib.TEXTURE_DTYPE.itemsize == 16
    assert _lib.INSTANCE_DTYPE.itemsize == 80
    assert C.sizeof(_lib.CrtTraceArgs) == 24    # Renderer.cpp:326-331


def test_calls_fail_loudly_without_init():
    hip = _lib.hip()
    # no crt_init has been made in this process: every entry point must refuse, never fall back
    assert hip.crt_sync() == -1
    assert hip.crt_upload_materials(None, 0, 1) == -1
    assert hip.crt_read_output(None, 0) == -1
    assert hip.crt_owned_rows() == 0
    assert b"not initialized" in hip.crt_error_string(-1)


def test_row_owner_is_a_partition():
    hip = _lib.hip()
    for n in (1, 2, 3, 4, 8):
        for band in (16, 32):
            owners = [hip.crt_row_owner(y, band, n) for y in range(2160)]
            assert set(owners) == set(range(n))
            assert all(owners[y] == (y // band) % n for y in range(2160))
    assert hip.crt_row_owner(0, 12, 2) < 0 and hip.crt_row_owner(-1, 16, 2) < 0 and hip.crt_row_owner(9, 8, 2) == 1


def test_render_in_host_only_session_is_an_error():
    from clraytracer_amd import driver, scenes
    with driver.Session(64, 48, host_only=True) as s:
        s.load_scene(scenes.get("tiny"))
        with pytest.raises(_lib.CrtError):
            s.render()
