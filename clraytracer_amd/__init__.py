"""clraytracer_amd -- MI355X-native (gfx950) implementation of CLRayTracer's per-pixel ray-trace path.

The product is two in-tree shared libraries: ``csrc/libcrt_hip.so`` (hand-written HIP kernels behind the
C-ABI of ``include/crt_api.h``) and ``host/libcrt_host.so`` (C++ mirror of the reference's
Renderer / ResourceManager / AssetManager API). This Python package only binds them (ctypes), provides
the seeded synthetic scenes and a headless driver for tests and ``bench.py``. Nothing here computes
pixels: if the libraries are missing, importing ``_lib.hip()`` / ``_lib.host()`` raises ImportError.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib", "scenes", "driver"]
