"""Headless driver: the reference's call order over the mirrored host API.

``Session`` replaces ``EngineMain.cpp:5-23`` + ``Engine.cpp:56-80``: initialise the renderer, load a
scene through ResourceManager (PrepareMeshes -> ImportTexture(skybox) -> ImportMesh... ->
PushMeshesToGPU -> PushTexturesToGPU -> Begin/Register/EndInstanceRegister) and render frames.
It only forwards to ``libcrt_host.so`` / ``libcrt_hip.so``; there is no Python compute path.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import CrtCounters, CrtError, CrtTraceArgs  # noqa: F401 (CrtError is re-exported: driver.CrtError)


class Session:
    def __init__(self, width, height, device=0, host_only=False, devices=None):
        """device: one HIP ordinal; devices: a list of ordinals -> several GPUs in this process (Renderer::InitializeDevices;
        the same GPU may be listed more than once to rehearse the multi-device path on a one-GPU box)."""
        self.h = _lib.host()
        self.hip = _lib.hip()
        self.width, self.height = int(width), int(height)
        self.host_only = bool(host_only)
        self.scene = None
        if host_only:
            ok = self.h.crth_initialize_host_only(self.width, self.height)
        elif devices is not None:
            ids = (C.c_int * len(devices))(*[int(d) for d in devices])
            ok = self.h.crth_initialize_devices(ids, len(devices), self.width, self.height)
        else:
            ok = self.h.crth_initialize(int(device), self.width, self.height)
        if not ok:
            rc = self.h.crth_last_error()
            raise CrtError(f"Renderer::Initialize failed ({rc}): {self.hip.crt_error_string(rc).decode()}")
        self.open = True

    # ---- lifecycle ----
    def close(self):
        if self.open:
            self.h.crth_terminate()
            self.open = False

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, what):
        rc = self.h.crth_last_error()
        if rc:
            raise CrtError(f"{what}: error {rc}: {self.hip.crt_error_string(rc).decode()}")

    # ---- scene ----
    def load_scene(self, scene, device_bvh_build=False):
        h = self.h
        h.crth_set_device_bvh_build(1 if device_bvh_build else 0)   # BuildBVH on the GPU (same bytes as the host build)
        h.crth_set_asset_root((scene.asset_root or "").encode())
        h.crth_prepare_meshes()
        tex = h.crth_import_texture(scene.skybox.encode())   # must be texture index 2 (Engine.cpp:60-61)
        self._check("ImportTexture(skybox)")
        assert tex == 2, tex
        handles = []
        for path in scene.meshes:
            handles.append(h.crth_import_mesh(path.encode()))
            self._check(f"ImportMesh({path})")
        h.crth_push_meshes()
        h.crth_push_textures()
        self._check("PushMeshesToGPU/PushTexturesToGPU")
        h.crth_begin_instances()
        for inst in scene.instances:
            p, keep = _lib.fptr(inst.matrix)
            h.crth_register_instance(handles[inst.mesh], int(inst.material), p)
        h.crth_end_instances()
        self._check("RegisterMeshInstance")
        self.set_camera(scene.camera_pos, scene.camera_front)
        self.scene = scene
        return handles

    def set_camera(self, pos, front):
        p, k1 = _lib.fptr(np.asarray(pos, np.float32))
        f, k2 = _lib.fptr(np.asarray(front, np.float32))
        self.h.crth_set_camera(p, f)

    def camera(self):
        iv = np.zeros(16, np.float32); ip = np.zeros(16, np.float32); pos = np.zeros(3, np.float32)
        self.h.crth_get_camera(iv.ctypes.data_as(C.POINTER(C.c_float)), ip.ctypes.data_as(C.POINTER(C.c_float)), pos.ctypes.data_as(C.POINTER(C.c_float)))
        return iv, ip, pos

    def resize(self, width, height):
        self.h.crth_resize(int(width), int(height))
        self._check("OnWindowResize")
        if width >= 16 and height >= 16:
            self.width, self.height = int(width), int(height)

    # ---- rendering through the mirrored Renderer ----
    def render(self, sun_angle=None, postprocess=False, shadows=False, pipelined=False, refraction=False, fxaa=False):
        self.h.crth_set_postprocess(1 if postprocess else 0)
        self.h.crth_set_shadows(1 if shadows else 0)          # extension, off upstream
        self.h.crth_set_refraction(1 if refraction else 0)    # extension, off upstream
        self.h.crth_set_fxaa(1 if fxaa else 0)                # extension: dead code upstream (kernel_main.cl:349)
        self.h.crth_set_pipelined(1 if pipelined else 0)      # frames in flight; output()/uploads wait
        frame = self.h.crth_render(float(self.scene.sun_angle if sun_angle is None else sun_angle))
        if frame == 0:
            self._check("Renderer::Render")
            raise CrtError("Renderer::Render failed")
        return frame

    def output(self):
        ptr = self.h.crth_map_output()
        if not ptr:
            self._check("Renderer::MapOutput")
            raise CrtError("Renderer::MapOutput returned null")
        return _lib.as_array(ptr, self.width * self.height * 4, np.float32).reshape(self.height, self.width, 4)

    # ---- direct C-ABI access (same device state the mirror drives) ----
    def trace_args(self, sun_angle=None):
        iv, ip, pos = self.camera()
        a = CrtTraceArgs()
        a.cameraPos[0], a.cameraPos[1], a.cameraPos[2] = float(pos[0]), float(pos[1]), float(pos[2])
        a.time = 0.0
        a.numMeshes = self.h.crth_num_instances()
        a.sunAngle = float(self.scene.sun_angle if sun_angle is None else sun_angle)
        return a, iv, ip

    def render_raw(self, flags=0, sun_angle=None, view=None):
        """crt_render with the session camera, or with explicit `view` = (invView[16], invProj[16], cameraPos[3]) (hazard H10: the
        boundary takes the matrices, so a test may hand it ones no camera produces)."""
        a, iv, ip = self.trace_args(sun_angle)
        if view is not None:
            iv, ip = np.ascontiguousarray(view[0], np.float32).reshape(16), np.ascontiguousarray(view[1], np.float32).reshape(16)
            a.cameraPos[0], a.cameraPos[1], a.cameraPos[2] = (float(x) for x in view[2])
        _lib.check(self.hip.crt_render(C.byref(a), iv.ctypes.data_as(C.POINTER(C.c_float)), ip.ctypes.data_as(C.POINTER(C.c_float)), int(flags)), "crt_render")

    def sync(self):
        _lib.check(self.hip.crt_sync(), "crt_sync")

    def last_kernel(self):
        """Name(s) of the Trace launch(es) of the most recently submitted frame (crt_debug_last_kernel)."""
        buf = C.create_string_buffer(128)
        _lib.check(self.hip.crt_debug_last_kernel(buf, len(buf)), "crt_debug_last_kernel")
        return buf.value.decode()

    def read_output(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        _lib.check(self.hip.crt_read_output(out.ctypes.data, out.size), "crt_read_output")
        return out

    def read_output_rgba8(self):
        """The frame as the bytes of upstream's RGBA8 render target (crt_read_output_rgba8)."""
        out = np.empty((self.height, self.width, 4), np.uint8)
        _lib.check(self.hip.crt_read_output_rgba8(out.ctypes.data, out.size), "crt_read_output_rgba8")
        return out

    def last_gather(self):
        """(bytes, bytes per pixel) the secondary devices sent to the first device for the last frame (crt_debug_last_gather)."""
        b, bpp = C.c_uint64(0), C.c_int(0)
        _lib.check(self.hip.crt_debug_last_gather(C.byref(b), C.byref(bpp)), "crt_debug_last_gather")
        return int(b.value), int(bpp.value)

    def read_rays(self):
        out = np.empty((self.height, self.width, 3), np.float32)
        _lib.check(self.hip.crt_read_rays(out.ctypes.data, out.size), "crt_read_rays")
        return out

    def counters(self):
        c = CrtCounters()
        _lib.check(self.hip.crt_get_counters(C.byref(c)), "crt_get_counters")
        return c.as_dict()

    def kernel_ms(self, which=2):
        return float(self.hip.crt_last_kernel_ms(int(which)))

    def query_hits(self, origins, dirs):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        out = np.zeros(len(o), _lib.RAYHIT_DTYPE)
        _lib.check(self.hip.crt_query_hits(o.ctypes.data, d.ctypes.data, len(o), self.h.crth_num_instances(), out.ctypes.data), "crt_query_hits")
        return out

    def set_row_bands(self, band_rows, rank, n_ranks):
        self.h.crth_set_row_bands(int(band_rows), int(rank), int(n_ranks))
        self._check("SetRowBands")

    def owned_rows(self):
        return int(self.hip.crt_owned_rows())

    # ---- host arenas (for the oracle in tests) ----
    def arenas(self):
        h = self.h
        return {
            "tris": _lib.as_array(h.crth_triangles(), h.crth_num_triangles(), _lib.TRI_DTYPE),
            "nodes": _lib.as_array(h.crth_nodes(), h.crth_num_nodes(), _lib.NODE_DTYPE),
            "roots": _lib.as_array(h.crth_roots(), h.crth_num_meshes(), np.uint32),
            "materials": _lib.as_array(h.crth_materials(), 256, _lib.MATERIAL_DTYPE),
            "textures": _lib.as_array(h.crth_textures(), 32, _lib.TEXTURE_DTYPE),
            "texels": _lib.as_array(h.crth_texels(), h.crth_texel_bytes(), np.uint8),
            "instances": _lib.as_array(h.crth_instances(), h.crth_num_instances(), _lib.INSTANCE_DTYPE),
            "num_materials": h.crth_num_materials(),
            "num_textures": h.crth_num_textures(),
        }

    def cpu_raycast(self, origins, dirs, nthreads=1, sse=False):
        """CPU_RayCast over many rays; sse=True: upstream's SSE instruction mix (approximate rcpps), the timing flavour."""
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        out = np.zeros(len(o), _lib.HITRECORD_DTYPE)
        (self.h.crth_cpu_raycast_sse if sse else self.h.crth_cpu_raycast)(o.ctypes.data, d.ctypes.data, len(o), out.ctypes.data, int(nthreads))
        return out
