"""ctypes bindings for the in-tree shared libraries.

* ``libcrt_hip.so``  -- the C-ABI drop-in boundary (``include/crt_api.h``): HIP kernels for gfx950.
* ``libcrt_host.so`` -- C linkage over the C++ host mirror of the reference's
  Renderer / ResourceManager / AssetManager / CPU_RayCast (``include/crt_host.h``).

Loading fails loudly (``ImportError``) when a library has not been built: there is no Python or
CPU fallback for the ray-trace path. Build with ``make`` at the repo root or
``__graft_entry__.build()``.
"""
import ctypes as C
import os

import numpy as np

# One hardware queue per frame slot's stream (read by the HIP runtime at its first call in the process; never overrides the caller's setting):
# with the runtime's default of four queues two of the three slot streams occasionally share one and their frames serialise
# (profiles/r06_hw_queues.txt: 16.6 -> 12.3 Gray/s on nanosuit-demo, the same as GPU_MAX_HW_QUEUES=2).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
HIP_SO = os.path.join(_HERE, "csrc", "libcrt_hip.so")
HOST_SO = os.path.join(_HERE, "host", "libcrt_host.so")


class CrtTraceArgs(C.Structure):
    _fields_ = [("cameraPos", C.c_float * 3), ("time", C.c_float), ("numMeshes", C.c_uint32), ("sunAngle", C.c_float)]


class CrtFrameStats(C.Structure):
    _fields_ = [("frames", C.c_uint64), ("sumMs", C.c_double * 4), ("extentMs", C.c_double), ("firstFrameMs", C.c_double)]


class CrtCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "primary", "secondary", "hits", "misses", "traversals", "pops",
                                          "innerVisits", "triTests", "capHits", "stackOverflows", "maxStack",
                                          "shadowRays", "shadowHits")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


# numpy views of the reference-layout structs (include/crt_types.h)
TRI_DTYPE = np.dtype([("v0", "<f4", 3), ("cx", "<f4"), ("v1", "<f4", 3), ("cy", "<f4"), ("v2", "<f4", 3), ("cz", "<f4"),
                      ("uv", "<u2", 6), ("mat", "<u2"), ("n", "<u2", 9)])
NODE_DTYPE = np.dtype([("min", "<f4", 3), ("leftFirst", "<u4"), ("max", "<f4", 3), ("triCount", "<u4")])
MATERIAL_DTYPE = np.dtype([("color", "<u4"), ("specularColor", "<u4"), ("albedo", "<u2"), ("specular", "<u2"),
                           ("shininess", "<u2"), ("roughness", "<u2")])
TEXTURE_DTYPE = np.dtype([("width", "<i4"), ("height", "<i4"), ("offset", "<i4"), ("padd", "<i4")])
INSTANCE_DTYPE = np.dtype([("inv", "<f4", (4, 4)), ("meshIndex", "<u2"), ("materialStart", "<u2"), ("pad", "u1", 12)])
RAYHIT_DTYPE = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("tri", "<u4"), ("instance", "<i4")])
HITRECORD_DTYPE = np.dtype([("normal", "<f4", 3), ("uv", "<f4", 2), ("distance", "<f4"), ("color", "<u4"), ("index", "<u4")])
assert TRI_DTYPE.itemsize == 80 and NODE_DTYPE.itemsize == 32 and MATERIAL_DTYPE.itemsize == 16
assert TEXTURE_DTYPE.itemsize == 16 and INSTANCE_DTYPE.itemsize == 80 and RAYHIT_DTYPE.itemsize == 20
assert HITRECORD_DTYPE.itemsize == 32

_f = C.c_float
_fp = C.POINTER(C.c_float)
_vp = C.c_void_p
_sz = C.c_size_t

# name -> (restype, argtypes); kept in sync with include/crt_api.h + include/crt_debug.h (tests/test_abi.py checks it)
HIP_API = {
    "crt_init": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "crt_init_devices": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int]),
    "crt_init_gpus": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "crt_num_devices": (C.c_int, []),
    "crt_peer_access": (C.c_int, [C.c_int]),
    "crt_gather_path": (C.c_char_p, []),
    "crt_get_cull_range": (C.c_int, [_fp, C.c_int, _fp, _fp, C.POINTER(C.c_uint64)]),
    "crt_debug_staggered_frames": (C.c_int, [C.POINTER(C.c_uint64)]),
    "crt_debug_inject_failure": (C.c_int, [C.c_int]),
    "crt_debug_last_kernel": (C.c_int, [C.c_char_p, C.c_size_t]),
    "crt_debug_last_gather": (C.c_int, [C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "crt_debug_build_stats": (C.c_int, [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "crt_debug_tlas_stats": (C.c_int, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "crt_debug_measure_clock": (C.c_int, [C.c_int, C.POINTER(C.c_double)]),
    "crt_shutdown": (C.c_int, []),
    "crt_resize": (C.c_int, [C.c_int, C.c_int]),
    "crt_set_row_bands": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "crt_row_owner": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "crt_band_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "crt_upload_triangles": (C.c_int, [_vp, _sz, _sz]),
    "crt_upload_bvh_nodes": (C.c_int, [_vp, _sz, _sz]),
    "crt_upload_bvh_roots": (C.c_int, [_vp, _sz, _sz]),
    "crt_upload_materials": (C.c_int, [_vp, _sz, _sz]),
    "crt_upload_texture_table": (C.c_int, [_vp, _sz]),
    "crt_upload_texels": (C.c_int, [_vp, _sz, _sz]),
    "crt_upload_instances": (C.c_int, [_vp, _sz, _sz]),
    "crt_build_bvh": (C.c_int, [_sz, _vp, C.c_int, _sz, _sz, C.POINTER(C.c_uint32)]),
    "crt_download_triangles": (C.c_int, [_vp, _sz, _sz]),
    "crt_download_bvh_nodes": (C.c_int, [_vp, _sz, _sz]),
    "crt_download_bvh_roots": (C.c_int, [_vp, _sz, _sz]),
    "crt_render": (C.c_int, [C.POINTER(CrtTraceArgs), _fp, _fp, C.c_int]),
    "crt_sync": (C.c_int, []),
    "crt_query_hits": (C.c_int, [_vp, _vp, C.c_int, C.c_uint32, _vp]),
    "crt_read_output": (C.c_int, [_vp, _sz]),
    "crt_read_output_rows": (C.c_int, [_vp, C.c_int, C.c_int]),
    "crt_read_output_rgba8": (C.c_int, [_vp, _sz]),
    "crt_map_host_frame": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "crt_map_host_frame_back": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "crt_read_rays": (C.c_int, [_vp, _sz]),
    "crt_output_device_ptr": (_vp, []),
    "crt_owned_rows": (C.c_int, []),
    "crt_last_kernel_ms": (C.c_float, [C.c_int]),
    "crt_frame_time_stats": (C.c_int, [C.POINTER(CrtFrameStats), C.c_int]),
    "crt_get_counters": (C.c_int, [C.POINTER(CrtCounters)]),
    "crt_get_culled_visits": (C.c_int, [C.POINTER(C.c_uint64)]),
    "crt_debug_read_stamps": (C.c_int, [_vp, _sz, C.POINTER(C.c_size_t)]),
    "crt_debug_read_frame_times": (C.c_int, [_vp, _sz, C.POINTER(C.c_size_t)]),
    "crt_error_string": (C.c_char_p, [C.c_int]),
    "crt_device_name": (C.c_char_p, []),
}

HOST_API = {
    "crth_initialize": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "crth_initialize_devices": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int]),
    "crth_initialize_host_only": (C.c_int, [C.c_int, C.c_int]),
    "crth_terminate": (None, []),
    "crth_last_error": (C.c_int, []),
    "crth_clear_error": (None, []),
    "crth_prepare_meshes": (None, []),
    "crth_import_texture": (C.c_int, [C.c_char_p]),
    "crth_import_texture_rgb8": (C.c_int, [C.c_char_p, C.c_int, C.c_int, _vp]),
    "crth_import_mesh": (C.c_int, [C.c_char_p]),
    "crth_push_meshes": (None, []),
    "crth_set_device_bvh_build": (None, [C.c_int]),
    "crth_set_mesh_cache": (None, [C.c_int]),
    "crth_qlz_decompress": (C.c_size_t, [_vp, _sz, _vp, _sz]),
    "crth_qlz_store": (C.c_size_t, [_vp, _sz, _vp]),
    "crth_qlz_compress": (C.c_size_t, [_vp, _sz, _vp]),
    "crth_jpeg_decode": (C.c_size_t, [_vp, _sz, _vp, _sz, C.POINTER(C.c_int), C.POINTER(C.c_char_p)]),
    "crth_set_asset_root": (None, [C.c_char_p]),
    "crth_push_textures": (None, []),
    "crth_push_materials": (None, []),
    "crth_create_material": (C.c_int, [C.c_int]),
    "crth_edit_material": (None, [C.c_int, _vp]),
    "crth_begin_instances": (None, []),
    "crth_register_instance": (C.c_uint, [C.c_int, C.c_int, _fp]),
    "crth_end_instances": (None, []),
    "crth_clear_instances": (None, []),
    "crth_set_mesh_matrix": (None, [C.c_uint, _fp]),
    "crth_set_mesh_position": (None, [C.c_uint, _fp]),
    "crth_set_instance_material": (None, [C.c_uint, C.c_int]),
    "crth_set_camera": (None, [_fp, _fp]),
    "crth_get_camera": (None, [_fp, _fp, _fp]),
    "crth_resize": (None, [C.c_int, C.c_int]),
    "crth_set_postprocess": (None, [C.c_int]),
    "crth_set_shadows": (None, [C.c_int]),
    "crth_set_refraction": (None, [C.c_int]),
    "crth_set_fxaa": (None, [C.c_int]),
    "crth_set_unorm8": (None, [C.c_int]),
    "crth_map_output_rgba8": (_vp, []),
    "crth_set_pipelined": (None, [C.c_int]),
    "crth_set_row_bands": (None, [C.c_int, C.c_int, C.c_int]),
    "crth_render": (C.c_uint, [_f]),
    "crth_map_output": (_vp, []),
    "crth_last_frame_ms": (C.c_float, []),
    "crth_cpu_raycast": (None, [_vp, _vp, C.c_int, _vp, C.c_int]),
    "crth_cpu_raycast_sse": (None, [_vp, _vp, C.c_int, _vp, C.c_int]),
    "crth_triangles": (_vp, []), "crth_num_triangles": (_sz, []),
    "crth_nodes": (_vp, []), "crth_num_nodes": (_sz, []),
    "crth_roots": (_vp, []), "crth_num_meshes": (C.c_int, []),
    "crth_materials": (_vp, []), "crth_num_materials": (C.c_int, []),
    "crth_textures": (_vp, []), "crth_num_textures": (C.c_int, []),
    "crth_texels": (_vp, []), "crth_texel_bytes": (_sz, []),
    "crth_instances": (_vp, []), "crth_num_instances": (C.c_uint, []),
    "crth_mesh_info": (None, [C.c_int, _vp]),
    "crth_build_bvh": (C.c_uint32, [_vp, _vp, C.c_int, _vp, _vp]),
    "crth_float_to_half": (C.c_uint16, [_f]),
    "crth_half_to_float": (_f, [C.c_uint16]),
    "crth_inverse_transform": (None, [_fp, _fp]),
    "crth_inverse": (None, [_fp, _fp]),
    "crth_perspective_fov_rh": (None, [_f, _f, _f, _f, _f, _fp]),
    "crth_look_at_rh": (None, [_fp, _fp, _fp, _fp]),
    "crth_write_obj": (C.c_int, [C.c_char_p, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _vp, C.c_int, _vp, C.c_int]),
    "crth_shm_barrier_open": (_vp, [C.c_char_p, C.c_int, C.c_int]),
    "crth_shm_barrier_wait": (C.c_int, [_vp, C.c_int]),
    "crth_shm_barrier_close": (None, [_vp]),
}


def _bind(path, table):
    if not os.path.exists(path):
        raise ImportError(f"{path} has not been built (run `make` at {ROOT} or __graft_entry__.build()); "
                          "the ray-trace path has no fallback without it")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in table.items():
        fn = getattr(lib, name)  # AttributeError = missing export: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


_hip = None
_host = None


def hip():
    """libcrt_hip.so (the C-ABI). Loading does not touch the GPU; crt_init does."""
    global _hip
    if _hip is None:
        _hip = _bind(HIP_SO, HIP_API)
    return _hip


def host():
    """libcrt_host.so (C++ host mirror); pulls in libcrt_hip.so through its rpath."""
    global _host
    if _host is None:
        hip()
        _host = _bind(HOST_SO, HOST_API)
    return _host


# include/crt_api.h: CrtStatus
CRT_OK, CRT_E_NOT_INITIALIZED, CRT_E_BAD_ARGUMENT, CRT_E_OUT_OF_RANGE, CRT_E_NO_DEVICE, CRT_E_UNSUPPORTED = 0, -1, -2, -3, -4, -5


class CrtError(RuntimeError):
    pass


def check(rc, what="crt call"):
    if rc != 0:
        msg = hip().crt_error_string(rc)
        raise CrtError(f"{what} failed with {rc}: {msg.decode() if msg else '?'}")


def fptr(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a.ctypes.data_as(_fp), a


def as_array(ptr, count, dtype):
    """Copy `count` records of `dtype` from a raw pointer into a numpy array."""
    if not ptr or count == 0:
        return np.zeros(0, dtype=dtype)
    nbytes = int(count) * np.dtype(dtype).itemsize
    buf = (C.c_char * nbytes).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=int(count)).copy()
