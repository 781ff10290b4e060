"""BASELINE config 1 beside the HIP path, in the driver's GPU test record.

Config 1 is the reference's own CPU-runnable case: `CPU_RayCast` (CPURayTrace.cpp:186-249; host mirror
clraytracer_amd/host/CPURayTrace.cpp, scalar IEEE flavour) over one primary ray per pixel of cornell-1k at 640x480.
Here the same 307,200 rays also go through the HIP closest-hit query (`crt_query_hits`, the Trace kernel's traversal):
CPU_RayCast and Trace share IntersectBVH / IntersectAABB / IntersectTriangle (CPURayTrace.cpp:42-128 vs
kernel_main.cl:84-160), and with IEEE reciprocals pinned on both sides a primary ray must find the same triangle at the
same distance, bit for bit. (The scene's one instance has the identity transform, so the two different summation orders of
the object-space transform -- Matrix.hpp:658-667 pairwise vs MathAndSTL.cl:100-102 left to right -- give the same floats.)
The CPU side is also checked against the oracle's restatement, as tests/test_cpu_raycast.py does without a GPU.
"""
import time

import numpy as np
import pytest

from clraytracer_amd import driver, scenes
import oracle_lib
from util import bits

pytestmark = pytest.mark.gpu


def test_config1_cpu_raycast_equals_the_hip_traversal(nthreads):
    sc = scenes.get("cornell-1k")
    w, h = 640, 480
    with driver.Session(w, h, device=0) as s:
        s.load_scene(sc)
        a = s.arenas()
        assert len(a["instances"]) == 1 and np.array_equal(a["instances"]["inv"][0], np.eye(4, dtype=np.float32))
        iv, ip, pos = s.camera()
        orc = oracle_lib.Oracle(a, nthreads=nthreads)
        rays = orc.raygen(w, h, iv, ip).reshape(-1, 3)
        # the HIP RayGen writes the same buffer (row a1)
        s.render_raw(2)                                   # CRT_RENDER_WRITE_RAYS
        assert np.array_equal(bits(s.read_rays().reshape(-1, 3)), bits(rays))
        origins = np.tile(pos, (len(rays), 1)).astype(np.float32)
        t0 = time.perf_counter()
        cpu = s.cpu_raycast(origins, rays, nthreads=1)
        t1 = time.perf_counter()
        gpu = s.query_hits(origins, rays)
        ref = orc.cpu_raycast(origins, rays)
    assert len(cpu) == w * h == 307200
    hit_cpu = cpu["distance"] < 1e29                       # CPURayTrace.cpp:217: a miss keeps RayacastMissDistance = 1e30
    hit_gpu = gpu["instance"] >= 0                         # the kernel's miss test is distance > 99998 (kernel_main.cl:219)
    assert 30000 < hit_cpu.sum() < w * h
    assert np.array_equal(hit_cpu, hit_gpu)
    # HitRecord::index is the hit instance's mesh index (CPURayTrace.cpp:213,247)
    assert np.array_equal(cpu["index"][hit_cpu], a["instances"]["meshIndex"][gpu["instance"][hit_gpu]].astype(np.uint32))
    assert np.array_equal(bits(cpu["distance"][hit_cpu]), bits(gpu["t"][hit_gpu]))
    # and the whole record against the oracle's restatement of CPU_RayCast
    assert cpu.tobytes() == ref.tobytes()
    print(f"config 1: {w}x{h} CPU_RayCast on one host thread {(t1 - t0) * 1e3:.1f} ms = {w * h / (t1 - t0) / 1e6:.2f} Mrays/s; "
          f"{int(hit_cpu.sum())} hits, distance bit-equal to the HIP traversal on all of them")
