"""The host side of an animated scene (round 6): crt_upload_instances recomputes only the records it replaced and REFITS the instance tree while the set
of cullable instances stays what the tree was built for (csrc/crt_instances.h rebuild_instance_master); a changed set, inner radii grown by a quarter or
256 refits bring a new median-split build. Whatever the tree's history, frames and work counters must equal the oracle's (which has no cull and no
tree: kernel_main.cl:198-217 loops over every instance) -- 401 instances (upstream's limit, Renderer.hpp:16) drifting, jumping, collapsing and coming back."""
import ctypes as C

import numpy as np
import pytest

from clraytracer_amd import _lib, driver, scenes
import oracle_lib
from util import bits

pytestmark = pytest.mark.gpu


def tlas_stats(s):
    b, r, n = C.c_uint64(0), C.c_uint32(0), C.c_uint32(0)
    _lib.check(s.hip.crt_debug_tlas_stats(C.byref(b), C.byref(r), C.byref(n)), "crt_debug_tlas_stats")
    return int(b.value), int(r.value), int(n.value)


def test_animated_instances_refit_the_tree_and_stay_exact(nthreads):
    tiny = scenes.get("tiny")
    W, H = 320, 200
    rng = np.random.default_rng(7)
    with driver.Session(W, H, device=0) as s:
        s.load_scene(tiny)
        s.h.crth_begin_instances()
        for k in range(len(tiny.instances), 401):
            m = scenes._trs(0.6 + 0.1 * (k % 5), (0.3, 1.0, 0.2), 0.37 * k, (float((k % 21) - 10) * 6.0, float((k // 21) - 9) * 6.0, -float(k % 7) * 2.0))
            pm, keep = _lib.fptr(m)
            s.h.crth_register_instance(k % 2, 0xFFFF, pm)
        s.h.crth_end_instances()
        s.set_camera((0.0, 0.0, 23.0 * 6.0), scenes._normalize((0.0, 0.0, -1.0)))
        arenas = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in s.arenas().items()}
        inst = arenas["instances"]
        assert len(inst) == 401
        iv, ip, pos = s.camera()

        def check(what):
            orc = oracle_lib.Oracle(arenas, nthreads=nthreads)
            want, st = orc.trace(orc.raygen(W, H, iv, ip), pos, tiny.sun_angle)
            s.render_raw(8)
            assert s.last_kernel() == "crt_trace_kernel<1,0,0,1,0>", s.last_kernel()          # the instance-tree instantiation
            assert s.counters() == st, what
            assert ((bits(s.read_output()) != bits(want)).any(axis=2)).sum() <= 2, what
        check("as loaded")
        builds0, _, nodes0 = tlas_stats(s)
        assert nodes0 == 2 * 401 - 1                                     # every instance cullable: a full binary tree over 401 leaves
        # (1) small drifts, whole table and dirty sub-ranges (upstream uploads [Min, Max) of what moved): refits only
        for step in range(6):
            lo, hi = (0, 401) if step % 2 == 0 else sorted(rng.integers(0, 401, size=2).tolist())
            hi = max(hi, lo + 1)
            inst["inv"][lo:hi, 3, :3] += rng.normal(scale=0.02, size=(hi - lo, 3)).astype(np.float32)
            assert s.hip.crt_upload_instances(inst[lo:hi].ctypes.data, lo, hi - lo) == 0
            check(f"drift {step}")
        builds1, refits1, nodes1 = tlas_stats(s)
        assert builds1 == builds0 and refits1 == 6 and nodes1 == nodes0
        # (2) forty instances jump across the scene: the refitted tree's radii grow past the threshold -> a new build, same frames
        far = rng.choice(401, size=40, replace=False)
        inst["inv"][far, 3, :3] += rng.uniform(-400, 400, size=(40, 3)).astype(np.float32)
        assert s.hip.crt_upload_instances(inst.ctypes.data, 0, 401) == 0
        check("jump")
        builds2, refits2, _ = tlas_stats(s)
        assert builds2 == builds1 + 1 and refits2 == 0
        # (3) three instances collapse to a singular matrix (their inverse is not invertible: never culled): the cullable set changes -> a build
        saved = inst["inv"][[5, 77, 300]].copy()
        inst["inv"][[5, 77, 300], 1, :3] = 0.0
        assert s.hip.crt_upload_instances(inst.ctypes.data, 0, 401) == 0
        with np.errstate(all="ignore"):
            check("collapsed")
        builds3, _, nodes3 = tlas_stats(s)
        assert builds3 == builds2 + 1 and nodes3 == 2 * 398 - 1
        # ... and come back through a sub-range upload
        inst["inv"][[5, 77, 300]] = saved
        assert s.hip.crt_upload_instances(inst[5:301].ctypes.data, 5, 296) == 0
        check("restored")
        builds4, _, nodes4 = tlas_stats(s)
        assert builds4 == builds3 + 1 and nodes4 == nodes0
        # (4) frames in flight between uploads see their own version of the tables
        for step in range(5):
            inst["inv"][:, 3, 1] += np.float32(0.01)
            assert s.hip.crt_upload_instances(inst.ctypes.data, 0, 401) == 0
            s.render_raw(4)
        check("after a pipelined burst")
