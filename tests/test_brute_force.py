"""Independent evidence for the (unpinned) oracle: its BVH traversal against an all-triangles search.

The oracle's `orc_closest_hits` restates upstream's instance loop + IntersectBVH (kernel_main.cl:124-160,198-217).
`oracle/brute_force.c` runs the same instance loop but tests EVERY triangle of every instance with its own
Moeller-Trumbore code and never reads a BVH node. Wherever the tree cannot legitimately hide a triangle, both must name the
same triangle with bit-identical t, u, v:
  * the ray origin lies outside the root box of every instance's mesh (object space) -> hazard H1 (`tnear > 0` rejects
    boxes that contain the origin) cannot bite, because every node box lies inside its root box;
  * the ray did not hit the 250-pop cap (hazard H2) in any instance;
  * no box on the path from the mesh root to the winning triangle's leaf has zero thickness on some axis (upstream's
    strict `tnear < tfar` never enters such a box: exactly axis-aligned walls of real assets are invisible upstream) --
    computed here from the node ARRAY alone, top-down, without any traversal code;
  * the winner is not tied (another triangle at exactly the same t: the first one tested wins, and a list and a tree
    test in different orders).
The few remaining differences are rays that graze a box edge (the slab test rounds differently from the triangle test);
they are bounded and in each of them the traversal found a farther hit or none, never a nearer one.
"""
import ctypes as C
import os

import numpy as np
import pytest

from clraytracer_amd import _lib, driver, scenes
import oracle_lib
from util import bits, seeded_rays

BRUTE_DTYPE = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("tri", "<u4"), ("instance", "<i4"), ("ties", "<u4")])


def brute_force(a, mesh_start, mesh_count, o, d, nthreads):
    L = oracle_lib.lib()
    L.brute_force_hits.restype = None
    L.brute_force_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    out = np.zeros(len(o), BRUTE_DTYPE)
    tris, inst = np.ascontiguousarray(a["tris"]), np.ascontiguousarray(a["instances"])
    L.brute_force_hits(tris.ctypes.data, inst.ctypes.data, len(inst), mesh_start.ctypes.data, mesh_count.ctypes.data,
                       o.ctypes.data, d.ctypes.data, len(o), out.ctypes.data, nthreads)
    return out


def closest_hits_with_caps(orc, o, d):
    L = oracle_lib.lib()
    L.orc_closest_hits_ex.restype = None
    L.orc_closest_hits_ex.argtypes = [C.POINTER(oracle_lib.OrcScene), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(oracle_lib.OrcStats), C.c_int, C.c_void_p]
    out = np.zeros(len(o), _lib.RAYHIT_DTYPE)
    capped = np.zeros(len(o), np.uint8)
    st = oracle_lib.OrcStats()
    L.orc_closest_hits_ex(C.byref(orc.s), o.ctypes.data, d.ctypes.data, len(o), out.ctypes.data, C.byref(st), orc.nthreads, capped.ctypes.data)
    return out, capped.astype(bool)


def thinnest_box_on_path(nodes, roots, ntris):
    """Per triangle: the smallest relative thickness (shortest / longest edge) of any box on the way from its mesh's root
    (exclusive: the root box is never tested, kernel_main.cl:126,133-135) down to its leaf. 0 = an exactly flat box, which
    upstream's strict `tnear < tfar` can never enter. Level by level over the node ARRAY; no traversal code involved."""
    ext = (nodes["max"] - nodes["min"]).astype(np.float64)
    rel = np.maximum(ext.min(axis=1), 0.0) / np.maximum(ext.max(axis=1), 1e-300)
    thin = np.full(ntris, np.inf)
    frontier = np.asarray(roots, np.int64)
    cur = np.full(len(frontier), np.inf)                       # root boxes do not count
    while len(frontier):
        leaf = nodes["triCount"][frontier] > 0
        for n, b in zip(frontier[leaf], cur[leaf]):
            thin[nodes["leftFirst"][n]: nodes["leftFirst"][n] + nodes["triCount"][n]] = b
        inner, ib = frontier[~leaf], cur[~leaf]
        left = nodes["leftFirst"][inner].astype(np.int64)
        frontier = np.concatenate([left, left + 1])
        cur = np.concatenate([np.minimum(ib, rel[left]), np.minimum(ib, rel[left + 1])])
    return thin


@pytest.mark.parametrize("name,nrays", [("tiny", 65536), ("cornell-1k", 65536), ("sponza-class-250k", 65536), ("sponza-sibenik", 65536),
                                        ("nanosuit-demo", 65536), ("multi-1M", 8192)])
def test_traversal_equals_all_triangles_search(name, nrays, nthreads):
    # (multi-1M: 16 instances x 125k triangles = 2 M triangle tests per ray -> fewer rays, to keep the CPU suite short)
    sc = scenes.get(name)
    with driver.Session(64, 48, host_only=True) as s:
        s.load_scene(sc)
        a = s.arenas()
        iv, ip, pos = s.camera()
        nmesh = s.h.crth_num_meshes()
        info = np.zeros((nmesh, 4), np.uint32)
        for m in range(nmesh):
            s.h.crth_mesh_info(m, info[m].ctypes.data)
    mesh_count, mesh_start = np.ascontiguousarray(info[:, 0]), np.ascontiguousarray(info[:, 1])
    o, d = seeded_rays(a, pos, nrays, seed=2024)
    # keep rays whose origin is outside every instance's root box (object space, generous margin)
    inst = a["instances"]
    keep = np.ones(len(o), bool)
    for i in range(len(inst)):
        oo = np.concatenate([o.astype(np.float64), np.ones((len(o), 1))], 1) @ inst["inv"][i].astype(np.float64)
        root = a["nodes"][a["roots"][inst["meshIndex"][i]]]
        lo, hi = root["min"].astype(np.float64), root["max"].astype(np.float64)
        pad = 1e-3 * (hi - lo).max() + 1e-4
        keep &= ~((oo[:, :3] > lo - pad) & (oo[:, :3] < hi + pad)).all(axis=1)
    o, d = np.ascontiguousarray(o[keep]), np.ascontiguousarray(d[keep])
    assert len(o) > nrays // 3, "too few rays start outside the root boxes"

    orc = oracle_lib.Oracle(a, nthreads=nthreads)
    bvh, capped = closest_hits_with_caps(orc, o, d)
    bf = brute_force(a, mesh_start, mesh_count, o, d, nthreads)
    thin = thinnest_box_on_path(a["nodes"], a["roots"], len(a["tris"]))
    hidden = thin <= 0.0

    hit_bf = bf["instance"] >= 0
    strict = hit_bf & ~capped & (bf["ties"] == 0) & ~hidden[bf["tri"]] & np.isfinite(bvh["t"])
    same = (bvh["instance"] == bf["instance"]) & (bvh["tri"] == bf["tri"])
    for f in ("t", "u", "v"):
        same &= bits(bvh[f]) == bits(bf[f])
    wrong = strict & ~same
    # a ray for which the all-triangles search finds nothing must find nothing in the tree either (it tests a subset)
    assert not ((~hit_bf) & (bvh["instance"] >= 0) & np.isfinite(bvh["t"])).any()
    # differences: the tree found something farther (or nothing), never something nearer or a different record at equal t
    for k in np.nonzero(wrong)[0]:
        assert bvh["instance"][k] < 0 or bvh["instance"][k] != bf["instance"][k] or bvh["t"][k] > bf["t"][k], (k, bvh[k], bf[k])
    # ... and they are explained by an almost-flat box (relative thickness < 1e-5: float noise on the axis-aligned walls of
    # real assets, where the slab test's tnear and tfar round to the same value) on the way to the winning triangle
    explained = wrong & (thin[bf["tri"]] < 1e-5)
    unexplained = wrong & ~explained
    frac_strict = strict.sum() / max(1, hit_bf.sum())
    print(f"{name}: {len(o)} rays, {hit_bf.sum()} hit, {strict.sum()} strictly comparable ({100 * frac_strict:.1f} %), {same[strict].sum()} identical, "
          f"{explained.sum()} hidden by almost-flat boxes, {unexplained.sum()} grazing differences; excluded: {int((hit_bf & capped).sum())} capped, "
          f"{int((hit_bf & (bf['ties'] > 0)).sum())} tied, {int((hit_bf & hidden[bf['tri']]).sum())} behind exactly flat boxes")
    assert strict.sum() > 2000
    assert unexplained.sum() <= max(2, int(2e-3 * strict.sum()))
    assert explained.sum() <= 0.02 * strict.sum()
    if name in ("tiny", "cornell-1k", "sponza-class-250k", "multi-1M"):
        assert frac_strict > 0.9        # synthetic scenes have no axis-aligned flats (the cornell box is tilted for that reason)
