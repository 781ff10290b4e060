"""Independent evidence for SURVEY.md 8a rows a3-a6 (instance loop, IntersectBVH, IntersectAABB, IntersectTriangle): a numpy restatement
written from /root/reference/CLRayTracer/kernels/kernel_main.cl:84-160 and :198-217 alone -- every ray a small state machine (stack of 32
slots, pop counter, running Triout), all rays advanced together by array operations -- compared with the C oracle's closest-hit records
bit for bit, and with its work counters. tests/test_brute_force.py checks the oracle's *answers* against an all-triangles search where
the tree cannot legitimately hide anything; this file checks the *procedure* -- visit order, the 250-pop cap, the stack that is never
bounds-checked (pinned: slot index modulo 32), the box test that rejects boxes containing the origin (H1), the XOR in the hit predicate
and the arithmetic blend through which inf * 0 poisons the running t (H4/H5), t carried across instances in object-space units (H6) --
on camera rays, rays that start inside the geometry and axis-parallel rays (0 * inf in the slab test).
Shared with the oracle: only the pinned builtin semantics of oracle/crt_oracle.h (dot / cross order, native_recip = 1 / x, fmin / fmax)."""
import numpy as np
import pytest

from clraytracer_amd import driver, scenes
import oracle_lib

F = np.float32
INF_MINUS = F(99999.0)


def dot3(a, b):
    return (a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]) + a[:, 2] * b[:, 2]


def cross3(a, b):
    return np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2], a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], 1)


def matmul_xyz(m, v, w):                     # MathAndSTL.cl:100-102
    return ((m[0, :3] * v[:, 0:1] + m[1, :3] * v[:, 1:2]) + m[2, :3] * v[:, 2:3]) + m[3, :3] * F(w)


def intersect_aabb(o, inv, bmin, bmax, min_so_far):              # kernel_main.cl:108-117
    tmin = (bmin - o) * inv
    tmax = (bmax - o) * inv
    lo, hi = np.fmin(tmin, tmax), np.fmax(tmin, tmax)
    tnear = np.fmax(np.fmax(lo[:, 0], lo[:, 1]), lo[:, 2])
    tfar = np.fmin(np.fmin(hi[:, 0], hi[:, 1]), hi[:, 2])
    ok = (tnear < tfar) & (tnear > F(0.0)) & (tnear < min_so_far)
    return np.where(ok, tnear, F(1e30)).astype(np.float32)


def intersect_bvh(o, d, nodes_min, nodes_max, left_first, tri_count, tv0, tv1, tv2, root, out_t, stats):
    """kernel_main.cl:124-160 for all rays of one instance at once. out_t: the running t each ray enters with. Returns (intersection, t, u, v, tri)."""
    n = len(o)
    inv = F(1.0) / d                                                            # native_recip (pinned: 1 / x)
    stack = np.zeros((n, 32), np.int64); stack[:, 0] = root                     # int nodesToVisit[32] = { rootNode }
    sp = np.ones(n, np.int64)                                                   # currentNodeIndex = 1
    prot = np.zeros(n, np.int64)
    inter = np.zeros(n, np.int64)
    t = out_t.astype(np.float32).copy(); u = np.zeros(n, np.float32); v = np.zeros(n, np.float32); tri = np.zeros(n, np.int64)   # Triout (u, v uninitialised upstream: pinned 0)
    node = np.full(n, -1, np.int64)                                             # -1: at the top of the while loop
    alive = np.ones(n, bool)
    rows = np.arange(n)
    while alive.any():
        # ---- while (currentNodeIndex > 0 && protection++ < 250): node = nodesToVisit[--currentNodeIndex]
        top = alive & (node < 0)
        nonempty = top & (sp > 0)
        go = nonempty & (prot < 250)
        stats["capHits"] += int((nonempty & ~go).sum())
        prot[nonempty] += 1                                                     # protection++ is evaluated exactly when the first operand is true
        alive &= ~(top & ~go)
        sp[go] -= 1
        node[go] = stack[rows[go], sp[go] & 31]
        stats["pops"] += int(go.sum())
        # ---- traverse:
        cur = alive & (node >= 0)
        if not cur.any():
            continue
        idx = rows[cur]
        nd = node[idx]
        leaf = tri_count[nd] > 0
        # leaves: every triangle of the leaf (kernel_main.cl:137-138), then `continue`
        li = idx[leaf]
        if len(li):
            first = left_first[nd[leaf]].astype(np.int64); cnt = tri_count[nd[leaf]].astype(np.int64)
            for k in range(int(cnt.max())):
                m = cnt > k
                r = li[m]; i = first[m] + k
                stats["triTests"] += len(r)
                x, y, z = tv0[i], tv1[i], tv2[i]
                e1, e2 = y - x, z - x                                           # kernel_main.cl:86-87
                h = cross3(d[r], e2)
                a = dot3(e1, h)
                f = F(1.0) / a
                s = o[r] - x
                uu = f * dot3(s, h)
                q = cross3(s, e1)
                vv = f * dot3(d[r], q)
                tt = f * dot3(e2, q)
                bad = ((tt > F(0.0)) ^ (tt < t[r])).astype(np.int64) + (uu < F(0.0)) + (uu > F(1.0)) + (vv < F(0.0)) + ((uu + vv) > F(1.0))
                passed = (bad == 0).astype(np.int64); notp = 1 - passed
                pf, nf = passed.astype(np.float32), notp.astype(np.float32)
                u[r] = uu * pf + (nf * u[r])                                    # kernel_main.cl:101-104: arithmetic blends
                v[r] = vv * pf + (nf * v[r])
                t[r] = tt * pf + (nf * t[r])
                tri[r] = i * passed + (notp * tri[r])
                inter[r] |= passed
            node[li] = -1
        # inner nodes (kernel_main.cl:142-157)
        ii = idx[~leaf]
        if len(ii):
            stats["innerVisits"] += len(ii)
            l = left_first[nd[~leaf]].astype(np.int64); r_ = l + 1
            d1 = intersect_aabb(o[ii], inv[ii], nodes_min[l], nodes_max[l], t[ii])
            d2 = intersect_aabb(o[ii], inv[ii], nodes_min[r_], nodes_max[r_], t[ii])
            sw = d1 > d2
            d1s, d2s = np.where(sw, d2, d1), np.where(sw, d1, d2)
            near, far = np.where(sw, r_, l), np.where(sw, l, r_)
            miss = d1s == F(1e30)
            node[ii[miss]] = -1                                                 # `continue`
            hit = ~miss
            hi = ii[hit]
            node[hi] = near[hit]
            push = hit & (d2s != F(1e30))
            pi = ii[push]
            stats["stackOverflows"] += int((sp[pi] >= 32).sum())
            stack[pi, sp[pi] & 31] = far[push]                                  # nodesToVisit[currentNodeIndex++] = rightIndex (pinned: index modulo 32)
            sp[pi] += 1
            if len(pi):
                stats["maxStack"] = max(stats["maxStack"], int(sp[pi].max()))
    return inter, t, u, v, tri


def closest_hits_numpy(a, origins, dirs):
    """kernel_main.cl:189-217: the instance loop around IntersectBVH; returns records like the oracle's closest-hit query and its work counters"""
    n = len(origins)
    nodes, tris = a["nodes"], a["tris"]
    nodes_min, nodes_max = np.ascontiguousarray(nodes["min"], np.float32), np.ascontiguousarray(nodes["max"], np.float32)
    left_first, tri_count = nodes["leftFirst"].astype(np.int64), nodes["triCount"].astype(np.int64)
    tv0, tv1, tv2 = (np.ascontiguousarray(tris[k], np.float32) for k in ("v0", "v1", "v2"))
    best = np.full(n, INF_MINUS, np.float32)                                    # besthit.distance = Infinite
    hit_inst = np.zeros(n, np.int64); any_hit = np.zeros(n, bool)
    ht = np.zeros(n, np.float32); hu = np.zeros(n, np.float32); hv = np.zeros(n, np.float32); htri = np.zeros(n, np.int64)
    stats = {"traversals": 0, "pops": 0, "innerVisits": 0, "triTests": 0, "capHits": 0, "stackOverflows": 0, "maxStack": 0}
    o, d = np.ascontiguousarray(origins, np.float32), np.ascontiguousarray(dirs, np.float32)
    with np.errstate(all="ignore"):
        for i, inst in enumerate(a["instances"]):
            m = np.ascontiguousarray(inst["inv"], np.float32)
            mo, md = matmul_xyz(m, o, 1.0), matmul_xyz(m, d, 0.0)               # the direction is not renormalised (H6)
            root = int(a["roots"][inst["meshIndex"]])
            stats["traversals"] += n
            inter, t, u, v, tri = intersect_bvh(mo, md, nodes_min, nodes_max, left_first, tri_count, tv0, tv1, tv2, root, best, stats)
            got = inter != 0
            hit_inst[got] = i; any_hit |= got
            ht[got], hu[got], hv[got], htri[got] = t[got], u[got], v[got], tri[got]
            best[got] = t[got]                                                  # besthit.distance = triout.t
    return {"t": np.where(any_hit, ht, best), "u": np.where(any_hit, hu, F(0)), "v": np.where(any_hit, hv, F(0)), "tri": np.where(any_hit, htri, 0),
            "instance": np.where(any_hit, hit_inst, -1)}, stats


def ray_sets(a, iv, ip, pos, orc, rng):
    """camera rays (a coarse grid of the frame), rays from inside the scene's boxes (H1), axis-parallel rays from grid points"""
    w, h = 128, 72
    cam = orc.raygen(w, h, iv, ip).reshape(-1, 3)
    sets = [(np.tile(np.asarray(pos, np.float32), (len(cam), 1)), cam)]
    lo = a["nodes"]["min"][a["roots"]].min(axis=0); hi = a["nodes"]["max"][a["roots"]].max(axis=0)
    inside = (lo + (hi - lo) * rng.rand(1500, 3)).astype(np.float32)
    dirs = rng.randn(1500, 3).astype(np.float32); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True).astype(np.float32)
    sets.append((inside, dirs))
    axis = np.zeros((600, 3), np.float32); axis[np.arange(600), rng.randint(0, 3, 600)] = np.where(rng.rand(600) < 0.5, 1.0, -1.0)
    grid = (np.round((lo + (hi - lo) * rng.rand(600, 3)) * 4) / 4).astype(np.float32)
    sets.append((grid, axis))
    return sets


@pytest.mark.parametrize("name", ["tiny", "cornell-1k", "nanosuit-demo"])
def test_numpy_restatement_of_the_traversal_matches_the_oracle(name, nthreads):
    sc = scenes.get(name)
    with driver.Session(256, 144, host_only=True) as s:
        s.load_scene(sc)
        a = {k: (np.array(v) if isinstance(v, np.ndarray) else v) for k, v in s.arenas().items()}
        iv, ip, pos = s.camera()
    orc = oracle_lib.Oracle(a, nthreads=nthreads)
    rng = np.random.RandomState(sum(map(ord, name)))
    total_hits = nan_t = 0
    for origins, dirs in ray_sets(a, iv, ip, pos, orc, rng):
        ref, st = orc.closest_hits(origins, dirs)
        got, gs = closest_hits_numpy(a, origins, dirs)
        assert np.array_equal(got["instance"], ref["instance"]) and np.array_equal(got["tri"], ref["tri"].astype(np.int64))
        for k in ("t", "u", "v"):
            assert np.array_equal(np.ascontiguousarray(got[k], np.float32).view(np.uint32), np.ascontiguousarray(ref[k], np.float32).view(np.uint32)), k
        for k in ("traversals", "pops", "innerVisits", "triTests", "capHits", "stackOverflows", "maxStack"):
            assert gs[k] == st[k], (k, gs[k], st[k])
        total_hits += int((ref["instance"] >= 0).sum()); nan_t += int(np.isnan(ref["t"]).sum())
    print(f"{name}: {total_hits} hit records ({nan_t} with NaN t) and all work counters identical")
    assert total_hits > 500


@pytest.mark.parametrize("case", [1, 2, 3, 4, 5], ids=["seed2", "seed3", "seed4", "seed5", "seed6"])
def test_numpy_traversal_on_pathological_scenes(case, tmp_path, nthreads):
    """The fuzz suite's grid-snapped soups and singular / mirrored / 1e-3 / 1e3 instance matrices (tests/test_gpu_fuzz.py: CASES) with its
    axis-parallel and grid-aligned rays: where hit records carry NaN t, the 250-pop cap stops rays and ties in t are decided by test order."""
    import test_gpu_fuzz as fz
    seed, sizes, grid, kinds = fz.CASES[case]
    rng = np.random.default_rng(seed)
    sc = fz.build_scene(tmp_path, rng, seed, sizes, grid, kinds)
    with np.errstate(all="ignore"), driver.Session(64, 64, host_only=True) as s:
        s.load_scene(sc)
        a = {k: (np.array(v) if isinstance(v, np.ndarray) else v) for k, v in s.arenas().items()}
    orc = oracle_lib.Oracle(a, nthreads=nthreads)
    o, d = fz.special_rays(rng, 3072)
    ref, st = orc.closest_hits(o, d)
    got, gs = closest_hits_numpy(a, o, d)
    assert np.array_equal(got["instance"], ref["instance"]) and np.array_equal(got["tri"], ref["tri"].astype(np.int64))
    for k in ("t", "u", "v"):
        gb, rb = np.ascontiguousarray(got[k], np.float32).view(np.uint32), np.ascontiguousarray(ref[k], np.float32).view(np.uint32)
        # a NaN must be a NaN on both sides; its sign / payload bits depend on the host FPU's choice for inf * 0 and are not compared
        nan = np.isnan(ref[k])
        assert np.array_equal(np.isnan(got[k]), nan) and np.array_equal(gb[~nan], rb[~nan]), k
    for k in ("traversals", "pops", "innerVisits", "triTests", "capHits", "stackOverflows", "maxStack"):
        assert gs[k] == st[k], (k, gs[k], st[k])
    print(f"fuzz seed {seed}: {int((ref['instance'] >= 0).sum())} hit records, {int(np.isnan(ref['t']).sum())} with NaN t, {st['capHits']} rays stopped by the 250-pop cap, max stack {st['maxStack']}: identical")


@pytest.mark.parametrize("levels,reverse", [(34, False), (48, False), (48, True), (300, False)])
def test_numpy_traversal_on_hand_built_deep_trees(levels, reverse, nthreads):
    """Hazard H2 where it bites: caterpillar trees (tests/test_gpu_deep_stack.py) push up to 47 far children before the first pop, so the
    unchecked 32-entry stack overflows (pinned: the slot index wraps modulo 32 and later entries overwrite earlier ones), and a 300-level
    tree runs into the 250-pop cap. Both restatements must lose the same entries and stop at the same pop."""
    import test_gpu_deep_stack as ds
    from clraytracer_amd import _lib
    tris, nodes = ds.caterpillar(levels, reverse)
    inst = np.zeros(1, _lib.INSTANCE_DTYPE); inst["inv"][0] = np.eye(4, dtype=np.float32)
    a = {"tris": tris, "nodes": nodes, "roots": np.zeros(1, np.uint32), "instances": inst,
         "materials": np.zeros(256, _lib.MATERIAL_DTYPE), "textures": np.zeros(32, _lib.TEXTURE_DTYPE), "texels": np.zeros(12, np.uint8),
         "num_materials": 1, "num_textures": 3}
    orc = oracle_lib.Oracle(a, nthreads=nthreads)
    rng = np.random.RandomState(levels)
    n = 1500
    o = np.zeros((n, 3), np.float32); o[:, :2] = rng.uniform(-20, 20, (n, 2)); o[:, 2] = rng.uniform(-5, 5, n)
    d = np.zeros((n, 3), np.float32); d[:, :2] = rng.uniform(-0.15, 0.15, (n, 2)); d[:, 2] = 1.0
    ref, st = orc.closest_hits(o, d)
    got, gs = closest_hits_numpy(a, o, d)
    assert np.array_equal(got["instance"], ref["instance"]) and np.array_equal(got["tri"], ref["tri"].astype(np.int64))
    for k in ("t", "u", "v"):
        assert np.array_equal(np.ascontiguousarray(got[k], np.float32).view(np.uint32), np.ascontiguousarray(ref[k], np.float32).view(np.uint32)), k
    for k in ("traversals", "pops", "innerVisits", "triTests", "capHits", "stackOverflows", "maxStack"):
        assert gs[k] == st[k], (k, gs[k], st[k])
    if not reverse:                                             # (the reversed tree finds its nearest triangle first and prunes the rest)
        assert st["maxStack"] >= min(levels - 1, 33) or st["capHits"] > 0, st
    print(f"caterpillar {levels}{' reversed' if reverse else ''}: max stack {st['maxStack']}, {st['stackOverflows']} pushes past slot 31, {st['capHits']} rays stopped by the cap: identical")
