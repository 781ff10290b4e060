"""Differential fuzz of the HIP path against the oracle on geometry a modelling tool never exports but a parity claim has to
survive: triangle soups on a coarse grid (coplanar faces, shared edges, exact ties in t), duplicated and zero-area triangles,
one- and two-triangle meshes, mirrored / tiny / huge / singular instance matrices (the inverse of a singular matrix is Inf/NaN
-- hazard H4's NaN path), rays along the axes (0 * Inf in the slab test) and origins exactly on grid planes.
Everything must be bit-identical: closest-hit records, work counters, and the rendered frame (kernel_main.cl:84-275)."""
import numpy as np
import pytest

from clraytracer_amd import driver, scenes
import oracle_lib
from util import bits

pytestmark = pytest.mark.gpu


def soup(rng, ntri, grid, degenerate):
    """ntri triangles with vertices on an integer grid of `grid` steps: many coplanar / coincident / edge-sharing faces."""
    v = rng.integers(0, grid, size=(ntri, 3, 3)).astype(np.float32) * np.float32(4.0 / grid) - np.float32(2.0)
    if degenerate and ntri >= 4:
        v[1] = v[0]                                        # exact duplicate: a tie in t, first tested wins
        v[2, 2] = v[2, 1]                                  # zero area
        v[3, 1] = v[3, 0]; v[3, 2] = v[3, 0]               # a point
    pos = v.reshape(-1, 3)
    n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    n = np.where(ln > 0, n / np.maximum(ln, 1e-30), np.array([[0, 1, 0]], np.float32)).astype(np.float32)
    uv = rng.random((ntri * 3, 2)).astype(np.float32)
    return scenes.Mesh(pos, uv, np.repeat(n, 3, axis=0), np.arange(ntri * 3, dtype=np.int32).reshape(ntri, 3), np.zeros(ntri, np.int32))


def matrices(rng, kind):
    m = scenes._trs(1.0, rng.normal(size=3), rng.uniform(0, 6.28), rng.uniform(-3, 3, size=3))
    if kind == "mirrored":
        m[0, :3] *= -1.0
    elif kind == "tiny":
        m[:3, :3] *= 1e-3
    elif kind == "huge":
        m[:3, :3] *= 1e3; m[3, :3] *= 50.0
    elif kind == "flat":                                   # singular: the mesh squashed into a plane, inverse has Inf/NaN
        m[1, :3] = 0.0
    elif kind == "zero":
        m[:3, :3] = 0.0
    return m.astype(np.float32)


def special_rays(rng, n):
    """Axis-parallel and grid-aligned rays next to ordinary ones."""
    o = rng.uniform(-6, 6, size=(n, 3))
    d = rng.normal(size=(n, 3))
    k = n // 4
    ax = rng.integers(0, 3, size=k)
    d[:k] = 0.0; d[np.arange(k), ax] = rng.choice([-1.0, 1.0], size=k)          # along one axis: two zero components
    d[k:2 * k, rng.integers(0, 3)] = 0.0                                          # in an axis plane: one zero component
    o[:2 * k] = np.round(o[:2 * k] * 2.0) / 2.0                                   # origins exactly on grid planes
    o[2 * k:3 * k] = np.round(o[2 * k:3 * k] * 4.0) / 4.0
    d[2 * k:3 * k] = -o[2 * k:3 * k] + rng.integers(-2, 3, size=(k, 3)) * 0.5     # aimed exactly at grid points (vertices, edges)
    ln = np.linalg.norm(d, axis=1, keepdims=True)
    d = np.where(ln > 0, d / np.maximum(ln, 1e-30), np.array([[0.0, 0.0, -1.0]]))
    return o.astype(np.float32), d.astype(np.float32)


def lattice_views(rng, count=4):
    """Explicit (invView, invProj, cameraPos) triples no camera produces (the boundary takes matrices, hazard H10): the special
    rays of special_rays() as whole FRAMES, so that every Trace kernel form -- not only crt_query_hits' own kernel -- meets
    them. RayGen (kernel_main.cl:277-287) computes dir = normalize((invView . (invProj . (c, 1, 1) / w)).xyz) with c = pixel / size * 2 - 1;
    with invProj = identity and an invView whose rows are signed, scaled unit vectors, a power-of-two frame gives rays from an origin on a
    grid plane through grid points of the soups' lattice ("fan"), rays confined to an axis plane (one zero component, "plane") and frames
    whose every ray runs along one axis (two zero components: 0 * Inf in the slab test, "axis")."""
    views = []
    ip = np.eye(4, dtype=np.float32)
    for k in range(count):
        kind = ("fan", "plane", "axis", "fan")[k % 4]
        perm = rng.permutation(3); sign = rng.choice([-1.0, 1.0], size=3)
        sx, sy = (float(rng.choice([0.5, 1.0, 2.0])), float(rng.choice([0.5, 1.0, 2.0])))
        if kind == "plane":
            sx = 0.0
        if kind == "axis":
            sx = sy = 0.0
        iv = np.zeros((4, 4), np.float32)
        iv[0, perm[0]] = sign[0] * sx; iv[1, perm[1]] = sign[1] * sy; iv[2, perm[2]] = sign[2]
        pos = np.round(rng.uniform(-6, 6, size=3) * 2.0) / 2.0             # on grid planes
        if kind != "fan":
            pos[perm[2]] = -sign[2] * 7.0                                   # outside the soups, looking at them along the axis
        views.append((kind, iv.reshape(16).copy(), ip.reshape(16).copy(), pos.astype(np.float32)))
    return views


FORMS = {"default": "crt_trace_kernel<", "wavefront": "crt_primary_kernel<", "refill": "crt_trace_refill_kernel<", "block": "crt_trace_block_kernel<",
         "ldstop": "crt_trace_ldstop_kernel<"}


@pytest.fixture(params=list(FORMS))
def form(request, monkeypatch):
    """The Trace kernel structure of the session (CRT_KERNEL, read by crt_init): the default megakernel and the three opt-in
    compaction forms. Returns (name, the prefix crt_debug_last_kernel must report)."""
    if request.param == "default":
        monkeypatch.delenv("CRT_KERNEL", raising=False)
    else:
        monkeypatch.setenv("CRT_KERNEL", request.param)
    return request.param, FORMS[request.param]


CASES = [  # seed, triangles per mesh, grid, instance kinds
    (1, [1, 2, 3], 3, ["plain", "mirrored", "plain"]),
    (2, [5, 17, 64], 4, ["plain", "tiny", "huge", "mirrored"]),
    (3, [300, 4], 5, ["plain", "flat", "plain"]),
    (4, [2500, 40], 9, ["plain", "plain", "mirrored", "zero", "huge"]),
    (5, [6000], 17, ["plain", "mirrored", "flat", "tiny"]),
    (6, [128, 129, 127], 2, ["plain", "plain", "plain"]),   # grid of 2: almost everything coincident, leaves that cannot split
]


def build_scene(tmp_path, rng, seed, sizes, grid, kinds):
    paths = [scenes._write_mesh(str(tmp_path), f"soup{i}", soup(rng, n, grid, degenerate=True), [((0.7, 0.5, 0.9), None)]) for i, n in enumerate(sizes)]
    sky = str(tmp_path / "sky.ppm")
    scenes.write_ppm(sky, scenes._skybox(64, 32))
    insts = [scenes.Instance(i % len(sizes), 0xFFFF, matrices(rng, kind)) for i, kind in enumerate(kinds)]
    return scenes.Scene(f"fuzz{seed}", str(tmp_path), sky, paths, insts, (0.5, 1.0, 9.0), scenes._normalize((-0.05, -0.1, -1.0)))


def check_frames(s, orc, sc, rng, prefix, cams, flags=8, tolerated=2):
    """Camera frames + lattice frames of the session's kernel form against the oracle: frame bits, every work counter, and the name
    of the kernel that rendered them. Returns (frames, rays, pixels that differ within the skybox-texel tolerance, rays stopped by the cap)."""
    frames = rays = flips = cap = 0
    views = []
    for cam in cams:
        s.set_camera(cam, scenes._normalize((-cam[0] or -0.05, -0.1 if cam[1] else 0.0, -1.0 if cam[2] else 0.0)))
        iv, ip, pos = s.camera()
        views.append((f"camera {cam}", iv.copy(), ip.copy(), pos.copy()))
    w0, h0 = s.width, s.height
    for group, size in ((views, (w0, h0)), (lattice_views(rng), (256, 128))):
        if (s.width, s.height) != size:
            s.resize(*size)
        for what, iv, ip, pos in group:
            s.render_raw(flags, view=(iv, ip, pos))
            assert s.last_kernel().startswith(prefix), (what, s.last_kernel())
            frame = s.read_output()
            want, fst = orc.trace(orc.raygen(s.width, s.height, iv, ip), pos, sc.sun_angle, shadows=bool(flags & 32))
            assert s.counters() == fst, what
            diff = np.nonzero((bits(frame) != bits(want)).any(axis=2))
            # skybox texel flips from the double atan2/acos (glibc vs OCML) are the one tolerated difference
            assert len(diff[0]) <= tolerated, f"{what}: {len(diff[0])} pixels differ, first {diff[0][:4]},{diff[1][:4]}"
            frames += 1; rays += fst["rays"]; flips += len(diff[0]); cap += fst["capHits"]
    s.resize(w0, h0)
    return frames, rays, flips, cap


@pytest.mark.parametrize("seed,sizes,grid,kinds", CASES, ids=[f"seed{c[0]}" for c in CASES])
def test_pathological_scenes_bit_exact(tmp_path, nthreads, form, seed, sizes, grid, kinds):
    """Every kernel form. Explicit rays go through crt_query_hits, which has a kernel of its own and never runs a compaction form, so
    that half is checked once (default form); the special rays reach the forms as lattice FRAMES (lattice_views)."""
    name, prefix = form
    rng = np.random.default_rng(seed)
    sc = build_scene(tmp_path, rng, seed, sizes, grid, kinds)
    with np.errstate(all="ignore"), driver.Session(208, 120, device=0) as s:
        s.load_scene(sc)
        a = s.arenas()
        orc = oracle_lib.Oracle(a, nthreads=nthreads)
        o, d = special_rays(rng, 8192)
        if name == "default":
            got = s.query_hits(o, d)
            cnt = s.counters()
            ref, st = orc.closest_hits(o, d)
            bad = np.nonzero(np.frombuffer(got.tobytes(), np.uint8).reshape(len(o), -1) != np.frombuffer(ref.tobytes(), np.uint8).reshape(len(o), -1))[0]
            assert len(bad) == 0, f"{len(np.unique(bad))} hit records differ, first ray {bad[0]}: o={o[bad[0]]} d={d[bad[0]]} gpu={got[bad[0]]} oracle={ref[bad[0]]}"
            assert cnt == st
        # outside, inside the soups (H1), along an axis; then the lattice frames
        frames, rays, _, _ = check_frames(s, orc, sc, rng, prefix, ((0.5, 1.0, 9.0), (0.0, 0.0, 0.25), (-7.0, 0.0, 0.0)))
        assert frames == 7 and rays > 0
        # frames in flight of the same form render the same bits
        s.render_raw(0); plain = s.read_output().copy()
        for _ in range(4):
            s.render_raw(4)
        assert s.last_kernel().startswith(prefix) and np.array_equal(bits(s.read_output()), bits(plain))


@pytest.mark.parametrize("seed,sizes,grid,kinds", CASES[1:], ids=[f"seed{c[0]}" for c in CASES[1:]])
def test_pathological_scenes_extensions_and_device_build(tmp_path, nthreads, seed, sizes, grid, kinds):
    """The same scenes through the opt-in paths: shadow rays (any-hit traversal), the device BVH builder (must reproduce the
    host builder's bytes on ties and unsplittable leaves), frames in flight."""
    rng = np.random.default_rng(seed)
    sc = build_scene(tmp_path, rng, seed, sizes, grid, kinds)
    with np.errstate(all="ignore"), driver.Session(208, 120, device=0) as s:
        s.load_scene(sc)
        a = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in s.arenas().items()}
        orc = oracle_lib.Oracle(a, nthreads=nthreads)
        iv, ip, pos = s.camera()
        rays = orc.raygen(s.width, s.height, iv, ip)
        s.render_raw(8 | 32)
        want, st = orc.trace(rays, pos, sc.sun_angle, shadows=True)
        assert s.counters() == st and (st["shadowRays"] > 0 or seed == 6)       # seed 6: every hit has a NaN t, nothing is lit
        assert ((bits(s.read_output()) != bits(want)).any(axis=2)).sum() <= 2
        s.render_raw(0)
        plain = s.read_output().copy()
        for _ in range(4):
            s.render_raw(4)
        assert np.array_equal(bits(s.read_output()), bits(plain))
    with np.errstate(all="ignore"), driver.Session(208, 120, device=0) as s:
        s.load_scene(sc, device_bvh_build=True)
        b = s.arenas()
        for k in ("tris", "nodes", "roots", "instances"):
            assert a[k].tobytes() == b[k].tobytes(), k
        s.render_raw(0)
        assert np.array_equal(bits(s.read_output()), bits(plain))


@pytest.mark.parametrize("seed,sizes,grid,kinds", [CASES[2], CASES[3], CASES[4]], ids=["seed3", "seed4", "seed5"])
def test_pathological_instances_through_the_instance_tree(tmp_path, nthreads, monkeypatch, seed, sizes, grid, kinds):
    """CRT_TLAS=1 forces the instance tree (normally used above 64 instances) onto the scenes whose instance matrices are
    singular, tiny or huge: bounds that are degenerate, Inf or NaN must never hide an instance the linear loop would visit.
    Plus 90 more instances of every kind, so the tree has real depth."""
    monkeypatch.setenv("CRT_TLAS", "1")
    rng = np.random.default_rng(seed)
    sc = build_scene(tmp_path, rng, seed, sizes, grid, kinds)
    all_kinds = ["plain", "mirrored", "tiny", "huge", "flat", "zero"]
    for k in range(90):
        sc.instances.append(scenes.Instance(k % len(sizes), 0xFFFF, matrices(rng, all_kinds[k % len(all_kinds)])))
    with np.errstate(all="ignore"), driver.Session(208, 120, device=0) as s:
        s.load_scene(sc)
        orc = oracle_lib.Oracle(s.arenas(), nthreads=nthreads)
        o, d = special_rays(rng, 8192)
        got = s.query_hits(o, d)
        cnt = s.counters()
        ref, st = orc.closest_hits(o, d)
        assert got.tobytes() == ref.tobytes() and cnt == st
        s.render_raw(8)
        iv, ip, pos = s.camera()
        want, fst = orc.trace(orc.raygen(s.width, s.height, iv, ip), pos, sc.sun_angle)
        assert s.counters() == fst
        assert ((bits(s.read_output()) != bits(want)).any(axis=2)).sum() <= 2
        s.render_raw(8 | 32)
        want, fst = orc.trace(orc.raygen(s.width, s.height, iv, ip), pos, sc.sun_angle, shadows=True)
        assert s.counters() == fst
        assert ((bits(s.read_output()) != bits(want)).any(axis=2)).sum() <= 2
