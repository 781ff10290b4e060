"""Known-answer tests for the oracle's scalar pieces and host-mirror <-> oracle bit parity.

The reference has no tests of its own, so these KATs are derived by hand from
the reference source semantics (cited inline) and pin both restatements against each other.
"""
import ctypes as C

import numpy as np
import pytest

from clraytracer_amd import _lib
import oracle_lib
from oracle_lib import f32

L = oracle_lib.lib()
H = _lib.host()


def tri_test(o, d, x, y, z, t0=99999.0, u0=0.0, v0=0.0, idx0=7, i=3):
    tuv = np.array([t0, u0, v0], np.float32)
    ti = C.c_uint32(idx0)
    p = [f32(np.array(a, np.float32)) for a in (o, d, x, y, z)]
    passed = L.orc_intersect_triangle(p[0][0], p[1][0], p[2][0], p[3][0], p[4][0], tuv.ctypes.data_as(C.POINTER(C.c_float)), C.byref(ti), i)
    return passed, tuv, ti.value


X, Y, Z = (0, 0, 0), (1, 0, 0), (0, 1, 0)


def test_triangle_front_hit():
    p, tuv, idx = tri_test((0.25, 0.25, 1), (0, 0, -1), X, Y, Z)
    assert p == 1 and idx == 3
    assert tuv.tolist() == [1.0, 0.25, 0.25]


def test_triangle_behind_origin_rejected():
    p, tuv, idx = tri_test((0.25, 0.25, -1), (0, 0, -1), X, Y, Z)
    assert p == 0 and idx == 7 and tuv.tolist() == [99999.0, 0.0, 0.0]


def test_triangle_farther_than_best_rejected():
    p, tuv, idx = tri_test((0.25, 0.25, 1), (0, 0, -1), X, Y, Z, t0=0.5)
    assert p == 0 and tuv[0] == 0.5


def test_triangle_edges_inclusive():
    # u == 0, v == 0, u+v == 1 all pass (strict comparisons, kernel_main.cl:99)
    assert tri_test((0, 0.5, 1), (0, 0, -1), X, Y, Z)[0] == 1
    assert tri_test((0.5, 0, 1), (0, 0, -1), X, Y, Z)[0] == 1
    assert tri_test((0.5, 0.5, 1), (0, 0, -1), X, Y, Z)[0] == 1
    assert tri_test((0.75, 0.5, 1), (0, 0, -1), X, Y, Z)[0] == 0


def test_triangle_in_plane_ray_passes_with_nan():
    # ray inside the triangle's plane: a == 0 -> f = inf (no epsilon test, kernel_main.cl:90) and every dot
    # product is 0, so u = v = t = inf*0 = NaN; every comparison is false -> passed == 1, NaN stored (hazard H4)
    p, tuv, idx = tri_test((0.25, 0.25, 0), (1, 0, 0), X, Y, Z)
    assert p == 1 and idx == 3 and np.all(np.isnan(tuv))


def test_triangle_parallel_ray_poisons_t_through_the_blend():
    # parallel ray off the plane: t = u = +inf -> rejected, but the arithmetic blend (kernel_main.cl:101-104)
    # computes inf*0 + 1*old = NaN: the running best t becomes NaN although nothing was hit
    p, tuv, idx = tri_test((0.25, 0.25, 1), (1, 0, 0), X, Y, Z)
    assert p == 0 and idx == 7 and np.all(np.isnan(tuv))
    # with a NaN best-t the XOR term is (t>0)^0: triangles *behind* the origin now pass, the index
    # follows them, and t stays NaN for good (t*1 + 0*NaN)
    p2, tuv2, idx2 = tri_test((0.25, 0.25, -1), (0, 0, -1), X, Y, Z, t0=float("nan"))
    assert p2 == 1 and idx2 == 3 and np.isnan(tuv2[0])
    p3, tuv3, idx3 = tri_test((0.25, 0.25, 1), (0, 0, -1), X, Y, Z, t0=float("nan"))
    assert p3 == 0 and idx3 == 7            # a proper front hit is now rejected


def aabb(o, d, bmin, bmax, best=99999.0):
    inv = (1.0 / np.array(d, np.float32)).astype(np.float32)
    p = [f32(np.array(a, np.float32)) for a in (o, inv, bmin, bmax)]
    return L.orc_intersect_aabb(p[0][0], p[1][0], p[2][0], p[3][0], best)


def test_aabb_hit_returns_tnear():
    assert aabb((0, 0, 5), (1e-3, 1e-3, -1), (-1, -1, -1), (1, 1, 1)) == pytest.approx(4.0, rel=1e-5)


def test_aabb_origin_inside_is_rejected():  # hazard H1, kernel_main.cl:115 (tnear > 0)
    assert aabb((0, 0, 0), (0.3, 0.2, -1), (-1, -1, -1), (1, 1, 1)) == np.float32(1e30)


def test_aabb_beyond_best_is_rejected():
    assert aabb((0, 0, 5), (1e-3, 1e-3, -1), (-1, -1, -1), (1, 1, 1), best=3.0) == np.float32(1e30)


def test_aabb_flat_box_never_entered():
    # zero-thickness box: tnear == tfar on that axis, and the test is strict (tnear < tfar)
    assert aabb((0.1, 5, 0.1), (1e-3, -1, 1e-3), (-1, 0, -1), (1, 0, 1)) == np.float32(1e30)


def test_half_conversions_all_bit_patterns():
    hs = np.arange(65536, dtype=np.uint16)
    ieee = hs.view(np.float16).astype(np.float32)
    ours = np.array([L.orc_half_to_float(int(h)) for h in hs], np.float32)
    fin = ~np.isnan(ieee)
    assert np.array_equal(ours[fin].view(np.uint32), ieee[fin].view(np.uint32))
    assert np.all(np.isnan(ours[~fin]))
    # the reference's bit-hack variant (Math.hpp:156-164) agrees on every finite normal/subnormal value
    ref = np.array([L.orc_half_to_float_ref(int(h)) for h in hs], np.float32)
    finite = np.isfinite(ieee)
    assert np.array_equal(ref[finite].view(np.uint32), ieee[finite].view(np.uint32))
    host = np.array([H.crth_half_to_float(int(h)) for h in hs], np.float32)
    assert np.array_equal(host.view(np.uint32), ref.view(np.uint32))


def test_float_to_half_round_half_up_and_saturation():
    # Math.hpp:190-197 adds 0x1000 before truncating: ties round up (not to even), out of range -> 0x7FFF
    assert L.orc_float_to_half(1.0) == 0x3C00
    assert L.orc_float_to_half(-2.0) == 0xC000
    assert L.orc_float_to_half(0.0) == 0
    tie = np.float32(1.0 + 2.0 ** -11)          # exactly between 0x3C00 and 0x3C01
    assert L.orc_float_to_half(float(tie)) == 0x3C01
    assert np.float32(tie).astype(np.float16).view(np.uint16) == 0x3C00   # IEEE would round to even
    assert L.orc_float_to_half(1e6) & 0x7FFF == 0x7FFF
    rng = np.random.RandomState(5)
    vals = np.concatenate([rng.uniform(-4, 4, 4000), rng.uniform(-1e-4, 1e-4, 2000), 10.0 ** rng.uniform(-9, 6, 2000)]).astype(np.float32)
    for v in vals:
        assert L.orc_float_to_half(float(v)) == H.crth_float_to_half(float(v))
    ok = np.abs(vals) < 60000
    ours = np.array([L.orc_float_to_half(float(v)) for v in vals[ok]], np.uint16).view(np.float16).astype(np.float32)
    big = np.abs(vals[ok]) > 1e-4
    assert np.max(np.abs(ours[big] - vals[ok][big]) / np.abs(vals[ok][big])) < 1e-3


def test_matrix_helpers_host_equals_oracle_and_are_inverses():
    rng = np.random.RandomState(3)
    for k in range(200):
        # TRS matrix, row-vector convention
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        x, y, z, w = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y + z * w), 2 * (x * z - y * w)],
                      [2 * (x * y - z * w), 1 - 2 * (x * x + z * z), 2 * (y * z + x * w)],
                      [2 * (x * z + y * w), 2 * (y * z - x * w), 1 - 2 * (x * x + y * y)]])
        M = np.eye(4); M[:3, :3] = R * rng.uniform(0.5, 2.0); M[3, :3] = rng.uniform(-20, 20, 3)
        M = M.astype(np.float32)
        a = np.zeros(16, np.float32); b = np.zeros(16, np.float32)
        pm, keep = f32(M)
        L.orc_inverse_transform(pm, a.ctypes.data_as(C.POINTER(C.c_float)))
        H.crth_inverse_transform(pm, b.ctypes.data_as(C.POINTER(C.c_float)))
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert np.allclose(M.astype(np.float64) @ a.reshape(4, 4), np.eye(4), atol=2e-5)
        G = (M + rng.normal(size=(4, 4)).astype(np.float32) * 0.1).astype(np.float32)
        pg, keep2 = f32(G)
        L.orc_inverse(pg, a.ctypes.data_as(C.POINTER(C.c_float)))
        H.crth_inverse(pg, b.ctypes.data_as(C.POINTER(C.c_float)))
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert np.allclose(a.reshape(4, 4), np.linalg.inv(G.astype(np.float64)), rtol=2e-3, atol=2e-4)


def test_camera_matrices_host_equals_oracle():
    a = np.zeros(16, np.float32); b = np.zeros(16, np.float32)
    pa, pb = a.ctypes.data_as(C.POINTER(C.c_float)), b.ctypes.data_as(C.POINTER(C.c_float))
    for (w, h) in ((1920, 1080), (1249, 720), (640, 480), (3840, 2160)):
        fov = np.float32(65.0) * (np.float32(3.14159265358) / np.float32(180.0))
        L.orc_perspective_fov_rh(float(fov), w, h, 0.01, 500.0, pa)
        H.crth_perspective_fov_rh(float(fov), w, h, 0.01, 500.0, pb)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        P = a.reshape(4, 4)
        assert P[1, 1] == pytest.approx(1.0 / np.tan(np.deg2rad(65.0) / 2), rel=1e-3)   # polynomial Sin/Cos (Math.hpp:92-112)
        assert P[0, 0] == pytest.approx(P[1, 1] * h / w, rel=1e-6) and P[2, 3] == -1.0 and P[3, 3] == 0.0
    rng = np.random.RandomState(9)
    for k in range(50):
        eye = rng.uniform(-30, 30, 3).astype(np.float32)
        fr = rng.normal(size=3); fr[1] *= 0.3; fr = (fr / np.linalg.norm(fr)).astype(np.float32)
        up = np.array([0, 1, 0], np.float32)
        pe, k1 = f32(eye); pf, k2 = f32(fr); pu, k3 = f32(up)
        L.orc_look_at_rh(pe, pf, pu, pa)
        H.crth_look_at_rh(pe, pf, pu, pb)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        V = a.reshape(4, 4).astype(np.float64)
        assert np.allclose(V[:3, :3].T @ V[:3, :3], np.eye(3), atol=1e-5)          # orthonormal
        assert np.allclose(np.append(eye, 1) @ V, [0, 0, 0, 1], atol=1e-4)          # eye maps to the origin
        assert np.allclose(np.append(fr, 0) @ V, [0, 0, -1, 0], atol=1e-5)          # front maps to -z (RH)


def test_sampling_helpers():
    tex = np.zeros(1, _lib.TEXTURE_DTYPE); tex["width"] = 8; tex["height"] = 4; tex["offset"] = 10
    t = tex.ctypes.data
    assert L.orc_sample_texture(t, 0.0, 0.0) == 10
    assert L.orc_sample_texture(t, 0.5, 0.5) == 2 * 8 + 10 + 4
    assert L.orc_sample_texture(t, 1.25, 3.75) == 3 * 8 + 10 + 2          # wrap: uv -= floor(uv)
    assert L.orc_sample_texture(t, -0.25, 0.0) == 10 + 6                  # floor, not trunc, on the device
    assert L.orc_sample_texture(t, float("nan"), 0.0) == 10               # (int)NaN pinned to 0
    # skybox: index = phi*width + theta + 2, ignoring texture.offset (hazard H9)
    d = np.array([0, 0, -1], np.float32)                                   # atan2pi(0,1) = 0, acospi(0) = .5
    assert L.orc_sample_skybox(f32(d)[0], t) == 2 * 8 + 0 + 2
    d = np.array([0, 1, 0], np.float32)
    assert L.orc_sample_skybox(f32(d)[0], t) == 0 * 8 + 4 + 2              # phi = 0; atan2(+0, -0.0) = pi -> theta = 4
    d = np.array([-1, 0, 0], np.float32)                                   # atan2pi(-1, -0) = -0.5 -> theta = -2
    assert L.orc_sample_skybox(f32(d)[0], t) == 2 * 8 - 2 + 2
    out = np.zeros(3, np.float32)
    rgb = np.array([255, 128, 1], np.uint8)
    L.orc_multiply_color(rgb.ctypes.data, 0x00FF80FF, out.ctypes.data_as(C.POINTER(C.c_float)))   # MathAndSTL.cl:243-249
    exp = np.array([(255 * 255) >> 8, (128 * 128) >> 8, (255 * 1) >> 8], np.float32) * np.float32(1.0 / 255.0)
    assert np.array_equal(out, exp)


def test_powf_identity_for_shininess_one():
    # the HIP kernel drops pow(x, 1.0f) (kernel_main.cl:250,265); the oracle keeps powf: must agree exactly
    import ctypes.util
    m = C.CDLL(ctypes.util.find_library("m"))
    m.powf.restype = C.c_float; m.powf.argtypes = [C.c_float, C.c_float]
    rng = np.random.RandomState(1)
    for v in np.concatenate([[0.0, 1.0, 1e-30, 1e-42, 3.0e38], rng.uniform(0, 2, 2000), 10.0 ** rng.uniform(-38, 0, 2000)]).astype(np.float32):
        assert np.float32(m.powf(float(v), 1.0)).view(np.uint32) == np.float32(v).view(np.uint32)
