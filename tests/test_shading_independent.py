"""Independent evidence for SURVEY.md 8a rows a7-a10 (miss -> skybox texel, attribute fetch + interpolation, SampleTexture + colour,
shading + bounce set-up): a numpy float32 restatement written from the reference's OpenCL text alone --
/root/reference/CLRayTracer/kernels/kernel_main.cl:177-274 (the body of `Trace` around the instance loop) and MathAndSTL.cl:100-119
(MatMul / Mat3Mul / reflect), :123-125 (constants), :243-266 (MultiplyColorU32, UNPACK_RGB8, SampleSkyboxPixel, SampleTexture) --
without consulting oracle/crt_oracle.c, and compared on the CPU with the C oracle's frames.

What is shared with the oracle on purpose: the closest-hit records (`orc_closest_hits`: rows a3-a6, which have their own independent
check, tests/test_brute_force.py), the RayGen buffer (row a1, bit-pinned elsewhere) and the *pinned builtin semantics* listed in
oracle/crt_oracle.h (dot/normalize/fmax/(int)/out-of-pool texel clamp ...), which are the specification both sides follow. Everything
else -- which instance matrix feeds what, the half decoding, barycentric weights, operation order of every sum, the colour integer
arithmetic, texture addressing, the shading terms, what carries over to the bounce ray -- is restated here in a different language
by vectorised array code that shares no line with the oracle.

Tolerance: bit-exact wherever only + - * / sqrt, comparisons and integer arithmetic are involved (every shaded pixel). The skybox index
goes through atan2/acos in double (numpy's versus glibc's): a pixel whose index differs by that last-place effect may flip to a
neighbouring texel; such pixels are counted and must stay below 1e-4 of the frame, and every other pixel must match bit for bit.
"""
import numpy as np
import pytest

from clraytracer_amd import driver, scenes
import oracle_lib

F = np.float32


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def dot3(a, b):                                     # pinned: (a.x*b.x + a.y*b.y) + a.z*b.z
    return (a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]) + a[:, 2] * b[:, 2]


def normalize(v):                                   # pinned: v * (1 / sqrt(dot(v, v)))
    inv = F(1.0) / np.sqrt(dot3(v, v))
    return v * inv[:, None]


def reflect(v, n):                                  # MathAndSTL.cl:117-119: v - n * dot(n, v) * 2.0f
    return v - (n * dot3(n, v)[:, None]) * F(2.0)


def mat_mul_xyz(m, v, w):
    """MathAndSTL.cl:100-102 on (v, w): m.x * v.xxxx + m.y * v.yyyy + m.z * v.zzzz + m.w * v.wwww, .xyz; m: (N, 4, 4) as stored"""
    return ((m[:, 0, :3] * v[:, 0:1] + m[:, 1, :3] * v[:, 1:2]) + m[:, 2, :3] * v[:, 2:3]) + m[:, 3, :3] * F(w)


def mat3_mul(m, v):
    """MathAndSTL.cl:104-106 on ConvertToMatrix3(m) (the xyz of rows x, y, z)"""
    return (m[:, 0, :3] * v[:, 0:1] + m[:, 1, :3] * v[:, 1:2]) + m[:, 2, :3] * v[:, 2:3]


def to_int(x):                                      # pinned: truncation, NaN -> 0, saturating
    x = np.asarray(x, np.float32)
    y = np.where(np.isnan(x), F(0), x)
    y = np.clip(y.astype(np.float64), -2147483648.0, 2147483647.0)
    return np.trunc(y).astype(np.int64)


def half(h):
    return np.ascontiguousarray(h, np.uint16).view(np.float16).astype(np.float32)


def fetch_texel(texels, idx):
    """texturePixels[idx] of the packed RGB8 pool; an index outside the pool is clamped (pinned)"""
    n = (len(texels) + 2) // 3
    i = np.clip(idx, 0, n - 1)
    return texels[3 * i].astype(np.uint32), texels[3 * i + 1].astype(np.uint32), texels[3 * i + 2].astype(np.uint32)


def trace_numpy(a, orc, rays, cam_pos, sun_angle):
    """kernel_main.cl:177-274 for every pixel at once; returns (rgb frame, mask of pixels that sampled the skybox)"""
    h, w, _ = rays.shape
    n = h * w
    texels = np.ascontiguousarray(a["texels"], np.uint8)
    tex = a["textures"]
    d = np.ascontiguousarray(rays.reshape(n, 3), np.float32).copy()
    o = np.tile(np.asarray(cam_pos, np.float32), (n, 1))
    sun = np.float32(sun_angle)
    light = np.tile(np.array([0.0, F(np.sin(np.float64(sun))), F(np.cos(np.float64(sun)))], np.float32), (n, 1))      # :181
    result = np.zeros((n, 3), np.float32)
    energy = np.ones((n, 3), np.float32)
    atm = np.tile(np.array([0.255, 0.25, 0.27], np.float32) * F(1.0), (n, 1))                                          # :185
    alive = np.arange(n)
    sky_mask = np.zeros(n, bool)
    u255 = F(1.0) / F(255.0)                                                                                            # MathAndSTL.cl:125
    for bounce in range(2):                                                                                             # :187
        if len(alive) == 0:
            break
        rec, _ = orc.closest_hits(o[alive], d[alive])
        miss = rec["t"] > F(99998.0)                                                                                    # :219, InfMinusOne
        # ---- miss: skybox (kernel_main.cl:219-224, MathAndSTL.cl:253-258; textures[2], pool offset 2) ----
        mi = alive[miss]
        if len(mi):
            dm = d[mi]
            tw, th = int(tex[2]["width"]), int(tex[2]["height"])
            atan2pi = (np.arctan2(dm[:, 0].astype(np.float64), (-dm[:, 2]).astype(np.float64)) / np.pi).astype(np.float32)
            acospi = (np.arccos(dm[:, 1].astype(np.float64)) / np.pi).astype(np.float32)
            theta = to_int((atan2pi * F(0.5)) * F(tw))
            phi = to_int(acospi * F(th))
            r, g, b = fetch_texel(texels, phi * tw + (theta + 2))                                                       # mad24(phi, width, theta + 2)
            skyc = np.stack([r, g, b], 1).astype(np.float32) * u255
            result[mi] = result[mi] + skyc * energy[mi]
            sky_mask[mi] = True
        # ---- hit (kernel_main.cl:226-271) ----
        hi = alive[~miss]
        if len(hi) == 0:
            break
        hr = rec[~miss]
        inst = a["instances"][hr["instance"]]
        m = np.ascontiguousarray(inst["inv"], np.float32)                       # hitInstance.inverseTransform, rows x, y, z, w
        tri = a["tris"][hr["tri"]]
        mat_index = np.minimum(inst["materialStart"].astype(np.int64) + tri["mat"].astype(np.int64), 255)
        mat = a["materials"][mat_index]
        uu, vv = hr["u"].astype(np.float32), hr["v"].astype(np.float32)
        bx, by, bz = (F(1.0) - uu) - vv, uu, vv                                                                          # :231
        nh = half(tri["n"])                                                     # normal0, normal1, normal2
        n0, n1, n2 = mat3_mul(m, nh[:, 0:3]), mat3_mul(m, nh[:, 3:6]), mat3_mul(m, nh[:, 6:9])
        normal = normalize((n0 * bx[:, None] + n1 * by[:, None]) + n2 * bz[:, None])                                    # :237
        uvh = half(tri["uv"])
        uv = (uvh[:, 0:2] * bx[:, None] + uvh[:, 2:4] * by[:, None]) + uvh[:, 4:6] * bz[:, None]                        # :239-241
        t_idx = np.minimum(mat["albedo"].astype(np.int64), 31)
        tx = tex[t_idx]
        uvf = uv - np.floor(uv)                                                                                         # MathAndSTL.cl:262
        us = to_int(tx["width"].astype(np.float32) * uvf[:, 0])
        vs = to_int(tx["height"].astype(np.float32) * uvf[:, 1])
        pr, pg, pb = fetch_texel(texels, vs * tx["width"].astype(np.int64) + tx["offset"].astype(np.int64) + us)        # :265
        col = mat["color"].astype(np.uint32)
        cr = (((col & 0xff) * pr) >> 8) & 0xff                                                                          # MathAndSTL.cl:243-249 (uchar stores)
        cg = ((((col >> 8) & 0xff) * pg) >> 8) & 0xff
        cb = ((((col >> 16) & 0xff) * pb) >> 8) & 0xff
        color = np.stack([cr, cg, cb], 1).astype(np.float32) * u255
        mo = mat_mul_xyz(m, o[hi], 1.0)                                          # meshRay of the winning instance (:206-207, :215)
        md = mat_mul_xyz(m, d[hi], 0.0)
        point = mo + hr["t"].astype(np.float32)[:, None] * md                                                            # :246
        new_o = point + normal * F(0.01)                                                                                # :252-253
        new_d = reflect(d[hi], normal)                                                                                  # :254
        L = light[hi]
        ndl = dot3(normal, -L)                                                                                          # :261
        ambient = (np.fmax(F(0.0) - ndl, F(0.1))[:, None] * atm[hi]) * color                                            # :262
        ndl = np.fmax(ndl, F(0.0))                                                                                      # :263
        specular = (((F(1.0) - F(0.5)) * ndl) * F(1.0))[:, None] * np.array([0.2, 0.2, 0.2], np.float32) * ndl[:, None]   # :264 (roughness .5, shadow 1, specularColor .2)
        rl = reflect(-L, normal)
        spec_light = (ndl * np.fmax(dot3(rl, md), F(0.0))) * F(0.2)                                                     # :265 (pow(x, 1.0f) == x)
        result[hi] = result[hi] + ((energy[hi] * (color * ndl[:, None]) + ambient) + spec_light[:, None])                # :267
        energy[hi] = energy[hi] * specular                                                                              # :268
        atm[hi] = atm[hi] * F(0.4)                                                                                      # :269
        light[hi] = new_d                                                                                               # :271
        o[hi], d[hi] = new_o, new_d
        alive = hi
    return result.reshape(h, w, 3), sky_mask.reshape(h, w)


@pytest.mark.parametrize("name,w,h", [("tiny", 512, 288), ("cornell-1k", 512, 288), ("sponza-sibenik", 640, 360), ("nanosuit-demo", 512, 288)])
def test_numpy_restatement_of_shading_matches_the_oracle(name, w, h, nthreads):
    sc = scenes.get(name)
    with driver.Session(w, h, host_only=True) as s:
        s.load_scene(sc)
        a = {k: (np.array(v) if isinstance(v, np.ndarray) else v) for k, v in s.arenas().items()}
        iv, ip, pos = s.camera()
    orc = oracle_lib.Oracle(a, nthreads=nthreads)
    rays = orc.raygen(w, h, iv, ip)
    ref, st = orc.trace(rays, pos, sc.sun_angle)
    got, sky = trace_numpy(a, orc, rays, pos, sc.sun_angle)
    differs = (bits(got) != bits(ref[..., :3])).any(axis=2)
    shaded_only = ~sky
    # every pixel that never touched the skybox: + - * / sqrt, comparisons, integers -> bit for bit
    assert st["hits"] >= 10000, st
    assert not (differs & shaded_only).any(), f"{int((differs & shaded_only).sum())} shaded pixels differ"
    # pixels with a skybox sample: the index passes through double atan2 / acos of two different math libraries
    flips = int((differs & sky).sum())
    assert flips <= 1e-4 * w * h, f"{flips} skybox pixels differ"
    if flips:
        # a flipped texel changes one bounce's sky colour, nothing else: bounded by the energy (<= 1) times one texel (<= 1)
        assert np.nanmax(np.abs(got[differs & sky] - ref[..., :3][differs & sky])) <= 1.0
    assert np.array_equal(bits(ref[..., 3]), bits(np.ones((h, w), np.float32)))                                         # :274 (result, 1.0f)
    print(f"{name}: {st['hits']} shaded hit records, {st['misses']} skybox samples, {int(differs.sum())} pixels differ ({flips} skybox flips)")


def postprocess_numpy(img):
    """kernel PostProcess (kernel_main.cl:342-359) with Saturation / Reinhard / GammaCorrect / Vignette (MathAndSTL.cl:132-169) for a whole
    float frame at once, written from that text: float32 throughout, sums in the order the source writes them, pow through float32 powf."""
    h, w, _ = img.shape
    rgb = np.ascontiguousarray(img[..., :3], np.float32).reshape(-1, 3).copy()
    ys, xs = np.mgrid[0:h, 0:w]
    uvx = (xs.reshape(-1).astype(np.float32)) / F(w)
    uvy = (ys.reshape(-1).astype(np.float32)) / F(h)
    # Saturation(rgb, 1.2f): P = sqrt(in.x*in.x*0.299f + (in.y*in.y*0.587f) + (in.z*in.z*0.114f)); P + (in - P) * change
    P = np.sqrt((rgb[:, 0] * rgb[:, 0]) * F(0.299) + ((rgb[:, 1] * rgb[:, 1]) * F(0.587)) + ((rgb[:, 2] * rgb[:, 2]) * F(0.114)))
    rgb = P[:, None] + (rgb - P[:, None]) * F(1.2)
    # Reinhard: luminance-preserving extended Reinhard with max_white_l = 0.8, then pow(x, 1 / 1.55)
    lw = np.array([[0.2126, 0.7152, 0.0722]], np.float32)
    with np.errstate(all="ignore"):
        l_old = dot3(rgb, np.broadcast_to(lw, rgb.shape))
        numerator = l_old * (F(1.0) + (l_old / (F(0.8) * F(0.8))))
        l_new = numerator / (F(1.0) + l_old)
        l_in = dot3(rgb, np.broadcast_to(lw, rgb.shape))
        rgb = rgb * (l_new / l_in)[:, None]
        rgb = np.power(rgb, F(1.0) / F(1.55), dtype=np.float32)
        rgb = np.power(rgb, F(1.0) / F(1.2), dtype=np.float32)                  # GammaCorrect
        # Vignette(uv): uv *= 1 - uv.yx; vig = uv.x * uv.y * 15; pow(vig, 0.15)
        vx, vy = uvx * (F(1.0) - uvy), uvy * (F(1.0) - uvx)
        vig = np.power((vx * vy) * F(15.0), F(0.15), dtype=np.float32)
    out = np.ones((h * w, 4), np.float32)
    out[:, :3] = rgb * vig[:, None]
    return out.reshape(h, w, 4)


@pytest.mark.parametrize("name", ["tiny", "sponza-sibenik"])
def test_numpy_restatement_of_postprocess_matches_the_oracle(name, nthreads):
    """SURVEY row a11 the same way: the oracle's PostProcess against a numpy restatement of the OpenCL text on real Trace frames (black pixels
    included: their 0/0 makes NaN on both sides). Tolerance 2e-5 absolute -- the only difference between the two is the last place of three powf
    calls per channel (numpy's float32 power versus glibc's powf), as between the oracle and the device (tests/test_gpu_parity.py)."""
    w, h = 384, 216
    sc = scenes.get(name)
    with driver.Session(w, h, host_only=True) as s:
        s.load_scene(sc)
        a = {k: (np.array(v) if isinstance(v, np.ndarray) else v) for k, v in s.arenas().items()}
        iv, ip, pos = s.camera()
    orc = oracle_lib.Oracle(a, nthreads=nthreads)
    frame, _ = orc.trace(orc.raygen(w, h, iv, ip), pos, sc.sun_angle)
    frame[0, :8, :3] = 0.0                                    # a few black pixels: Reinhard divides 0 by 0 (MathAndSTL.cl:137-141)
    ref = orc.postprocess(frame)
    got = postprocess_numpy(frame)
    assert np.array_equal(np.isnan(ref), np.isnan(got)) and np.isnan(ref[0, :8, :3]).all()
    ok = ~np.isnan(ref)
    assert np.max(np.abs(ref[ok] - got[ok])) <= 2e-5
    assert (bits(ref)[ok] == bits(got)[ok]).mean() > 0.5      # most values agree to the bit; the rest differ in the last place of powf


def test_numpy_restatement_of_raygen_matches_the_oracle():
    """SURVEY row a1 (kernel_main.cl:277-287): coord = (i / W, j / H) * 2 - 1; target = invProj . (coord, 1, 1); target /= target.w;
    dir = normalize((invView . target).xyz) -- MatMul in the row-vector convention of MathAndSTL.cl:100-102, every sum left to right."""
    w, h = 640, 360
    sc = scenes.get("tiny")
    with driver.Session(w, h, host_only=True) as s:
        s.load_scene(sc)
        a = {k: (np.array(v) if isinstance(v, np.ndarray) else v) for k, v in s.arenas().items()}
        iv, ip, pos = s.camera()
    ref = oracle_lib.Oracle(a, nthreads=2).raygen(w, h, iv, ip)
    ivm, ipm = np.asarray(iv, np.float32).reshape(4, 4), np.asarray(ip, np.float32).reshape(4, 4)
    jj, ii = np.mgrid[0:h, 0:w]
    cx = (ii.astype(np.float32) / F(w)) * F(2.0) - F(1.0)
    cy = (jj.astype(np.float32) / F(h)) * F(2.0) - F(1.0)

    def matmul(m, x, y, z, wv):          # m.x * v.xxxx + m.y * v.yyyy + m.z * v.zzzz + m.w * v.wwww
        return [((m[0, c] * x + m[1, c] * y) + m[2, c] * z) + m[3, c] * wv for c in range(4)]
    one = np.ones_like(cx)
    t = matmul(ipm, cx, cy, one, one)
    t = [t[0] / t[3], t[1] / t[3], t[2] / t[3], t[3] / t[3]]
    v = matmul(ivm, *t)
    inv = F(1.0) / np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
    got = np.stack([v[0] * inv, v[1] * inv, v[2] * inv], axis=-1).astype(np.float32)
    assert np.array_equal(bits(got), bits(ref))
