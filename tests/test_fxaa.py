"""The oracle's FXAA stage (orc_fxaa) -- an EXTENSION: upstream's FXAA function (kernel_main.cl:289-340) is dead code, its call
is commented out (kernel_main.cl:349), so nothing of upstream's ever produced a frame to pin this against ("parity unpinned").
What can be pinned is the restatement itself: a second, independently written numpy version of the same arithmetic must agree
bit for bit, and the filter must behave like the filter upstream sketches (flat areas untouched, edges blended along the edge)."""
import numpy as np
import pytest

import oracle_lib
from util import bits

F = np.float32


def dot_luma(rgb):
    return (rgb[..., 0] * F(0.299) + rgb[..., 1] * F(0.587)) + rgb[..., 2] * F(0.114)


def texel(img, i, j):
    h, w, _ = img.shape
    return img[np.clip(j, 0, h - 1), np.clip(i, 0, w - 1), :3]


def linear(img, s, t):
    """CLK_NORMALIZED_COORDS_TRUE | CLK_FILTER_LINEAR | CLK_ADDRESS_CLAMP_TO_EDGE (OpenCL 1.2 spec 8.2)."""
    h, w, _ = img.shape
    u = s * F(w) - F(0.5); v = t * F(h) - F(0.5)
    fu = np.floor(u); fv = np.floor(v)
    a = (u - fu)[..., None]; b = (v - fv)[..., None]
    i0 = fu.astype(np.int64); j0 = fv.astype(np.int64)
    one = F(1.0)
    return (((texel(img, i0, j0) * ((one - a) * (one - b)) + texel(img, i0 + 1, j0) * (a * (one - b)))
             + texel(img, i0, j0 + 1) * ((one - a) * b)) + texel(img, i0 + 1, j0 + 1) * (a * b))


def fxaa_numpy(img):
    img = np.ascontiguousarray(img, F)
    h, w, _ = img.shape
    jj, ii = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    uvx = ii.astype(F) / F(w); uvy = jj.astype(F) / F(h)
    rgb = texel(img, ii, jj)
    nw = dot_luma(texel(img, ii - 1, jj - 1)); ne = dot_luma(texel(img, ii + 1, jj - 1))
    sw = dot_luma(texel(img, ii - 1, jj + 1)); se = dot_luma(texel(img, ii + 1, jj + 1))
    m = dot_luma(rgb)
    with np.errstate(all="ignore"):
        dx = -((nw + ne) - (sw + se)); dy = ((nw + sw) - (ne + se))
        reduce_ = np.fmax(((nw + ne) + sw + se) * F(0.25 * 0.125), F(1.0 / 128.0))
        rcp = F(1.0) / (np.fmin(np.abs(dx), np.abs(dy)) + reduce_)
        dx = np.fmin(F(8), np.fmax(F(-8), dx * rcp)) / F(w)
        dy = np.fmin(F(8), np.fmax(F(-8), dy * rcp)) / F(h)
        a = (linear(img, uvx + dx * F(-0.166667), uvy + dy * F(-0.166667)) + linear(img, uvx + dx * F(0.166667), uvy + dy * F(0.166667))) * F(0.5)
        b = a * F(0.5) + (linear(img, uvx + dx * F(-0.5), uvy + dy * F(-0.5)) + linear(img, uvx + dx * F(0.5), uvy + dy * F(0.5))) * F(0.25)
        lb = dot_luma(b)
        lo = np.fmin(m, np.fmin(np.fmin(nw, ne), np.fmin(sw, se)))
        hi = np.fmax(m, np.fmax(np.fmax(nw, ne), np.fmax(sw, se)))
        pick_a = (lb < lo) | (lb > hi)
    out = np.ones_like(img)
    out[..., :3] = np.where(pick_a[..., None], a, b)
    return out


@pytest.mark.parametrize("shape", [(29, 37), (16, 16), (64, 1), (1, 64), (120, 200)])
def test_oracle_fxaa_equals_numpy_restatement(shape):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    h, w = shape
    img = np.ones((h, w, 4), F)
    img[..., :3] = rng.random((h, w, 3), dtype=F) * F(1.5)
    # blocks of flat colour and hard edges, like a traced frame, on top of the noise
    img[h // 3: 2 * h // 3, w // 4: 3 * w // 4, :3] = F(0.25)
    img[: h // 4, :, :3] *= F(0.0)
    got = oracle_lib.fxaa(img)
    assert np.array_equal(bits(got), bits(fxaa_numpy(img)))


def test_flat_frame_is_a_fixed_point_and_edges_are_blended():
    img = np.ones((40, 56, 4), F); img[..., :3] = (0.3, 0.5, 0.7)
    assert np.abs(oracle_lib.fxaa(img) - img).max() < 1e-6  # (i / w) * w is not always i: the four tap weights are not exactly 1/4
    img[:, 28:, :3] = (0.9, 0.9, 0.9)                      # a vertical edge
    out = oracle_lib.fxaa(img)
    assert np.abs(out[:, :26] - img[:, :26]).max() < 1e-6 and np.abs(out[:, 31:] - img[:, 31:]).max() < 1e-6
    col = out[20, 26:31, 0]
    assert np.all(np.diff(col) >= 0) and 0.3 < col[2] < 0.9  # blended across, monotone
    assert np.all(out[..., 3] == 1.0)


def test_rows_and_non_finite_pixels():
    rng = np.random.default_rng(5)
    img = np.ones((33, 47, 4), F); img[..., :3] = rng.random((33, 47, 3), dtype=F)
    full = oracle_lib.fxaa(img)
    part = oracle_lib.fxaa(img, 8, 20)                     # only rows [8, 20) are written
    assert np.array_equal(bits(part[8:20]), bits(full[8:20]))
    assert np.array_equal(bits(part[:8]), bits(img[:8])) and np.array_equal(bits(part[20:]), bits(img[20:]))
    img[10, 10, 1] = np.nan; img[20, 30, 0] = np.inf       # hazard H4 can store NaN: the filter must stay defined
    out = oracle_lib.fxaa(img)
    assert np.array_equal(bits(out), bits(fxaa_numpy(img)))
    far = np.ones(img.shape[:2], bool); far[3:18, 3:18] = False; far[13:28, 23:38] = False
    assert np.isfinite(out[far]).all()
