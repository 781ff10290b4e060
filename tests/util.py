"""Shared helpers for the parity tests."""
import numpy as np


def seeded_rays(scene_arenas, cam_pos, n, seed):
    """n world-space rays: half from the camera towards the scene, half random origins/directions."""
    rng = np.random.RandomState(seed)
    tris = scene_arenas["tris"]
    inst = scene_arenas["instances"]
    # aim at random triangle centroids pushed through a random instance transform (approximate world
    # positions are good enough: we only need rays that hit things)
    k = rng.randint(0, len(tris), size=n)
    target = (tris["v0"][k] + tris["v1"][k] + tris["v2"][k]) / 3.0
    ii = rng.randint(0, len(inst), size=n)
    inv = inst["inv"][ii].astype(np.float64)
    fwd = np.linalg.inv(inv)
    tw = np.einsum("ni,nij->nj", np.concatenate([target.astype(np.float64), np.ones((n, 1))], 1), fwd)[:, :3]
    origins = np.tile(np.asarray(cam_pos, np.float64), (n, 1))
    half = n // 2
    origins[half:] = tw[half:] + rng.normal(size=(n - half, 3)) * 8.0 + np.array([0, 6.0, 0])
    d = tw + rng.normal(size=(n, 3)) * 0.05 - origins
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return origins.astype(np.float32), d.astype(np.float32)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def rmse(a, b):
    d = a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)
    return float(np.sqrt(np.mean(d * d)))
