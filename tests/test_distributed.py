"""N>1 path on CPU: two gloo ranks shard a frame by row bands exactly as bench.py does on GPUs.

There is no CPU render path in the product, so each rank "renders" its bands with the oracle (allowed in tests). The test
checks what the multi-GPU bench relies on: the band ownership function partitions the frame, the shards stitch to the
single-process frame bit for bit with no data-path collective, and bench.aggregate() SUMs the work counters and takes
the MAX of the times across ranks.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, BAND = 96, 80, 16


def _worker(rank, world, port, outdir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        import oracle_lib
        from clraytracer_amd import _lib, driver, scenes
        sc = scenes.get("tiny")
        with driver.Session(W, H, host_only=True) as s:          # replicated scene on every rank
            s.load_scene(sc)
            arenas = s.arenas(); iv, ip, pos = s.camera()
        orc = oracle_lib.Oracle(arenas, nthreads=2)
        rays = orc.raygen(W, H, iv, ip)
        hip = _lib.hip()
        own = np.array([hip.crt_row_owner(y, BAND, world) == rank for y in range(H)])
        frame = np.zeros((H, W, 4), np.float32)
        cnt = {k: 0 for k in bench.COUNTER_KEYS}
        y = 0
        while y < H:                                              # trace each owned band
            if own[y]:
                y1 = y
                while y1 < H and own[y1]:
                    y1 += 1
                part, st = orc.trace(rays, pos, sc.sun_angle, row0=y, row1=y1)
                frame[y:y1] = part[y:y1]
                for k in cnt:
                    cnt[k] += st[k]
                y = y1
            else:
                y += 1
        tot, tmax, kmax = bench.aggregate(dist, cnt, int(own.sum()) * W, elapsed_s=1.0 + rank, kernel_ms_mean=2.0 * (rank + 1), device="cpu")
        np.save(os.path.join(outdir, f"frame{rank}.npy"), frame)
        np.save(os.path.join(outdir, f"own{rank}.npy"), own)
        if rank == 0:
            full, st_full = orc.trace(rays, pos, sc.sun_angle)
            np.save(os.path.join(outdir, "full.npy"), full)
            np.save(os.path.join(outdir, "tot.npy"), np.array([tot["rays"], tot["innerVisits"], tot["pixels"], tmax, kmax,
                                                                 st_full["rays"], st_full["innerVisits"], tot["alg_bytes"],
                                                                 bench.algorithmic_bytes(st_full, W * H)], np.float64))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_row_band_sharding(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    full = np.load(tmp_path / "full.npy")
    owns = [np.load(tmp_path / f"own{r}.npy") for r in range(world)]
    assert np.array_equal(owns[0] ^ owns[1], np.ones(H, bool))           # disjoint and complete
    assert owns[0][:BAND].all() and owns[1][BAND:2 * BAND].all()           # bands alternate
    stitched = np.zeros_like(full)
    for r in range(world):
        fr = np.load(tmp_path / f"frame{r}.npy")
        assert np.all(fr[~owns[r]] == 0)
        stitched[owns[r]] = fr[owns[r]]
    assert np.array_equal(stitched.view(np.uint32), full.view(np.uint32))
    rays, inner, pixels, tmax, kmax, rays_full, inner_full, alg, alg_full = np.load(tmp_path / "tot.npy")
    assert rays == rays_full and inner == inner_full and pixels == W * H  # SUM over ranks
    assert tmax == 2.0 and kmax == 4.0                                     # MAX over ranks
    assert alg == alg_full


def test_algorithmic_bytes_formula():
    sys.path.insert(0, ROOT)
    import bench
    c = dict(innerVisits=10, triTests=3, traversals=2, hits=1, misses=1)
    assert bench.algorithmic_bytes(c, 4) == 64 * 10 + 48 * 3 + 80 * 2 + 214 + 19 + 16 * 4
