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


def test_band_plan_partitions_every_frame():
    """crt_band_plan (the block list behind the multi-GPU gather and the band-only read-back) against crt_row_owner, for
    ragged heights, several band heights and rank counts: the plans of all ranks tile the frame exactly once."""
    sys.path.insert(0, ROOT)
    import ctypes as C
    from clraytracer_amd import _lib
    hip = _lib.hip()
    for height in (16, 17, 48, 200, 360, 1080, 2160, 2161):
        for band in (8, 16, 32):
            for n in (1, 2, 3, 4, 8):
                seen = np.zeros(height, np.int32)
                for r in range(n):
                    out = (C.c_int * 4)()
                    assert hip.crt_band_plan(height, band, r, n, out) == 0
                    first, full, tail_row, tail_rows = list(out)
                    rows = [first + k * band * n + j for k in range(full) for j in range(band)] + [tail_row + j for j in range(tail_rows)]
                    assert all(0 <= y < height and hip.crt_row_owner(y, band, n) == r for y in rows), (height, band, n, r)
                    seen[rows] += 1
                assert (seen == 1).all(), (height, band, n)
    assert hip.crt_band_plan(100, 12, 0, 2, (C.c_int * 4)()) != 0          # band height must be a multiple of the tile height


def _gather_worker(rank, world, port, outdir):
    """Each rank fills the rows crt_band_plan gives it in its own frame-shaped buffer, then the bands travel to rank 0 with
    point-to-point sends that follow the same block list (what hipMemcpy2DAsync does between devices in crt_init_devices
    sessions): rank 0 must end up with the whole frame."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import ctypes as C
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from clraytracer_amd import _lib
        hip = _lib.hip()
        Hh, Ww = 90, 40                                              # 5.6 bands of 16 rows: a partial band at the bottom
        truth = torch.arange(Hh * Ww * 4, dtype=torch.float32).reshape(Hh, Ww, 4)

        def blocks(r):
            out = (C.c_int * 4)()
            assert hip.crt_band_plan(Hh, BAND, r, world, out) == 0
            first, full, tail_row, tail_rows = list(out)
            return [(first + k * BAND * world, BAND) for k in range(full)] + ([(tail_row, tail_rows)] if tail_rows else [])

        frame = torch.zeros(Hh, Ww, 4)
        for y0, n in blocks(rank):
            frame[y0:y0 + n] = truth[y0:y0 + n]                     # "render" the owned bands
        if rank == 0:
            for src in range(1, world):
                for y0, n in blocks(src):
                    buf = torch.empty(n, Ww, 4)
                    dist.recv(buf, src=src)
                    frame[y0:y0 + n] = buf
            assert torch.equal(frame, truth)
            np.save(os.path.join(outdir, "gathered.npy"), frame.numpy())
        else:
            for y0, n in blocks(rank):
                dist.send(frame[y0:y0 + n].contiguous(), dst=0)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_follows_the_band_plan(tmp_path):
    world = 2
    port = 29100 + (os.getpid() % 500)
    mp.spawn(_gather_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "gathered.npy")
    assert np.array_equal(got, np.arange(90 * 40 * 4, dtype=np.float32).reshape(90, 40, 4))


def _node_barrier_worker(rank, world, port, outdir):
    """bench.py's timed-region bracket on CPU ranks: the shared-memory node barrier set up over the gloo group. A shared counter that is only
    consistent if nobody passes barrier k before everybody has arrived at it; then the timing pattern of the bench (barrier, t0, work, barrier, t1)."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from clraytracer_amd.node_barrier import NodeBarrier
        nb = NodeBarrier.create(dist, timeout_ms=20000)
        assert nb is not None                                        # both ranks run on this host
        log = np.lib.format.open_memmap(os.path.join(outdir, "log.npy"), mode="r+")
        for k in range(200):
            log[rank, k] = k + 1                                     # "I have arrived at barrier k"
            if rank == 1 and k % 50 == 0:
                time.sleep(0.002)                                    # a straggler
            nb.wait()
            assert log[1 - rank, k] == k + 1, (rank, k)              # ... and so has the other rank, before I leave it
        def region_overhead(bar):
            over = []
            for _ in range(50):
                bar(); t0 = time.perf_counter()
                t_end = t0 + 0.001
                while time.perf_counter() < t_end:
                    pass
                bar(); over.append(time.perf_counter() - t0 - 0.001)
            return np.array(over)
        np.save(os.path.join(outdir, f"over{rank}.npy"), region_overhead(nb.wait))
        np.save(os.path.join(outdir, f"gloo{rank}.npy"), region_overhead(dist.barrier))
        nb.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_node_barrier_between_two_ranks(tmp_path):
    world = 2
    port = 29300 + (os.getpid() % 500)
    np.save(tmp_path / "log.npy", np.zeros((2, 200), np.int64))
    mp.spawn(_node_barrier_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    over = np.concatenate([np.load(tmp_path / f"over{r}.npy") for r in range(world)])
    gloo = np.concatenate([np.load(tmp_path / f"gloo{r}.npy") for r in range(world)])
    # microseconds, not a TCP round trip -- judged against the gloo barrier measured the same way on the same (possibly busy) host
    assert np.median(over) < max(200e-6, 0.5 * np.median(gloo)), (np.median(over), np.median(gloo))
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("crt_bench_")]      # the creating rank unlinked the object


def _node_barrier_other_hosts_worker(rank, world, port, outdir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from clraytracer_amd import node_barrier
        node_barrier._host_id = lambda: f"host-{rank}"              # as if every rank ran on another machine
        nb = node_barrier.NodeBarrier.create(dist, timeout_ms=5000)
        open(os.path.join(outdir, f"nb{rank}.txt"), "w").write("none" if nb is None else "made")
        dist.barrier()                                               # what bench.py keeps using then
    finally:
        dist.destroy_process_group()


def test_node_barrier_is_not_made_across_hosts(tmp_path):
    """Ranks that do not share a host (or a /dev/shm) must ALL fall back to torch.distributed's barrier: a mixed group would deadlock."""
    world = 2
    port = 29400 + (os.getpid() % 500)
    mp.spawn(_node_barrier_other_hosts_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert [open(tmp_path / f"nb{r}.txt").read() for r in range(world)] == ["none", "none"]
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("crt_bench_")]


def test_node_barrier_times_out_instead_of_hanging():
    """Every wait has a deadline: a rank that died cannot hang the others (the barrier returns -1 and the caller gives up loudly)."""
    sys.path.insert(0, ROOT)
    from clraytracer_amd import _lib
    h = _lib.host()
    name = f"/crt_test_timeout_{os.getpid()}".encode()
    a = h.crth_shm_barrier_open(name, 2, 1)
    assert a and h.crth_shm_barrier_open(name, 3, 0) is None          # rank-count mismatch is refused
    assert h.crth_shm_barrier_wait(a, 30) == -1                       # the second rank never comes
    h.crth_shm_barrier_close(a)
    assert h.crth_shm_barrier_open(name, 2, 0) is None                # unlinked
