#!/usr/bin/env python3
"""bench.py -- Mrays/s of the HIP ray-trace path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One *step* = one frame through the hot path: crt_render (RayGen fused into Trace, both bounces) on a
scene that is already resident in HBM. By default three frames are in flight (CRT_RENDER_ASYNC: frames
k+1, k+2 are submitted while frame k runs, each on its own HIP stream and output buffer, so the long-ray
tail of one frame is hidden behind the next; with N >= 4 ranks eight, because a rank's share of the
frame shrinks and its slowest tile does not); all K frames are complete before the
clock stops. Before the W warm-up steps the scene is rendered for --prewarm-ms (untimed; 100 ms) so that the timed steps see the clocks a
renderer runs at, not the ramp of the idle GPU the box hands over (`config.prewarm_ms`).
`--frames-in-flight 1` gives the reference's Render()+clFinish per frame (Renderer.cpp:305-367).
Workload:
  N == 1 : BASELINE config 4 -- `multi-1M` (8 meshes, 1,000,960 triangles, 16 instances, textures),
           1920x1080, primary + one reflection bounce.
  N  > 1 : BASELINE config 5 -- the same scene at 3840x2160, the frame cut into 16-row bands dealt
           round-robin to the ranks (replicated scene, no data-path collective); total work is fixed
           as N grows ("scaling": "strong"). torch.distributed (RCCL) is only used for the barrier
           and the MAX-over-ranks of the wall time.
The N == 1 line also carries `scale_base_n1`: the 3840x2160 frame on this one GPU (outside the timed region), the base
of the N > 1 lines. Rays = primary rays + secondary rays actually traced, counted on the device by an instrumented
launch outside the timed region (and checked against the oracle in tests/).

The JSON line also carries
  roofline     : the contract's HBM roofline of the dominant kernel (crt_trace_kernel): `bound` "hbm", `achieved` = bytes that really
                 leave L2 per launch (`traffic`: FETCH_SIZE + WRITE_SIZE of rocprofv3 --pmc passes of this command -- measured by two short child runs
                 spawned at the very END of this run, when every other number is final and the session is closed (`traffic_live_ok`, `traffic_source`
                 "live: ...", 60 s budget; --no-live-pmc or any failure keeps the committed passes' figure, `traffic_committed_profile`);
                 null when neither exists) / the device time per launch measured
                 live with HIP events on the launch streams, `peak` = 8 TB/s, `frac` = achieved / peak (never printed above 1).
                 SURVEY.md 8d's layout-independent ALGORITHMIC bytes are `algorithmic_*`: they exceed what reaches HBM 30x over
                 (L1/L2/Infinity Cache, the instance cull), so `algorithmic_over_hbm_peak` (> 1) is a work measure, not a
                 fraction of any roofline (`frac_definition` says so in the line). Secondary ceilings from the same PMC passes:
                 `l2` (TCP_TCC_READ_REQ x 64 B / device time against 34.5 TB/s), `l1_hit`, `l2_hit`, `lds_instructions_per_launch`.
                 `chain` = a MODEL, not a bound, and only for the workload it was calibrated on (multi-1M, default size): real
                 (post-cull) child-pair fetches per cycle per CU against the dependent-gather microbenchmark at the kernel's
                 cache-hit mix (tools/ubench/chain.hip, profiles/r*_ubench_chain.json). `valu` / `vmem_pipe`: issue-side accounting.
  with_shadow_rays / config2 / config3 / config3_with_shadow_rays / dense_view / reference_assets / scale_base_n1 : the other BASELINE configs
                 (cornell-1k; sponza-class-250k plain and "primary + 1 shadow ray" as written; the 3840x2160 frame on one GPU), the dense view of the
                 bench scene and upstream's own Sponza + Sibenik assets -- each outside the contract's timed region, each with an `hbm` object
                 (FETCH_SIZE + WRITE_SIZE per launch from the committed rocprofv3 passes of exactly that workload / this run's device time per frame).
  wavefront_compaction / in_wave_refill / in_wave_block_compaction : "wavefront compaction on" (config 4 as written) three ways -- across waves through a
                 queue, and inside the wave (lane refill; phase-separated regrouping) -- each against the default kernel in the same mode.
  many_instances / animated_instances : 401 instances (upstream's limit) with the instance tree vs the linear loop; every instance re-uploaded per frame.
  cpu_baseline : the reference's CPU path timed on this box's host cores (rank 0, N == 1 only): the mirrored
                 CPU_RayCast (CPURayTrace.cpp:186-249, SSE flavour with upstream's rcpps/dpps instruction mix) over the
                 primary rays of the bench frame at 1 thread and at all usable cores, and the scalar Trace oracle
                 (both bounces) on the same frame.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from clraytracer_amd.measure import (HBM_PEAK_GBS, GATHER_CEILING_UNIFORM, GATHER_DIVERGENT_L2, COUNTER_KEYS, aggregate, algorithmic_bytes, chain_ceiling,
                                      cpu_model, hbm_object, live_pmc_traffic, pmc_traffic, pmc_valu, pmc_vmem, secondary_ceilings, usable_cpus)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scene", default="multi-1M")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--band-rows", type=int, default=16)
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="1 = synchronous frames (Render()+clFinish), 2..8 = pipelined; default 3, or 8 when a rank renders 1/4 of the frame or less (N >= 4)")
    ap.add_argument("--shadows", action="store_true", help="extension: one any-hit shadow ray per lit first hit (CRT_RENDER_SHADOWS); not the reference's semantics")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config5", action="store_true", help="skip the 3840x2160 one-GPU point of the N = 1 line (profiling runs: it launches the same kernel at another size)")
    ap.add_argument("--no-extras", action="store_true", help="only the contract's timed region (+ synchronous frames): no config-5 point, no sub-records (A/B runs)")
    ap.add_argument("--timed-region-only", action="store_true", help="profiling aid: after the warm-up run ONLY the contract's timed region (no synchronous leg, no clock-probe frames, "
                                                                     "no extras), so that a kernel trace of the run holds launches of one mode only")
    ap.add_argument("--diag-mix3", action="store_true", help="profiling aid: every step is ONE dispatch tracing the frame three times with interleaved tile lists "
                                                             "(CRT_RENDER_DIAG_MIX3: the wave mix of three frames in flight, visible to a PMC pass); synchronous; rates are per 3 frames")
    ap.add_argument("--prewarm-ms", type=float, default=100.0, help="untimed rendering before the warm-up steps, so that the timed steps do not measure the clock ramp of an idle GPU")
    ap.add_argument("--live-pmc", action="store_true", help="run the live PMC passes even with --no-extras (tests)")
    ap.add_argument("--no-live-pmc", action="store_true", help="take roofline.traffic from the committed profile instead of measuring it now with two short rocprofv3 --pmc child runs")
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()
    if args.diag_mix3:
        args.frames_in_flight, args.no_extras, args.no_cpu_baseline = 1, True, True
    if args.timed_region_only:
        args.no_extras, args.no_cpu_baseline = True, True
    if args.no_extras:
        args.no_config5 = True

    # stdout carries exactly ONE line (the JSON); anything native libraries print to fd 1 on the way (RCCL's banner)
    # goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n = args.gpus
    # --gpus N>1 WITHOUT a launcher: all N GPUs are driven from this one process through crt_init_devices (replicated scene,
    # 16-row bands per device, peer-copy gather into GPU 0's frame: every timed frame ends as a whole frame on one device).
    inproc = world == 1 and n > 1
    if world != n and not inproc:
        n = world

    # A rank's share of the frame shrinks with N but its slowest tiles do not: a frame on a slot lasts at least as long as its
    # slowest wave (0.3-0.5 ms on multi-1M), so the frames a slot can deliver per second are capped and a small share needs
    # more slots to keep the GPU full. One GPU: 3 vs 4 vs 6 slots measure the same.
    flight = args.frames_in_flight if args.frames_in_flight > 0 else (3 if n < 4 else 8)
    flight = max(1, min(8, flight))
    os.environ["CRT_FRAMES_IN_FLIGHT"] = str(flight)   # read by crt_init
    # One hardware queue per slot's stream; read by the HIP runtime at its first call. Always (r6), not only for more than four slots: with the
    # runtime's default of four queues two of the THREE slot streams occasionally land on one queue (other streams of the process take queues
    # too) and their frames serialise -- one line of round 6's evidence run read nanosuit-demo at 12.6 instead of 16.6 Gray/s that way, which is
    # exactly what GPU_MAX_HW_QUEUES=2 reproduces (profiles/r06_hw_queues.txt); eight queues measure the same as four when nothing collides.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    import ctypes as C
    import numpy as np
    import torch
    from clraytracer_amd import _lib, driver, scenes

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ray-trace path has no CPU fallback")
    # CRT_BENCH_REHEARSE=1: every rank shares GPU 0 and the control plane runs over gloo -- a functional rehearsal of the
    # N>1 path on a one-GPU box (numbers are meaningless: the ranks time-share one device)
    rehearse = os.environ.get("CRT_BENCH_REHEARSE") == "1"
    device_index = 0 if rehearse else local_rank
    torch.cuda.set_device(device_index)
    red_device = "cpu"
    dist = None
    ctl = None              # process group for barriers/reductions (None = the default gloo group)
    control_plane = None
    rccl_seen = None        # what the RCCL probe's all-reduce of 1 over the ranks returned on this rank (= the number of ranks RCCL really connected)
    if n > 1 or os.environ.get("CRT_BENCH_FORCE_DIST") == "1":   # FORCE_DIST: exercise the RCCL control plane with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                                  # CRT_BENCH_FORCE_DIST without a launcher
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_PORT", "29533")):
                os.environ.setdefault(k, v)
        # The control plane is ONE process group created the same way on every rank: gloo over TCP (always available). RCCL
        # only ever carried a barrier and two tiny reductions here (no pixel data crosses GPUs), so whether it comes up is
        # decided COLLECTIVELY: every rank tries to create an nccl (= RCCL) sub-group and to reduce one value through it,
        # the outcomes are MIN-reduced over gloo, and only if all ranks succeeded do barriers/reductions use RCCL. (Round
        # 1 fell back per rank inside an `except`, which deadlocks when only some ranks fail.)
        dist.init_process_group(backend="gloo")
        ctl = None
        if not rehearse and os.environ.get("CRT_BENCH_BACKEND", "nccl") == "nccl":
            ok = 1
            try:
                ctl = dist.new_group(backend="nccl")
                probe = torch.ones(1, device="cuda")
                dist.all_reduce(probe, group=ctl)
                torch.cuda.synchronize()
                rccl_seen = int(probe.item())
                ok = int(rccl_seen == world)
            except Exception as e:  # pragma: no cover - depends on the node
                sys.stderr.write(f"[bench] rank {rank}: RCCL group unavailable ({e})\n")
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                ctl = None
        red_device = "cuda" if ctl is not None else "cpu"
        control_plane = "rccl" if ctl is not None else "gloo"

    inproc_devices = None
    if inproc:
        have = torch.cuda.device_count()
        if rehearse:
            inproc_devices = [0] * n
        elif have >= n:
            inproc_devices = list(range(n))
        else:
            raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible (CRT_BENCH_REHEARSE=1 lists GPU 0 {n} times: functional rehearsal)")
    width = args.width or (1920 if n == 1 else 3840)
    height = args.height or (1080 if n == 1 else 2160)
    # scene files are generated once (rank 0) into the shared cache directory; with several ranks rank 0 also imports the
    # meshes once on the host so that their .clm caches exist (AssetManager.cpp:363-381: ImportMesh prefers <stem>.clm) --
    # N ranks then read N caches instead of running N OBJ parses at once, and build their BVHs on their GPUs (crt_build_bvh)
    if rank == 0:
        sc = scenes.get(args.scene)
        if n > 1:
            with driver.Session(64, 64, host_only=True) as pre:
                pre.load_scene(sc)
    if dist is not None:
        dist.barrier(group=ctl)
    if rank != 0:
        sc = scenes.get(args.scene)

    single = None
    if inproc:
        # the same workload on ONE GPU first (own session: the library drives one session at a time)
        with driver.Session(width, height, device=inproc_devices[0]) as s1:
            s1.load_scene(sc)
            a1, iv1, ip1 = s1.trace_args()
            fp1 = C.POINTER(C.c_float)
            hip1 = _lib.hip()
            fl1 = (4 if flight > 1 else 0) | (32 if args.shadows else 0)
            for _ in range(3):
                _lib.check(hip1.crt_render(C.byref(a1), iv1.ctypes.data_as(fp1), ip1.ctypes.data_as(fp1), fl1), "crt_render")
            _lib.check(hip1.crt_sync(), "crt_sync")
            t0 = time.perf_counter()
            for _ in range(10):
                hip1.crt_render(C.byref(a1), iv1.ctypes.data_as(fp1), ip1.ctypes.data_as(fp1), fl1)
            _lib.check(hip1.crt_sync(), "crt_sync")
            single = (time.perf_counter() - t0) / 10
    t_load = time.time()
    s = driver.Session(width, height, device=device_index, devices=inproc_devices)
    s.load_scene(sc, device_bvh_build=(n > 1))
    if not inproc:
        s.set_row_bands(args.band_rows, rank, n)
    t_load = time.time() - t_load
    if dist is not None:                       # the slowest rank's load is the job's
        tl = torch.tensor([t_load], dtype=torch.float64, device=red_device)
        dist.all_reduce(tl, op=dist.ReduceOp.MAX, group=ctl)
        t_load = float(tl.item())

    # instrumented launch (untimed): rays and work counters of this rank's share of the frame
    s.render_raw(8 | (32 if args.shadows else 0))
    cnt = s.counters()
    culled = C.c_uint64()
    _lib.check(_lib.hip().crt_get_culled_visits(C.byref(culled)), "crt_get_culled_visits")
    pair_fetches = cnt["innerVisits"] - int(culled.value)        # child-pair records the device really fetches per frame
    own_rows = s.owned_rows()

    # The barrier that brackets the timed regions. All ranks of the contract run on one node, so they meet in shared memory
    # (clraytracer_amd/node_barrier.py, host/ShmBarrier.cpp): a rank leaves within a cache-line transfer of the last arrival, where a gloo /
    # RCCL barrier's latency and exit skew (0.1-0.3 ms idle, milliseconds on a loaded host: tools/barrier_cost.py) would be counted as rendering
    # time of a 2.6 ms region (N = 8: 20 x 0.13 ms). Ranks on different hosts, or CRT_BENCH_SHM_BARRIER=0, keep torch.distributed's barrier.
    node_barrier = None
    if dist is not None and world > 1 and os.environ.get("CRT_BENCH_SHM_BARRIER", "1") != "0":
        from clraytracer_amd.node_barrier import NodeBarrier
        node_barrier = NodeBarrier.create(dist)

    def barrier():
        torch.cuda.synchronize()
        if node_barrier is not None:
            node_barrier.wait()
        elif dist is not None:
            dist.barrier(group=ctl)
        torch.cuda.synchronize()

    # static camera: build the C arguments once so the timed loop is one C call per frame
    targs, iv, ip = s.trace_args()
    fp = C.POINTER(C.c_float)
    p_args, p_iv, p_ip = C.byref(targs), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp)
    hip = _lib.hip()
    crt_render = hip.crt_render
    flags = (4 if flight > 1 else 0) | (32 if args.shadows else 0) | (1024 if args.diag_mix3 else 0)   # CRT_RENDER_ASYNC, CRT_RENDER_SHADOWS, CRT_RENDER_DIAG_MIX3
    stats = _lib.CrtFrameStats()

    # Clock pre-warm (untimed, before the W warm-up steps): the box hands over an idle GPU at its idle clocks, and W = 5 frames
    # (1.5 ms) do not bring them up -- a 20-step timed region right behind them measured the DVFS ramp (5.85 ms of device time
    # against 5.45 ms for the same burst repeated). Render for --prewarm-ms so the timed steps see the clocks a renderer runs at.
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
        for _ in range(32):
            _lib.check(crt_render(p_args, p_iv, p_ip, flags), "crt_render")
        _lib.check(hip.crt_sync(), "crt_sync")
    for _ in range(args.warmup):
        _lib.check(crt_render(p_args, p_iv, p_ip, flags), "crt_render")
    _lib.check(hip.crt_sync(), "crt_sync")
    _lib.check(hip.crt_frame_time_stats(None, 1), "crt_frame_time_stats")
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rc = crt_render(p_args, p_iv, p_ip, flags)   # flight == 1: returns when the frame is complete
    rc2 = hip.crt_sync()                             # every frame finished
    barrier()
    elapsed = time.perf_counter() - t0
    _lib.check(rc, "crt_render")
    _lib.check(rc2, "crt_sync")
    # HIP events on the launch streams, read back after the clock stopped: per-launch Trace durations (these overlap
    # when frames are in flight) and the device-time extent first start -> last end
    _lib.check(hip.crt_frame_time_stats(C.byref(stats), 0), "crt_frame_time_stats")
    launch_ms = stats.sumMs[2] / max(1, stats.frames)
    extent_ms = stats.extentMs / max(1, stats.frames)
    # steady state: the first frame's latency is the pipeline's fill time; what follows it is K - 1 frames at the steady cadence.
    # A 20-step run and a 200-step run agree on this figure; the contract's `value` includes the fill.
    steady_ms = (stats.extentMs - stats.firstFrameMs) / max(1, stats.frames - 1) if stats.frames > 1 else extent_ms

    # N > 1: the same workload on ONE GPU -- rank 0 renders the whole frame, untimed by the contract clock, while the others wait at the
    # next barrier -- so the line carries its own strong-scaling reference. Like for like (r6): the same region shape as the timed one
    # (W warm-up frames, K frames + sync, fill included) on clocks the timed region has just warmed; rounds 1-5 timed 3 + 10 frames
    # BEFORE the pre-warm, which under-read the single GPU (11.1-11.5 vs 12.3 Gray/s) and so flattered the speed-up by 7-10 %.
    if n > 1 and rank == 0 and not inproc:
        s.set_row_bands(args.band_rows, 0, 1)
        for _ in range(max(args.warmup, 3)):
            _lib.check(crt_render(p_args, p_iv, p_ip, flags), "crt_render")
        _lib.check(hip.crt_sync(), "crt_sync")
        t0 = time.perf_counter()
        for _ in range(args.steps):
            rc = crt_render(p_args, p_iv, p_ip, flags)
        _lib.check(hip.crt_sync(), "crt_sync")
        single = (time.perf_counter() - t0) / args.steps
        s.set_row_bands(args.band_rows, rank, n)
        _lib.check(hip.crt_frame_time_stats(None, 1), "crt_frame_time_stats")

    def measure_view(sess, vflags, frames, label):
        """An extra view / flag set outside the contract's timed region: counted launch for the rays, then `frames` frames."""
        sess.render_raw(8 | (vflags & 32))
        c = sess.counters()
        a2, iv2, ip2 = sess.trace_args()
        q = (C.byref(a2), iv2.ctypes.data_as(fp), ip2.ctypes.data_as(fp))
        for _ in range(5):
            _lib.check(crt_render(*q, vflags), "crt_render")
        _lib.check(hip.crt_sync(), "crt_sync")
        _lib.check(hip.crt_frame_time_stats(None, 1), "crt_frame_time_stats")
        t0 = time.perf_counter()
        for _ in range(frames):
            r = crt_render(*q, vflags)
        _lib.check(hip.crt_sync(), "crt_sync")
        dt = (time.perf_counter() - t0) / frames
        _lib.check(r, "crt_render")
        st = _lib.CrtFrameStats()
        _lib.check(hip.crt_frame_time_stats(C.byref(st), 0), "crt_frame_time_stats")
        steady = (st.extentMs - st.firstFrameMs) / max(1, st.frames - 1) if st.frames > 1 else dt * 1e3
        return {"value": round(c["rays"] / dt / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(dt * 1e3, 4), "frames": frames,
                "rays_per_frame": c["rays"], "shadow_rays_per_frame": c["shadowRays"],
                "primary_hit_fraction": round(c["secondary"] / max(1, c["primary"]), 4),
                "inner_visits_per_ray": round(c["innerVisits"] / max(1, c["rays"]), 2), "tri_tests_per_ray": round(c["triTests"] / max(1, c["rays"]), 2),
                "steady_state": {"value": round(c["rays"] / (steady * 1e-3) / 1e6, 2), "ms_per_step": round(steady, 4)},
                "workload": label}

    # the same K frames the reference's way -- one at a time, Render() + clFinish (Renderer.cpp:305-367) -- reported
    # next to the headline as `synchronous_frames` (not part of the contract's timed region above)
    sync_elapsed = None
    if flight > 1 and not args.timed_region_only:
        sflags = flags & ~4
        for _ in range(min(args.warmup, 3)):
            _lib.check(crt_render(p_args, p_iv, p_ip, sflags), "crt_render")
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            rc = crt_render(p_args, p_iv, p_ip, sflags)
        barrier()
        sync_elapsed = time.perf_counter() - t0
        _lib.check(rc, "crt_render")

    # the shader clock the device holds while such frames are in flight: one probe wave per XCD watches s_memtime against the
    # 100 MHz s_memrealtime for 300 us on its own stream while 24 frames run (cycle-based figures below use this, not 2.4 GHz)
    clock_meas = None
    if rank == 0 and not args.timed_region_only:
        for _ in range(24):
            crt_render(p_args, p_iv, p_ip, flags)
        ghz = C.c_double(0.0)
        if hip.crt_debug_measure_clock(300, C.byref(ghz)) == 0 and ghz.value > 0.5:
            clock_meas = float(ghz.value)
        _lib.check(hip.crt_sync(), "crt_sync")
    if dist is not None:
        dist.barrier(group=ctl)

    # N > 1: the same K frames with every rank's bands DELIVERED per frame -- copied to pinned host memory behind the
    # frame's kernels (CRT_RENDER_READBACK moves only the rows a rank owns), the last copy complete before the clock
    # stops. The headline above leaves the tiles in each rank's HBM; this is the rate at which a whole frame reaches
    # one place (host memory of the node) that a consumer can read.
    deliver_elapsed = deliver8_elapsed = None
    if n > 1 and not inproc:
        def delivered(dflags):
            ptr, nbytes = C.c_void_p(), C.c_size_t()
            for _ in range(min(args.warmup, 3)):
                _lib.check(crt_render(p_args, p_iv, p_ip, dflags), "crt_render")
            _lib.check(hip.crt_map_host_frame(C.byref(ptr), C.byref(nbytes)), "crt_map_host_frame")
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                rc = crt_render(p_args, p_iv, p_ip, dflags)
            rc2 = hip.crt_map_host_frame(C.byref(ptr), C.byref(nbytes))      # waits for the last frame's copy
            rc3 = hip.crt_sync()
            barrier()
            dt = time.perf_counter() - t0
            _lib.check(rc, "crt_render"); _lib.check(rc2, "crt_map_host_frame"); _lib.check(rc3, "crt_sync")
            return dt
        deliver_elapsed = delivered(flags | 128)                             # float4 HDR: 16 B per pixel over each rank's PCIe link
        # ... and as what upstream displays: the frame through its RGBA8 render target (CRT_RENDER_UNORM8: the pixels are
        # quantised, hazard H8), 4 B per pixel
        deliver8_elapsed = delivered(flags | 128 | 64)

    # one process driving N GPUs: what the gather moves per frame, and the same K frames through upstream's RGBA8 render target
    # (CRT_RENDER_UNORM8: the secondaries send 4 B per pixel -- the bytes their Trace epilogue stores -- instead of float4 bands)
    inproc_gather = None
    if inproc:
        gb, bpp = s.last_gather()
        for _ in range(min(args.warmup, 3)):
            _lib.check(crt_render(p_args, p_iv, p_ip, flags | 64), "crt_render")
        _lib.check(hip.crt_sync(), "crt_sync")
        t0 = time.perf_counter()
        for _ in range(args.steps):
            rc = crt_render(p_args, p_iv, p_ip, flags | 64)
        rc2 = hip.crt_sync()
        dt8 = time.perf_counter() - t0
        _lib.check(rc, "crt_render"); _lib.check(rc2, "crt_sync")
        gb8, bpp8 = s.last_gather()
        inproc_gather = {"gather_bytes_per_frame": gb, "bytes_per_pixel": bpp, "path": hip.crt_gather_path().decode(),
                         "rgba8_frames": {"gather_bytes_per_frame": gb8, "bytes_per_pixel": bpp8, "ms_per_step": round(dt8 * 1e3 / args.steps, 4),
                                          "note": "the same K frames with CRT_RENDER_UNORM8 (upstream's RGBA8 render target, Renderer.cpp:63,192): the bytes are gathered, "
                                                  "the float frame is rebuilt on the first device only when it is read"}}

    tot, elapsed_max, kernel_ms_max = aggregate(dist, cnt, own_rows * width, elapsed, extent_ms, red_device, ctl)
    if sync_elapsed is not None:
        _, sync_elapsed, _ = aggregate(dist, cnt, own_rows * width, sync_elapsed, 0.0, red_device, ctl)
    if deliver_elapsed is not None:
        _, deliver_elapsed, _ = aggregate(dist, cnt, own_rows * width, deliver_elapsed, 0.0, red_device, ctl)
        _, deliver8_elapsed, _ = aggregate(dist, cnt, own_rows * width, deliver8_elapsed, 0.0, red_device, ctl)

    staggered_frames = 0
    if rank == 0:
        v_ = C.c_uint64()
        if hip.crt_debug_staggered_frames(C.byref(v_)) == 0:
            staggered_frames = int(v_.value)
    # per-rank times of the timed region (N > 1): who was slowest, and by how much
    rank_ms = [elapsed * 1e3 / args.steps]
    if dist is not None and not inproc:
        tv = torch.zeros(n, dtype=torch.float64, device=red_device)
        tv[rank] = elapsed * 1e3 / args.steps
        dist.all_reduce(tv, op=dist.ReduceOp.SUM, group=ctl)
        rank_ms = [float(x) for x in tv.tolist()]
    if rank == 0:
        rays_per_frame = tot["rays"]
        ms_per_step = elapsed_max * 1e3 / args.steps
        value = rays_per_frame * args.steps / elapsed_max / 1e6
        # ---- roofline of the dominant kernel on this rank (per launch = this rank's share of one frame) ----
        # device time per launch = extent of the timed region on the launch streams / K (HIP events). With one frame in
        # flight that is the launch duration; with frames in flight launches overlap, each lasts longer
        # (launch_duration_ms, what a kernel trace shows) and shares the machine with the others.
        my_bytes = algorithmic_bytes(cnt, own_rows * width)
        dev_s = extent_ms * 1e-3
        traffic, traffic_src = pmc_traffic("crt_trace_kernel", sc.name, width, height, args.shadows) if n == 1 else (None, None)
        traffic_committed = traffic
        # (the contract's `traffic` is re-measured by THIS run at the very end -- live_pmc_traffic below, after every other number is final and
        # the session is closed; until then, and if that fails, the committed profile's figure stands)
        want_live = n == 1 and not args.shadows and not args.diag_mix3 and not args.no_live_pmc and (args.live_pmc or not args.no_extras)
        hbm_achieved = None if traffic is None else traffic / dev_s / 1e9
        hbm_frac = None if hbm_achieved is None else hbm_achieved / HBM_PEAK_GBS
        num_cus = int(hip.crt_device_name().decode().split(",")[-1].split()[0])
        # shader clock: measured beside frames in flight; CRT_SCLK_GHZ overrides; the 2.4 GHz nominal clock only as a last resort
        # (r6: a run that skips the probe -- --timed-region-only, i.e. the kernel-trace passes under rocprofv3, where kernels are serialised and a probe
        # "beside frames in flight" would measure nothing -- takes the clock the newest committed UN-profiled bench line measured, not the nominal one)
        committed_clock = None
        if clock_meas is None and "CRT_SCLK_GHZ" not in os.environ:
            import glob
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_line.json")), reverse=True):
                try:
                    ch = (json.load(open(path)).get("roofline") or {}).get("chain") or {}
                    if str(ch.get("clock_source", "")).startswith("crt_debug_measure_clock") and ch.get("clock_ghz"):
                        committed_clock = (float(ch["clock_ghz"]), os.path.relpath(path, ROOT))
                        break
                except Exception:
                    continue
        clock_ghz = float(os.environ["CRT_SCLK_GHZ"]) if "CRT_SCLK_GHZ" in os.environ else (clock_meas or (committed_clock[0] if committed_clock else 2.4))
        clock_src = "CRT_SCLK_GHZ" if "CRT_SCLK_GHZ" in os.environ else (
            "crt_debug_measure_clock: s_memtime / s_memrealtime beside 24 frames in flight" if clock_meas else (
                f"not measured in this run ({'--timed-region-only: probe skipped' if args.timed_region_only else 'probe failed'}); taken from the un-profiled line {committed_clock[1]}"
                if committed_clock else f"NOMINAL 2.4 GHz ({'--timed-region-only: probe skipped' if args.timed_region_only else 'probe failed'}, no committed line to take it from): cycle-based figures are approximate"))
        ndev_here = n if inproc else 1                                          # in-process: counters are summed over the devices
        gather_rate = pair_fetches / ndev_here / (dev_s * clock_ghz * 1e9 * num_cus)   # 64-B records per cycle per CU
        warnings = []
        roofline_error = None
        if hbm_frac is not None and hbm_frac > 1.0:
            # a true bound cannot be exceeded: the committed profile does not belong to this build / workload -> no fraction is printed
            roofline_error = f"measured HBM traffic {traffic} B per launch over {dev_s * 1e3:.4f} ms = {hbm_achieved:.0f} GB/s exceeds the {HBM_PEAK_GBS:.0f} GB/s peak: stale profile ({traffic_src})"
            warnings.append(roofline_error)
            hbm_achieved = hbm_frac = None
        second = secondary_ceilings("crt_trace_kernel", sc.name, width, height, dev_s) if (n == 1 and not args.shadows) else None
        # the chain microbenchmark's table reproduces multi-1M's cache-hit mix at the default frame size: a model for that workload only
        chain_applies = n == 1 and sc.name == "multi-1M" and (width, height) == (1920, 1080) and not args.shadows and not args.diag_mix3
        ceil = chain_ceiling() if chain_applies else None
        chain = None
        if ceil:
            chain = {"kind": "model: a microbenchmark's rate at this workload's cache-hit mix, not a bound (coherent packets can exceed it)",
                     "achieved": round(gather_rate, 4), "ceiling": round(ceil["full"], 4), "frac": round(gather_rate / ceil["full"], 4),
                     "unit": "64-B records per cycle per CU", "calibrated_for": "multi-1M 1920x1080, primary + reflection bounce",
                     "ceiling_at_28_of_64_lanes": None if ceil["lanes28"] is None else round(ceil["lanes28"], 4),
                     "frac_of_ceiling_at_28_lanes": None if ceil["lanes28"] is None else round(gather_rate / ceil["lanes28"], 4),
                     "ceiling_source": ceil["source"], "clock_ghz": round(clock_ghz, 4), "clock_source": clock_src, "cus": num_cus,
                     "note": "ceiling = tools/ubench/chain.hip: every lane of 8 waves/SIMD chasing its own chain of 64-B records (4 x dwordx4 + the inner step's "
                             "arithmetic + LDS push per hop) at the trace kernel's measured hit mix (" + str(ceil.get("mix")) + ": %.0f %% of records L1-resident, %.0f %% from L2, the rest from the "
                             "Infinity Cache); the same rate from 2 to 8 waves/SIMD" % (100.0 * (ceil.get("hot") or 0.0), 100.0 * (ceil.get("warm") or 0.0)) + ", i.e. a throughput limit of the CU's vector-memory path, not latency. The "
                             "trace kernel runs with ~28 of 64 lanes working per vector instruction (ceiling_at_28_of_64_lanes) and its packets are partly "
                             "coherent (several lanes per record), which is how it can sit above that second figure"}
        out = {
            "metric": "Mrays/s (primary + traced secondary rays), ms/frame at 1920x1080" if n == 1 else "Mrays/s (primary + traced secondary rays), 3840x2160 tiled over N GPUs",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{sc.name}: {sc.num_tris} triangles, {len(sc.meshes)} meshes, {len(sc.instances)} instances, "
                                   f"{width}x{height}, primary + 1 reflection bounce" + (" + 1 shadow ray per lit first hit (extension)" if args.shadows else "") + ", RayGen fused",
                       "scene": sc.name, "width": width, "height": height, "rays_per_frame": int(rays_per_frame),
                       "primary": int(tot["primary"]), "secondary": int(tot["secondary"]), "shadow": int(tot["shadowRays"]),
                       "primary_hit_fraction": round(tot["secondary"] / max(1.0, tot["primary"]), 4),
                       "tiling": (f"16-row bands round-robin over {n} devices driven by ONE process (crt_init_devices {inproc_devices}), replicated scene, "
                                  "every frame gathered into device 0 by peer copies inside the timed region") if inproc
                                 else f"{args.band_rows}-row bands round-robin over {n} rank(s), replicated scene",
                       "frames_in_flight": flight, "control_plane": control_plane,
                       "barrier": None if dist is None else ("shared-memory node barrier (host/ShmBarrier.cpp) between torch.cuda.synchronize() calls; group set-up and reductions over " + str(control_plane)
                                                            if node_barrier is not None else "torch.distributed barrier over " + str(control_plane)), "diag_mix3": bool(args.diag_mix3), "prewarm_ms": args.prewarm_ms,
                       "stagger": {"mode": ("CRT_STAGGER_US=" + os.environ["CRT_STAGGER_US"]) if "CRT_STAGGER_US" in os.environ
                                           else "automatic: the first frame of slots 1.. of a burst that follows a burst longer than the slot count is held back by slot x latency / slots (up to 3 slots)",
                                   "frames_held_back_in_session": staggered_frames},
                       "n_gt_1_lines": "lines with n_gpus > 1 render BASELINE config 5 (3840x2160), not this 1920x1080 frame: their base is `scale_base_n1` here / "
                                       "`single_gpu_same_workload` there, never this line's `value`" if n == 1 else None,
                       "device": hip.crt_device_name().decode(), "scene_load_s": round(t_load, 2),
                       "scene_load": ("max over ranks; meshes from the .clm caches rank 0 wrote once, BVH built on each GPU (crt_build_bvh)" if n > 1
                                      else "OBJ import (or .clm cache) + host SAH build + uploads")},
            "steady_state": {"value": round(rays_per_frame / (steady_ms * 1e-3) / 1e6, 2) if (n == 1 or inproc) else None,
                             "unit": "Mrays/s", "ms_per_step": round(steady_ms, 4), "first_frame_ms": round(stats.firstFrameMs, 4),
                             "note": "(device extent of the timed region - the first frame's latency) / (K - 1): the rate behind the pipeline's fill, which `value` "
                                     "includes -- a 20-step and a 200-step run agree on this figure (rank 0's device)"},
            "inner_visits_per_s": round(tot["innerVisits"] * args.steps / elapsed_max, 0),
            "tri_tests_per_s": round(tot["triTests"] * args.steps / elapsed_max, 0),
            "kernel_ms": {"crt_trace_kernel_launch_mean": round(launch_ms, 4), "device_time_per_frame": round(extent_ms, 4),
                          "frame_latency_mean": round(stats.sumMs[0] / max(1, stats.frames), 4),
                          "device_time_per_frame_max_over_ranks": round(kernel_ms_max, 4)},
            "roofline": {"bound": "hbm",
                         "achieved": None if hbm_achieved is None else round(hbm_achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": None if hbm_frac is None else round(hbm_frac, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_committed_profile": traffic_committed, "traffic_live_ok": False, "traffic_live_note": None if want_live else "not attempted (--no-live-pmc / --no-extras / --shadows / N > 1)",
                         "traffic_live_child_rc": [],
                         "frac_definition": "achieved / peak with achieved = `traffic` (FETCH_SIZE + WRITE_SIZE bytes per launch from rocprofv3 --pmc passes of this command -- measured by "
                                            "two child runs of this very run when traffic_source says `live`, else the committed profile: every byte that leaves L2, Infinity-Cache hits included, so an upper bound on HBM bytes) / device_time_per_launch_ms "
                                            "(HIP events on the launch streams, this run). HBM is NOT what binds this kernel (dependent 64-B gathers through the CU's "
                                            "vector-memory path are: see `chain`, a model); SURVEY 8d's algorithmic bytes are a work measure (algorithmic_over_hbm_peak > 1)",
                         "error": roofline_error,
                         "hbm_frac": None if hbm_frac is None else round(hbm_frac, 4),
                         "l2": None if not second else second.get("l2"), "l1_hit": None if not second else second.get("l1_hit"),
                         "l2_hit": None if not second else second.get("l2_hit"),
                         "lds_instructions_per_launch": None if not second else second.get("lds_instructions_per_launch"),
                         "secondary_source": None if not second else second.get("source"),
                         "chain": chain,
                         "kernel": "crt_trace_kernel<COUNT=false, STAMP=false, SHADOW=%s, TLAS=false, REFRACT=false>" % ("true" if args.shadows else "false"),
                         "launch_duration_ms": round(launch_ms, 4), "launches_in_flight": flight, "device_time_per_launch_ms": round(extent_ms, 4),
                         "algorithmic_bytes_per_launch": int(my_bytes),
                         "algorithmic_rate_gbs": round(my_bytes / dev_s / 1e9, 2),
                         "algorithmic_over_hbm_peak": round(my_bytes / dev_s / 1e9 / HBM_PEAK_GBS, 3),
                         "algorithmic_over_traffic": None if traffic is None else round(my_bytes / traffic, 1),
                         "algorithmic_note": "SURVEY 8d bytes from reference struct sizes; served from L1/L2/Infinity Cache and removed by the instance cull, "
                                             "hence far above the HBM traffic: algorithmic_over_hbm_peak is a work measure, NOT a roofline fraction",
                         "bytes_per_ray": round(my_bytes / max(1, cnt["rays"]), 1),
                         "inner_visits_per_ray": round(cnt["innerVisits"] / max(1, cnt["rays"]), 2),
                         "tri_tests_per_ray": round(cnt["triTests"] / max(1, cnt["rays"]), 2),
                         "gather": {"pair_fetches_per_launch": int(pair_fetches), "culled_root_visits_per_launch": int(culled.value),
                                    "records_per_cycle_per_cu": round(gather_rate, 4),
                                    "uniform_l1_reference": round(GATHER_CEILING_UNIFORM, 4), "divergent_l2_reference": round(GATHER_DIVERGENT_L2, 4),
                                    "note": "reference points from tools/ubench/gather.hip (independent fetches): every lane on one L1-resident record / every lane on a different L2-resident record"},
                         "vmem_pipe": pmc_vmem("crt_trace_kernel", sc.name, width, height, dev_s, clock_ghz, num_cus) if (n == 1 and not args.shadows) else None,
                         "valu": pmc_valu("crt_trace_kernel", sc.name, width, height, dev_s, clock_ghz, num_cus) if (n == 1 and not args.shadows) else None},
        }
        if warnings:
            out["warnings"] = warnings
        if n > 1:
            out["config"]["rccl_ranks_seen"] = rccl_seen      # N ranks behind one RCCL communicator (None: gloo control plane / one process)
            out["config"]["gather_path"] = hip.crt_gather_path().decode() if inproc else "none: every rank keeps its bands (delivered_to_host* copy them to pinned host memory)"
            if inproc:
                out["config"]["peer_access"] = [int(hip.crt_peer_access(d)) for d in range(n)]
                ig = inproc_gather
                ig["rgba8_frames"]["value"] = round(rays_per_frame * args.steps / (ig["rgba8_frames"]["ms_per_step"] * 1e-3 * args.steps) / 1e6, 2)
                out["inprocess_gather"] = ig
        if sync_elapsed is not None:
            out["synchronous_frames"] = {"value": round(rays_per_frame * args.steps / sync_elapsed / 1e6, 2), "unit": "Mrays/s",
                                         "ms_per_step": round(sync_elapsed * 1e3 / args.steps, 4),
                                         "note": "same K frames one at a time (the reference's Render() + clFinish), max over ranks"}
        if deliver_elapsed is not None:
            out["delivered_to_host"] = {"value": round(rays_per_frame * args.steps / deliver_elapsed / 1e6, 2), "unit": "Mrays/s",
                                        "ms_per_step": round(deliver_elapsed * 1e3 / args.steps, 4),
                                        "note": "same K frames, every rank's bands copied to pinned host memory per frame inside the timed region (float4), max over ranks"}
            out["delivered_to_host_rgba8"] = {"value": round(rays_per_frame * args.steps / deliver8_elapsed / 1e6, 2), "unit": "Mrays/s",
                                              "ms_per_step": round(deliver8_elapsed * 1e3 / args.steps, 4),
                                              "note": "the same with the frame taken through upstream's RGBA8 render target (CRT_RENDER_UNORM8): 4 B per pixel instead of 16"}
        if single is not None:
            out["single_gpu_same_workload"] = {"value": round(rays_per_frame / single / 1e6, 2), "unit": "Mrays/s",
                                               "ms_per_step": round(single * 1e3, 4),
                                               "note": "rank 0 alone rendering the whole frame right after the timed region: the same region shape (W warm-up frames, K frames + sync, same frames in flight)"
                                                       if not inproc else "one device alone rendering the whole frame, 3 + 10 frames, same mode"}
            out["speedup_vs_single_gpu_same_workload"] = round(value / (rays_per_frame / single / 1e6), 3)
        if n > 1:
            out["per_rank_ms_per_step"] = [round(x, 4) for x in rank_ms]
            out["imbalance_max_over_mean"] = round(max(rank_ms) / (sum(rank_ms) / len(rank_ms)), 4) if not inproc else None
            out["scaling_note"] = ("speedup_vs_single_gpu_same_workload = value / single_gpu_same_workload.value (the same 3840x2160 frame on one GPU, measured in this run); "
                                   "do not divide by the n_gpus = 1 line's value, which is the 1920x1080 frame. No 1 -> 8 GPU curve had been measured on hardware when this was written."
                                   + (" REHEARSAL: every rank shares GPU 0, the numbers are meaningless." if rehearse else ""))
        # Everything outside the contract's timed region -- the other BASELINE configs and views, the kernel forms, instances, the device BuildBVH, the
        # CPU baseline -- lives in clraytracer_amd/bench_extras.py (r6); it may close and reopen the session and hands back the one that is open.
        from types import SimpleNamespace
        from clraytracer_amd import bench_extras
        s = bench_extras.run(SimpleNamespace(args=args, cnt=cnt, crt_render=crt_render, device_index=device_index, flags=flags, fp=fp, hip=hip, measure_view=measure_view,
                                             n=n, out=out, rays_per_frame=rays_per_frame, sc=sc, s=s, width=width, height=height))
        # the CPU baseline: the one leg of this file that loads the oracle (as the checker's timing twin, never as the thing measured)
        if n == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib
            threads = args.cpu_threads or min(usable_cpus(), 64)
            iv, ip, pos = s.camera()
            orc = oracle_lib.Oracle(s.arenas(), nthreads=threads)
            rays = orc.raygen(width, height, iv, ip)
            # (1) the reference's CPU path: CPU_RayCast (CPURayTrace.cpp:186-249) once per pixel of the bench frame -- one
            # primary ray, closest hit + albedo, no lighting, no bounce. SSE flavour = upstream's rcpps/dpps instruction mix.
            flat = np.ascontiguousarray(rays.reshape(-1, 3))
            origins = np.ascontiguousarray(np.tile(np.asarray(pos, np.float32), (len(flat), 1)))
            # bounded sample, about 10-20 s of CPU work in all: the whole frame once on one thread, the frame repeated on all
            # usable cores for ~0.5 s of wall time, one frame of the scalar Trace oracle
            sub = slice(0, len(flat), 1)
            t0 = time.perf_counter(); s.cpu_raycast(origins[sub], flat[sub], nthreads=1, sse=True); dt1 = time.perf_counter() - t0
            dtn, rec, reps, t_all = None, None, 0, time.perf_counter()
            while reps < 3 or (time.perf_counter() - t_all < 0.5 and reps < 50):   # best of the repetitions (thread start-up, first touch)
                t0 = time.perf_counter(); rec = s.cpu_raycast(origins, flat, nthreads=threads, sse=True); d = time.perf_counter() - t0
                dtn = d if dtn is None else min(dtn, d)
                reps += 1
            hits_cpu = int((rec["distance"] < 1e29).sum())
            # (2) the scalar Trace oracle: the whole path (both bounces, shading) on the same frame
            t0 = time.perf_counter()
            _, st = orc.trace(rays, pos, sc.sun_angle, shadows=args.shadows)
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {
                "value": round(len(flat) / dtn / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                "sample": f"CPU_RayCast (host mirror of CPURayTrace.cpp:186-249, SSE flavour) over the {len(flat)} primary rays of the bench frame "
                          f"({width}x{height}, {sc.name}), {threads} threads, best of {reps} passes over the frame: {dtn:.3f} s",
                "cpu_model": cpu_model(), "logical_cpus": os.cpu_count(), "usable_cpus": usable_cpus(),
                "cpu_raycast_1_thread": {"value": round(len(flat[sub]) / dt1 / 1e6, 3), "unit": "Mrays/s", "cores": 1,
                                         "sample": f"the same frame once ({len(flat[sub])} rays, {dt1:.2f} s)"},
                "primary_hits_cpu_vs_gpu": [hits_cpu, int(cnt["secondary"])],
                "primary_hits_consistent": bool(abs(hits_cpu - cnt["secondary"]) <= 1e-3 * max(1, cnt["secondary"])),
                "trace_oracle": {"value": round(st["rays"] / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                                 "sample": f"scalar restatement of kernel_main.cl Trace, both bounces, one full frame ({st['rays']} rays, {dt:.2f} s)",
                                 "rays_match_gpu": bool(st["rays"] == cnt["rays"])}}
            # (3) BASELINE config 1 as written: cornell-1k (984 triangles), 640x480, one primary ray per pixel through CPU_RayCast on the
            # host -- the reference's own CPU path at the size its plumbing config names (a host-only session: no GPU involved)
            s.close()
            c1 = scenes.get("cornell-1k")
            with driver.Session(640, 480, host_only=True) as hs:
                hs.load_scene(c1)
                iv1, ip1, pos1 = hs.camera()
                o1 = oracle_lib.Oracle(hs.arenas(), nthreads=threads)
                d1 = np.ascontiguousarray(o1.raygen(640, 480, iv1, ip1).reshape(-1, 3))
                og1 = np.ascontiguousarray(np.tile(np.asarray(pos1, np.float32), (len(d1), 1)))
                t0 = time.perf_counter(); r1 = hs.cpu_raycast(og1, d1, nthreads=1, sse=True); c1_dt1 = time.perf_counter() - t0
                c1_dtn = None
                for _ in range(30):
                    t0 = time.perf_counter(); hs.cpu_raycast(og1, d1, nthreads=threads, sse=True); d = time.perf_counter() - t0
                    c1_dtn = d if c1_dtn is None else min(c1_dtn, d)
            out["cpu_baseline"]["config1_cornell_1k_640x480"] = {
                "value": round(len(d1) / c1_dtn / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                "one_thread": {"value": round(len(d1) / c1_dt1 / 1e6, 3), "unit": "Mrays/s", "cores": 1},
                "primary_hits": int((r1["distance"] < 1e29).sum()),
                "sample": f"CPU_RayCast (SSE flavour) over the {len(d1)} primary rays of cornell-1k at 640x480: {c1_dt1:.3f} s on one thread, best of 30 passes on {threads}: {c1_dtn:.4f} s"}
        if want_live:
            # The contract's `traffic`, measured by THIS run: two short child runs of bench.py under rocprofv3 --pmc (one counter each). Every other
            # number of the line is final by now and this process's session is closed, so the children have the GPU to themselves; both passes
            # together get 60 s, after which (or on any failure) the committed profile's figure stays and the line says so.
            s.close()
            # (every number of the line is final here -- keep a copy where a profiler hang or fault in the next seconds cannot take it)
            try:
                side = os.path.join(ROOT, "gpurun_out")
                if os.path.isdir(side) and os.access(side, os.W_OK):
                    with open(os.path.join(side, "bench_line_before_live_pmc.json"), "w") as f_side:
                        json.dump(out, f_side)
            except OSError:
                pass
            live = live_pmc_traffic(sc.name, width, height, extra_args=["--frames-in-flight", flight, "--band-rows", args.band_rows, "--prewarm-ms", args.prewarm_ms],
                                    budget_s=float(os.environ.get("CRT_BENCH_LIVE_PMC_BUDGET_S", "60")))
            rl = out["roofline"]
            rl["traffic_live_ok"], rl["traffic_live_note"], rl["traffic_live_child_rc"], rl["traffic_live_seconds"] = live["ok"], live["note"], live["child_rc"], live["seconds"]
            if live["ok"]:
                ach = live["bytes"] / dev_s / 1e9
                if ach / HBM_PEAK_GBS <= 1.0:
                    rl["traffic"], rl["traffic_source"] = live["bytes"], live["note"]
                    rl["achieved"], rl["frac"], rl["hbm_frac"] = round(ach, 2), round(ach / HBM_PEAK_GBS, 4), round(ach / HBM_PEAK_GBS, 4)
                    rl["algorithmic_over_traffic"] = round(my_bytes / live["bytes"], 1)
                    rl["error"] = None
                else:
                    rl["traffic_live_ok"] = False
                    rl["traffic_live_note"] = f"live passes read {live['bytes']} B per launch = {ach:.0f} GB/s, above the peak: discarded"
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    s.close()
    if dist is not None:
        dist.barrier(group=ctl)
        if node_barrier is not None:
            node_barrier.close()              # (the creating rank unlinks the shared-memory object)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
