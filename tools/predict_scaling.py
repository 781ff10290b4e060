#!/usr/bin/env python3
"""Predict the N-GPU lines of bench.py on ONE GPU (run on the GPU box). Every rank's share of the 3840x2160 frame (16-row bands, rank r of N) is
rendered one rank after the other; the N-rank rate is total rays / the slowest rank's time.

  tools/predict_scaling.py [W H]            steady state: 40 frames per rank (what profiles/r04_predicted_scaling.md holds)
  tools/predict_scaling.py --driver [K W5]  round 6: time EXACTLY what bench.py times for N > 1 at the driver's command
                                            (`--steps 20 --warmup 5`): 100 ms pre-warm, W warm-up frames, crt_sync, then K crt_render(ASYNC) +
                                            crt_sync -- pipeline fill included -- with the frames in flight bench.py uses for that N (3, or 8
                                            for N >= 4; CRT_FLIGHT overrides), and the same-workload single-GPU reference measured the way the
                                            N > 1 line measures `single_gpu_same_workload` (the same region shape). Reports per rank: ms per step, the
                                            first frame's latency (fill), the steady cadence; per N: predicted value and
                                            speedup_vs_single_gpu_same_workload, without and with a control-plane overhead per timed region
                                            (CRT_BARRIER_US, default 0: measure it with tools/barrier_cost.py).
Environment: CRT_PRED_N=1,2,4,8  CRT_BAND=16  CRT_FLIGHT=<slots>  CRT_BARRIER_US=<us>
A PREDICTION from one GPU: eight processes on one host, eight GPUs' clocks and the real control plane only show on a node."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

argv = [a for a in sys.argv[1:] if a != "--driver"]
DRIVER = "--driver" in sys.argv
NS = [int(x) for x in os.environ.get("CRT_PRED_N", "1,2,4,8").split(",")]
BAND = int(os.environ.get("CRT_BAND", "16"))
BARRIER_US = float(os.environ.get("CRT_BARRIER_US", "0"))


def slots_for(n):
    return int(os.environ["CRT_FLIGHT"]) if "CRT_FLIGHT" in os.environ else (3 if n < 4 else 8)


def session(flight, W, H):
    os.environ["CRT_FRAMES_IN_FLIGHT"] = str(flight)           # read by crt_init
    if flight > 4:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    from clraytracer_amd import driver, scenes
    s = driver.Session(W, H, device=0)
    s.load_scene(scenes.get("multi-1M"), device_bvh_build=True)
    return s


if not DRIVER:
    from clraytracer_amd import _lib
    W, H = (int(argv[0]), int(argv[1])) if len(argv) > 1 else (3840, 2160)
    FLIGHT = int(os.environ.get("CRT_FLIGHT", "1"))
    FLAGS = 4 if FLIGHT > 1 else 0
    with session(FLIGHT, W, H) as s:
        targs, iv, ip = s.trace_args()
        fp = C.POINTER(C.c_float)
        a = (C.byref(targs), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp))
        hip = _lib.hip()
        for n in NS:
            times, rays = [], 0
            for r in range(n):
                s.set_row_bands(BAND, r, n)
                hip.crt_render(*a, 8); rays += s.counters()["rays"]
                for _ in range(6):
                    hip.crt_render(*a, FLAGS)
                hip.crt_sync()
                t0 = time.perf_counter()
                for _ in range(40):
                    hip.crt_render(*a, FLAGS)
                hip.crt_sync()
                times.append((time.perf_counter() - t0) / 40)
            print(f"flight={FLIGHT} {W}x{H} N={n}: per-rank ms/frame {np.round(np.array(times) * 1e3, 3)} -> predicted {rays / max(times) / 1e6:.0f} Mrays/s "
                  f"({rays / max(times) / 1e6 / n:.0f} per GPU), imbalance max/mean {max(times) / np.mean(times):.3f}")
    sys.exit(0)

# ---- --driver: the timed region of bench.py, rank by rank ----
from clraytracer_amd import _lib  # noqa: E402
K = int(argv[0]) if len(argv) > 0 else 20
WARM = int(argv[1]) if len(argv) > 1 else 5
W, H = 3840, 2160
REPS = int(os.environ.get("CRT_PRED_REPS", "5"))              # the driver runs the region once; the median of REPS regions is what it should expect
hip = _lib.hip()
fp = C.POINTER(C.c_float)
results = []
by_flight = {}
for n in NS:
    by_flight.setdefault(slots_for(n), []).append(n)
for flight, ns in by_flight.items():
    flags = 4 if flight > 1 else 0
    with session(flight, W, H) as s:
        targs, iv, ip = s.trace_args()
        a = (C.byref(targs), iv.ctypes.data_as(fp), ip.ctypes.data_as(fp))
        stats = _lib.CrtFrameStats()

        def region():
            """bench.py's timed region for the bands currently set: pre-warm, warm-up, K frames + sync. Returns (ms per step, first-frame ms, steady ms)."""
            t_pre = time.perf_counter()
            while (time.perf_counter() - t_pre) * 1e3 < 100.0:
                for _ in range(32):
                    hip.crt_render(*a, flags)
                hip.crt_sync()
            for _ in range(WARM):
                hip.crt_render(*a, flags)
            hip.crt_sync()
            hip.crt_frame_time_stats(None, 1)
            t0 = time.perf_counter()
            for _ in range(K):
                hip.crt_render(*a, flags)
            hip.crt_sync()
            dt = time.perf_counter() - t0
            hip.crt_frame_time_stats(C.byref(stats), 0)
            steady = (stats.extentMs - stats.firstFrameMs) / max(1, stats.frames - 1)
            return dt * 1e3 / K, stats.firstFrameMs, steady

        # the single-GPU reference of the N > 1 lines: the whole frame on this GPU in the SAME region shape and the same frames in flight, as
        # bench.py measures `single_gpu_same_workload` since round 6 (rounds 1-5: 3 + 10 frames before the pre-warm, which under-read it)
        s.set_row_bands(BAND, 0, 1)
        hip.crt_render(*a, 8); rays_all = s.counters()["rays"]
        single_runs = sorted(region()[0] for _ in range(REPS))
        single_ms = single_runs[len(single_runs) // 2]
        for n in ns:
            ranks = []
            rays = 0
            for r in range(n):
                s.set_row_bands(BAND, r, n)
                hip.crt_render(*a, 8); rays += s.counters()["rays"]
                runs = [region() for _ in range(REPS)]
                runs.sort(key=lambda x: x[0])
                ranks.append(runs[len(runs) // 2] + (runs[0][0], runs[-1][0]))
            slow = max(x[0] for x in ranks)
            region_ms = slow * K
            value = rays / (slow * 1e-3) / 1e6
            value_b = rays * K / ((region_ms + BARRIER_US * 1e-3) * 1e-3) / 1e6
            single_value = rays_all / (single_ms * 1e-3) / 1e6
            rec = {"n": n, "frames_in_flight": flight, "steps": K, "warmup": WARM, "rays_per_frame": rays,
                   "per_rank_ms_per_step": [round(x[0], 4) for x in ranks], "per_rank_first_frame_ms": [round(x[1], 3) for x in ranks],
                   "per_rank_steady_ms": [round(x[2], 4) for x in ranks], "per_rank_ms_per_step_min_max_of_reps": [[round(x[3], 4), round(x[4], 4)] for x in ranks],
                   "predicted_value_mrays": round(value, 0), "single_gpu_same_workload_mrays": round(single_value, 0), "single_ms": round(single_ms, 4),
                   "speedup_vs_single_gpu_same_workload": round(value / single_value, 3),
                   "control_plane_us_per_region": BARRIER_US, "predicted_value_with_control_plane": round(value_b, 0),
                   "speedup_with_control_plane": round(value_b / single_value, 3),
                   "steady_state_value_mrays": round(rays / (max(x[2] for x in ranks) * 1e-3) / 1e6, 0)}
            results.append(rec)
            print(json.dumps(rec), flush=True)
print("# N | slots | slowest rank ms/step (K=%d incl. fill) | first frame ms | steady ms | predicted Gray/s | x single GPU same workload | with %.0f us control plane" % (K, BARRIER_US))
for r in sorted(results, key=lambda r: r["n"]):
    print(f"# {r['n']} | {r['frames_in_flight']} | {max(r['per_rank_ms_per_step']):.4f} | {max(r['per_rank_first_frame_ms']):.3f} | {max(r['per_rank_steady_ms']):.4f} | "
          f"{r['predicted_value_mrays'] / 1e3:.2f} | {r['speedup_vs_single_gpu_same_workload']:.2f} | {r['speedup_with_control_plane']:.2f}")
