#!/usr/bin/env python3
"""What bench.py's two barriers cost a timed region of N ranks: N CPU ranks (gloo, no GPU touched) run bench.py's
    barrier(); t0 = now; <work: sleep W ms, the same on every rank>; barrier(); elapsed = now - t0
and report max-over-ranks(elapsed) - W = what the control plane adds to the region the driver times, plus the exit skew of one barrier
(time.perf_counter is CLOCK_MONOTONIC: comparable across the processes of one host).
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29555 tools/barrier_cost.py [work_ms] [reps]
Measures torch.distributed's gloo barrier and the shared-memory node barrier bench.py brackets its timed regions with since round 6
(clraytracer_amd/node_barrier.py). The ranks hide the GPUs from themselves (HIP_VISIBLE_DEVICES): nothing here touches a device."""
import os
import sys
import time

os.environ["HIP_VISIBLE_DEVICES"] = ""
os.environ["ROCR_VISIBLE_DEVICES"] = ""
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

work_ms = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
from clraytracer_amd.node_barrier import NodeBarrier  # noqa: E402
nb = NodeBarrier.create(dist)
import statistics as st
q = lambda a, p: sorted(a)[min(len(a) - 1, int(p * len(a)))]
for label, bar in (("torch.distributed gloo barrier", dist.barrier), ("shared-memory node barrier", nb.wait if nb else None)):
    if bar is None:
        continue
    for _ in range(20):
        bar()
    over, skew, lat = [], [], []
    for _ in range(reps):
        bar()
        t0 = time.perf_counter()
        t_end = t0 + work_ms * 1e-3
        while time.perf_counter() < t_end:          # busy wait like a host thread that submits frames
            pass
        bar()
        t1 = time.perf_counter()
        v = torch.tensor([t1 - t0, t0, t1], dtype=torch.float64)
        allv = [torch.zeros(3, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allv, v)
        el = max(float(x[0]) for x in allv)
        over.append((el - work_ms * 1e-3) * 1e6)
        skew.append((max(float(x[1]) for x in allv) - min(float(x[1]) for x in allv)) * 1e6)
        lat.append((max(float(x[2]) for x in allv) - (max(float(x[1]) for x in allv) + work_ms * 1e-3)) * 1e6)
    if rank == 0:
        print(f"{label}, {world} ranks on {os.cpu_count()} CPUs, {reps} regions of {work_ms} ms: control-plane overhead per timed region (max over ranks of elapsed - work): "
              f"median {st.median(over):.1f} us, p90 {q(over, 0.9):.1f} us, max {max(over):.1f} us; start skew after a barrier median {st.median(skew):.1f} us (p90 {q(skew, 0.9):.1f}); "
              f"closing barrier after the last rank finished: median {st.median(lat):.1f} us (p90 {q(lat, 0.9):.1f})", flush=True)
if nb:
    nb.close()
dist.destroy_process_group()
