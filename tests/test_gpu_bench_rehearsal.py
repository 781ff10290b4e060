"""bench.py's N > 1 paths rehearsed on ONE GPU (CRT_BENCH_REHEARSE=1: every rank / device state shares GPU 0, control plane over
gloo): the JSON line the driver will read on the 8-GPU node must come out complete, from one process driving N device states
(crt_init_devices) and from N ranks under torch.distributed.run. The numbers mean nothing here (the ranks time-share a device);
the fields and their consistency do. No 8-GPU curve has been measured by this repo: the driver's SCALE_rNN.json is the record."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(cmd, extra_env=None):
    env = dict(os.environ, CRT_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def check_common(d, n):
    assert d["n_gpus"] == n and d["unit"] == "Mrays/s" and d["value"] > 0 and d["scaling"] == "strong"
    assert d["config"]["width"] == 3840 and d["config"]["height"] == 2160 and d["config"]["scene"] == "multi-1M"
    assert d["config"]["rays_per_frame"] > 3840 * 2160                      # primary + bounce rays of the WHOLE frame, summed over the shares
    assert d["config"]["scene_load_s"] > 0 and "max over ranks" in d["config"]["scene_load"]
    assert d["single_gpu_same_workload"]["value"] > 0
    assert d["roofline"]["chain"] is None or d["roofline"]["chain"]["ceiling"] > 0
    assert "cpu_baseline" not in d                                          # rank 0 at N = 1 only


def test_in_process_two_device_states():
    d = run_bench([sys.executable, "bench.py", "--gpus", "2", "--steps", "6", "--warmup", "2"])
    check_common(d, 2)
    assert d["config"]["peer_access"] == [2, 2] and "same-device" in d["config"]["gather_path"]
    assert "crt_init_devices" in d["config"]["tiling"] and d["config"]["control_plane"] in (None, "gloo")
    assert d["synchronous_frames"]["value"] > 0


def test_two_ranks_under_torch_distributed_run():
    d = run_bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                   "--master-port", "29571", "bench.py", "--gpus", "2", "--steps", "6", "--warmup", "2"])
    check_common(d, 2)
    assert d["config"]["control_plane"] == "gloo"                           # the rehearsal never uses RCCL (both ranks sit on GPU 0)
    assert d["delivered_to_host"]["value"] > 0 and d["delivered_to_host_rgba8"]["value"] > 0
    assert "16-row bands round-robin over 2 rank(s)" in d["config"]["tiling"]
