"""A barrier for the ranks of one node (host/ShmBarrier.cpp), set up over an existing torch.distributed group.

bench.py's contract brackets the timed region with a barrier on both sides. With eight ranks that region is 20 x 0.13 ms = 2.6 ms, and a
TCP (gloo) barrier's latency plus exit skew -- 0.1-0.3 ms on an idle host, milliseconds on a loaded one (tools/barrier_cost.py) -- would be
counted as rendering time. All ranks of the contract run on ONE node, so they can meet in shared memory instead: a rank leaves within a
cache-line transfer of the last arrival. torch.distributed still creates the group, checks that the ranks really share a host, and carries the
reductions. No pixel data is involved either way (no data-path collective)."""
import os
import socket

from . import _lib


def _host_id():
    boot = ""
    try:
        boot = open("/proc/sys/kernel/random/boot_id").read().strip()
    except OSError:
        pass
    # ranks see each other's /dev/shm only inside one IPC / mount namespace: the shm directory's device + inode tells containers apart
    st = os.stat("/dev/shm")
    return f"{socket.gethostname()}|{boot}|{st.st_dev}:{st.st_ino}"


class NodeBarrier:
    """wait() = every rank of `dist`'s default group has arrived. Falls back to nothing: create() returns None when the ranks do not share
    a host (or the object cannot be made) and the caller keeps using dist.barrier."""

    def __init__(self, handle, world):
        self.handle, self.world, self.h = handle, world, _lib.host()

    @classmethod
    def create(cls, dist, timeout_ms=120000):
        import torch
        rank, world = dist.get_rank(), dist.get_world_size()
        h = _lib.host()
        ids = [None] * world
        dist.all_gather_object(ids, _host_id())
        same_host = len(set(ids)) == 1
        name = [f"/crt_bench_{os.getpid()}_{os.environ.get('MASTER_PORT', '0')}" if rank == 0 else None]
        dist.broadcast_object_list(name, src=0)
        handle = None
        if same_host and rank == 0:
            handle = h.crth_shm_barrier_open(name[0].encode(), world, 1)
        dist.barrier()                                     # the object exists (or not) before anybody opens it
        if same_host and rank != 0:
            handle = h.crth_shm_barrier_open(name[0].encode(), world, 0)
        ok = torch.tensor([1 if handle else 0], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)          # all or none: a mixed group would deadlock
        if int(ok.item()) == 0:
            if handle:
                h.crth_shm_barrier_close(handle)
            return None
        b = cls(handle, world)
        b.timeout_ms = int(timeout_ms)
        b.wait()                                           # first use outside anything timed
        return b

    def wait(self):
        rc = self.h.crth_shm_barrier_wait(self.handle, self.timeout_ms)
        if rc != 0:
            raise SystemExit(f"node barrier: not every one of the {self.world} ranks arrived within {self.timeout_ms} ms (rc {rc})")

    def close(self):
        if self.handle:
            self.h.crth_shm_barrier_close(self.handle)
            self.handle = None
