#!/usr/bin/env python3
"""Diagnostic: per-wave start/end stamps of one crt_trace_kernel launch (run on the GPU box).
    python tools/wave_timeline.py [scene] [width] [height]
Prints the distribution of wave durations, residency over time and per-XCD busy spans."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clraytracer_amd import _lib, driver, scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "multi-1M"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
with driver.Session(w, h, device=0) as s:
    s.load_scene(scenes.get(name))
    if os.environ.get("CRT_TL_RANKS"):      # this GPU plays rank 0 of N (16-row bands)
        s.set_row_bands(16, 0, int(os.environ["CRT_TL_RANKS"]))
    for _ in range(4):
        s.render_raw(0)
    s.render_raw(16)
    n = C.c_size_t(0)
    _lib.check(s.hip.crt_debug_read_stamps(None, 0, C.byref(n)))
    st = np.zeros((n.value, 8), np.uint64)
    _lib.check(s.hip.crt_debug_read_stamps(st.ctypes.data, n.value, C.byref(n)))
    ms = s.kernel_ms(2)
st = st[st[:, 1] > 0]
t0 = st[:, 0].min()
start = (st[:, 0] - t0).astype(np.float64) / 100.0     # us
end = (st[:, 1] - t0).astype(np.float64) / 100.0
dur = end - start
xcc = (st[:, 3] & 0xF).astype(int)
print(f"{name} {w}x{h}: kernel {ms * 1e3:.1f} us (events), {len(st)} waves stamped, span {end.max():.1f} us")
print("wave duration us: mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max()))
print("shader cycles per wave: mean %.0f max %d ; clock ~ %.2f GHz" % (st[:, 2].mean(), st[:, 2].max(), np.median(st[:, 2].astype(np.float64) / np.maximum(dur, 1e-3)) / 1e3))
edges = np.linspace(0, end.max(), 21)
print("time slice (us) : resident waves (avg)   [the stamped instantiation runs 6 waves/SIMD: capacity 256 CUs x 24 = 6144; the plain kernel 8: 8192]")
for a, b in zip(edges[:-1], edges[1:]):
    ov = np.clip(np.minimum(end, b) - np.maximum(start, a), 0, None).sum() / (b - a)
    print(f"  {a:7.1f}-{b:7.1f}: {ov:7.0f}")
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"XCC {x}: {m.sum()} waves, first start {start[m].min():.1f}, last end {end[m].max():.1f}, sum dur {dur[m].sum() / 1e3:.1f} ms")
if os.environ.get("CRT_KERNEL", "tile") == "persistent":
    hi = lambda a: (a >> np.uint64(32)).astype(np.float64).sum()
    lo = lambda a: (a & np.uint64(0xFFFFFFFF)).astype(np.float64).sum()
    cyc = st[:, 2].astype(np.float64)
    for nm, col in (("service", 4), ("inner", 5), ("leaf", 6), ("enter", 7)):
        print("%-8s passes/trips %.3fM  lanes %.2fM  -> %.1f lanes per trip" % (nm, hi(st[:, col]) / 1e6, lo(st[:, col]) / 1e6, lo(st[:, col]) / max(1.0, hi(st[:, col]))))
    print("cycles per wave: mean %.0f max %.0f ; total wave-cycles %.2fG" % (cyc.mean(), cyc.max(), cyc.sum() / 1e9))
    sys.exit(0)
lo32 = lambda a: (a & np.uint64(0xFFFFFFFF)).astype(np.float64)
hi32 = lambda a: (a >> np.uint64(32)).astype(np.float64)
outer, enter, desc = lo32(st[:, 4]), lo32(st[:, 5]), lo32(st[:, 6])
service = hi32(st[:, 6])                    # CRT_KERNEL=refill: service steps (shade + refill) per wave
inner2, leaf_iters = hi32(st[:, 4]), hi32(st[:, 5])
leaf, lanev = (st[:, 7] >> np.uint64(32)).astype(np.float64), (st[:, 7] & np.uint64(0xFFFFFFFF)).astype(np.float64)
cyc = st[:, 2].astype(np.float64)
print("wave-level loop trips (sum over waves): outer %.2fM enter %.2fM descent %.2fM leaf %.2fM ; lane-level node visits %.2fM -> lanes active per descent trip %.1f" % (
    outer.sum() / 1e6, enter.sum() / 1e6, desc.sum() / 1e6, leaf.sum() / 1e6, lanev.sum() / 1e6, lanev.sum() / max(1.0, desc.sum())))
print("wave-level step executions: first inner step %.2fM, second inner step %.2fM, leaf steps %.2fM with %.2fM triangle iterations, instance entries %.2fM" % (
    desc.sum() / 1e6, inner2.sum() / 1e6, leaf.sum() / 1e6, leaf_iters.sum() / 1e6, enter.sum() / 1e6))
print("  -> vector loads if every step took the vector path: inner 4 x %.2fM = %.2fM, leaf 3 x %.2fM = %.2fM, entries 4 x %.2fM = %.2fM (uniform entries use one scalar load instead)" % (
    (desc.sum() + inner2.sum()) / 1e6, 4 * (desc.sum() + inner2.sum()) / 1e6, leaf_iters.sum() / 1e6, 3 * leaf_iters.sum() / 1e6, enter.sum() / 1e6, 4 * enter.sum() / 1e6))
if service.sum() > 0:
    print("refill: service steps %.3fM (%.1f per wave), loop trips %.2fM -> %.1f trips per service step" % (service.sum() / 1e6, service.sum() / len(st), outer.sum() / 1e6, outer.sum() / service.sum()))
print("cycles per descent trip: all waves %.0f ; slowest 1%% of waves %.0f" % (cyc.sum() / max(1.0, desc.sum()), cyc[dur >= np.percentile(dur, 99)].sum() / max(1.0, desc[dur >= np.percentile(dur, 99)].sum())))
top = np.argsort(-dur)[:8]
print("slowest waves: descent trips", desc[top].astype(int), "leaf trips", leaf[top].astype(int), "enter", enter[top].astype(int), "lane visits", lanev[top].astype(int))
print("slowest waves (us):", np.round(dur[top], 1), "start", np.round(start[top], 1))
