"""BASELINE.json config 5: multi-1M at 3840x2160 rendered as 8 row-band ranks (on one GPU, one after the other),
stitched, and compared bit for bit with the single-rank frame; per-rank work counters must add up. The single-rank
3840x2160 frame itself is compared with the CPU oracle (frame bits and every counter)."""
import numpy as np
import pytest

from clraytracer_amd import _lib, driver, scenes
import oracle_lib
from util import bits

pytestmark = pytest.mark.gpu


def test_config5_4k_eight_bands_stitch(nthreads):
    sc = scenes.get("multi-1M")
    with driver.Session(3840, 2160, device=0) as s:
        s.load_scene(sc)
        s.render_raw(8)
        full = s.read_output()
        total = s.counters()
        # the 4K frame against the oracle (10.9 M rays: about a second on the GPU box's host cores)
        orc = oracle_lib.Oracle(s.arenas(), nthreads=nthreads)
        iv, ip, pos = s.camera()
        ref, st = orc.trace(orc.raygen(3840, 2160, iv, ip), pos, sc.sun_angle)
        assert np.array_equal(bits(full), bits(ref)) and total == st
        acc = {k: 0 for k in total}
        stitched = np.zeros_like(full)
        hip = _lib.hip()
        own_all = np.array([hip.crt_row_owner(y, 16, 8) for y in range(2160)])
        for r in range(8):
            s.set_row_bands(16, r, 8)
            s.resize(3840, 2160)
            s.render_raw(8)
            part = s.read_output()
            c = s.counters()
            for k in acc:
                acc[k] = max(acc[k], c[k]) if k == "maxStack" else acc[k] + c[k]
            stitched[own_all == r] = part[own_all == r]
        assert np.array_equal(bits(stitched), bits(full))
        assert acc == total
        assert total["primary"] == 3840 * 2160
