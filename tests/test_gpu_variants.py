"""Opt-in kernel structures must be bit-identical to the default megakernel (own module: one device session at a time)."""
import numpy as np
import pytest

from clraytracer_amd import driver, scenes
from util import bits

pytestmark = pytest.mark.gpu
FLAG_COUNT = 8


@pytest.mark.parametrize("variant", ["wavefront"])
def test_kernel_variants_are_bit_identical(nthreads, monkeypatch, variant):
    """The opt-in kernel structure (CRT_KERNEL=wavefront: one launch per bounce with ballot compaction in between) renders
    the same bits and counts the same work as the default megakernel. (Round 1's `persistent` and `lds` structures were
    retired in round 2; DESIGN.md keeps their measurements.)"""
    sc = scenes.get("tiny")
    monkeypatch.delenv("CRT_KERNEL", raising=False)
    with driver.Session(256, 144, device=0) as s:
        s.load_scene(sc)
        s.render_raw(FLAG_COUNT)
        ref = s.read_output(); ref_cnt = s.counters()
    monkeypatch.setenv("CRT_KERNEL", variant)
    with driver.Session(256, 144, device=0) as s:
        s.load_scene(sc)
        s.render_raw(FLAG_COUNT)
        got = s.read_output(); cnt = s.counters()
        s.render_raw(0)
        got2 = s.read_output()
    assert np.array_equal(bits(got), bits(ref)) and np.array_equal(bits(got2), bits(ref))
    assert cnt == ref_cnt
