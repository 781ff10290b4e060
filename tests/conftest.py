import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    # make sure the native libraries exist (the driver normally ran __graft_entry__.build() before)
    from clraytracer_amd import _lib
    import oracle_lib
    if not all(os.path.exists(p) for p in (_lib.HIP_SO, _lib.HOST_SO, oracle_lib.ORACLE_SO)):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def nthreads():
    return min(16, os.cpu_count() or 1)
