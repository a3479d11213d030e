import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "physically-based-renderer_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def ora():
    from oracle import ora as _ora

    _ora.build()
    return _ora


@pytest.fixture(scope="session")
def pbr():
    import pbr_amd

    if not os.path.exists(pbr_amd.ptc.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    return pbr_amd


def rel_l2(a, b):
    import numpy as np

    a = a[..., :3].astype(np.float64)
    b = b[..., :3].astype(np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))
