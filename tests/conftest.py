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


def run_torchrun(nproc, script_args, env=None, cwd=None, timeout=600):
    """`python -m torch.distributed.run` on 127.0.0.1 with a port the OS just handed out.  Between closing that probe socket and the
    rendezvous store's bind another process can take the port (or it is still in TIME_WAIT from an earlier run): that one failure,
    and only that one, is retried with a new port."""
    import socket
    import subprocess

    r = None
    for _ in range(4):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + list(script_args)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=dict(env or os.environ, MASTER_ADDR="127.0.0.1"), cwd=cwd)
        if r.returncode == 0 or "EADDRINUSE" not in r.stderr:
            break
    return r
