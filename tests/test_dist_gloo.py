"""N > 1 path on the CPU: world_size 2 and 3 over gloo (SURVEY §8e — tiles sharded, one reduce)."""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from conftest import run_torchrun  # noqa: E402


@pytest.mark.parametrize("world", [2, 3])
def test_tile_sharding_and_reduce_gloo(ora, world):
    r = run_torchrun(world, [os.path.join(HERE, "_dist_worker.py")], env=dict(os.environ, OMP_NUM_THREADS="1"), timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "DIST_OK" in r.stdout
