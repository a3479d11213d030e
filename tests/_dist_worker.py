"""Worker of test_dist_gloo.py: world_size ranks over gloo on the CPU.  Each rank renders ITS tiles (with the
oracle standing in for the GPU renderer — the tile walk and the single reduce are what is under test) into a
full-frame buffer; pbr_amd.dist.reduce_framebuffer sums them onto rank 0, which checks the assembled image
against the undivided render, bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "physically-based-renderer_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from oracle import ora  # noqa: E402
from pbr_amd import dist as pdist  # noqa: E402
from pbr_amd import scenes  # noqa: E402


def main():
    rank, world, _ = pdist.env_rank_world()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h, spp = 100, 70, 3
    o = ora.Oracle().load_scene(scenes.cornell_box())
    part = o.render(w, h, spp, seed=11, max_bounces=4, tile_rank=rank, tile_count=world, n_threads=2)
    mask = pdist.owned_mask(w, h, rank, world)
    assert (part[~mask] == 0).all() and (part[mask][:, 3] == 1).all()
    buf = torch.from_numpy(part.copy())
    pdist.reduce_framebuffer(buf, 0)
    ok = 1
    if rank == 0:
        full = o.render(w, h, spp, seed=11, max_bounces=4, n_threads=2)
        ok = int(np.array_equal(buf.numpy().view(np.uint32), full.view(np.uint32)))
    flag = torch.tensor([ok])
    dist.broadcast(flag, 0)
    dist.destroy_process_group()
    if rank == 0:
        print("DIST_OK" if ok else "DIST_MISMATCH", flush=True)
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
