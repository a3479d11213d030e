"""Worker of test_gpu_parity.py::test_render_sharded_two_ranks_on_one_gpu: the product's N > 1 path
(pbr_amd.dist.render_sharded: tile-sharded HIP render, zero-copy torch view of the library's radiance buffer, one
reduce onto rank 0) with world_size ranks sharing this box's single GPU over gloo.  Rank 0 checks the assembled frame
against its own undivided render, bit for bit.  (The driver's real multi-GPU runs use RCCL, one GPU per rank.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "physically-based-renderer_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import pbr_amd  # noqa: E402
from pbr_amd import dist as pdist  # noqa: E402
from pbr_amd import scenes  # noqa: E402


def main():
    rank, world, _ = pdist.env_rank_world()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    w, h, spp = 200, 136, 6
    pt = pbr_amd.PathTracer(0).load_scene(scenes.atrium(0.05))
    img = pdist.render_sharded(pt, w, h, spp, seed=4, max_bounces=6, rank=rank, world=world, samples_per_batch=4)
    ok = 1
    if rank == 0:
        full = pt.render(w, h, spp, seed=4, max_bounces=6)
        ok = int(np.array_equal(img.view(np.uint32), full.view(np.uint32)))
    else:
        assert img is None
    flag = torch.tensor([ok])
    dist.broadcast(flag, 0)
    dist.destroy_process_group()
    if rank == 0:
        print("DIST_GPU_OK" if ok else "DIST_GPU_MISMATCH", flush=True)
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
