"""Multi-GPU: one process per GPU, pixel tiles sharded over ranks, one framebuffer reduce onto rank 0.

SURVEY.md §8(e): the path shards by 32×32-pixel tile (tile t of a Morton walk over the tile grid
belongs to rank t mod world_size; the scene is replicated).  Every rank renders ALL samples of its
tiles into a zero-initialised full-frame fp32 buffer, then ONE collective — `reduce(sum, dst=0)`,
33 MB at 1080p — assembles the image.  Tiles are disjoint, so the sum is x + 0 and the N-GPU image
is bit-identical to the 1-GPU image.  There is no other exchange on the data path.

The reference has no distributed code at all (single process, single queue:
src/pbr_engine/core/pbr/core/GpuHandle.cpp:77-81); this module is the MI355X-native addition.
torch.distributed backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import os

import numpy as np


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


class _DevArray:
    """__cuda_array_interface__ view of the context's radiance buffer (w*h*4 fp32) for torch.as_tensor."""

    def __init__(self, ptr: int, n_floats: int):
        self.__cuda_array_interface__ = {"shape": (n_floats,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def radiance_tensor(pt, w: int, h: int):
    """Zero-copy torch view (device of the context) of the resolved full-frame radiance buffer."""
    import torch

    ptr = pt.radiance_device_ptr()
    if not ptr:
        raise RuntimeError("no radiance buffer: call frame_begin/frame_resolve first")
    return torch.as_tensor(_DevArray(ptr, w * h * 4), device=f"cuda:{pt.device}").view(h, w, 4)


def reduce_framebuffer(buf, dst: int = 0):
    """The path's single collective: sum the per-rank full-frame buffers onto rank `dst` (in place).
    `buf` is a torch tensor (cuda → RCCL over xGMI; cpu → gloo)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(buf, dst=dst, op=dist.ReduceOp.SUM)
    return buf


def render_sharded(pt, w, h, spp, seed=1, max_bounces=8, integrator=0, rank=0, world=1, samples_per_batch=None):
    """Render this rank's tiles with `pt` (a PathTracer on this rank's GPU), reduce onto rank 0.
    Returns the full image (numpy) on rank 0, None elsewhere."""
    pt.frame_begin(w, h, spp, seed, max_bounces, integrator, tile_rank=rank, tile_count=world)
    left = spp
    k = samples_per_batch or spp
    while left > 0:
        n = min(k, left)
        pt.frame_add_samples(n)
        left -= n
    pt.frame_resolve()
    pt.sync()
    if world > 1:
        import torch

        t = radiance_tensor(pt, w, h)
        reduce_framebuffer(t, 0)
        torch.cuda.synchronize(pt.device)
    return pt.read_radiance() if rank == 0 else None


def owned_mask(w: int, h: int, rank: int, world: int, tile: int = 32) -> np.ndarray:
    """Boolean mask of the pixels rank `rank` owns (host restatement of the tile walk, for tests)."""
    tx, ty = (w + tile - 1) // tile, (h + tile - 1) // tile
    side = 1
    while side < tx or side < ty:
        side <<= 1

    def compact(x):
        x &= 0x55555555
        x = (x ^ (x >> 1)) & 0x33333333
        x = (x ^ (x >> 2)) & 0x0F0F0F0F
        x = (x ^ (x >> 4)) & 0x00FF00FF
        x = (x ^ (x >> 8)) & 0x0000FFFF
        return x

    mask = np.zeros((h, w), bool)
    t = 0
    for m in range(side * side):
        cx, cy = compact(m), compact(m >> 1)
        if cx >= tx or cy >= ty:
            continue
        if t % world == rank:
            mask[cy * tile : (cy + 1) * tile, cx * tile : (cx + 1) * tile] = True
        t += 1
    return mask
