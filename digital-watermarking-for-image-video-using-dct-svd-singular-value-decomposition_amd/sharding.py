"""Frame sharding across the GPUs of one node.

Frames (or independent images) are embarrassingly parallel: rank r of W takes
the contiguous range [r*N//W, (r+1)*N//W).  The only exchange step is the one
the reference's video loop implies (watermark decomposed once, reused for
every frame - SURVEY.md 3.5): rank 0 owns the watermark's tile SVD and
broadcasts the singular values (and, for ranks that also extract, Uw / Vwt)
over RCCL (``backend="nccl"`` on ROCm) - or gloo in the CPU tests.  There is
no all-reduce and no data-path collective per frame.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def frame_range(rank: int, world_size: int, n_frames: int) -> Tuple[int, int]:
    """Contiguous per-rank frame range; ranges tile [0, n_frames) exactly."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world_size")
    if n_frames < 0:
        raise ValueError("n_frames < 0")
    return rank * n_frames // world_size, (rank + 1) * n_frames // world_size


def all_ranges(world_size: int, n_frames: int):
    return [frame_range(r, world_size, n_frames) for r in range(world_size)]


def broadcast_watermark(tensors, src: int = 0, group=None, force: bool = False):
    """Broadcast the watermark decomposition tensors (Sw[, Uw, Vwt]) from
    ``src`` in place.  No-op without an initialised process group (1 GPU), or with
    a group of one rank unless ``force`` (the collective is then issued all the
    same: a single-GPU rehearsal of the RCCL call the N > 1 runs make)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return tensors
    if dist.get_world_size(group) == 1 and not force:
        return tensors
    for t in tensors:
        dist.broadcast(t, src=src, group=group)
    return tensors


def gather_scalars(value: float, group=None) -> np.ndarray:
    """All-gather one float per rank (per-rank PSNR / timing for the report)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return np.array([value], np.float64)
    ws = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    mine = torch.tensor([value], dtype=torch.float64, device=dev)
    out = [torch.zeros_like(mine) for _ in range(ws)]
    dist.all_gather(out, mine, group=group)
    return np.array([float(o.item()) for o in out], np.float64)
