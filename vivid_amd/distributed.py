"""Multi-GPU helpers: one process per GPU, `torch.distributed` (backend "nccl" = RCCL on ROCm).

The sampler has no data-path collective: independent seeds are split across ranks exactly as the
reference's driver does (``generate_images.py:199-200``) and a sample's noise is a function of its
seed only (``StackedRandomGenerator``, :120-134), so results do not depend on placement.  The only
real exchange on the reference's north-star path is the end-of-run all_reduce(SUM) of fp64 feature
moments in ``calculate_metrics.py:176-182,236`` (not an all-gather): ``vivid_amd.metrics.MomentBank``
keeps them in one flat buffer and reduces it with one collective.  Rendezvous follows
``torch_utils/distributed.py:23-48`` (env://).
"""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np
import torch


def init(backend: Optional[str] = None) -> None:
    """env:// process group; RCCL ("nccl") when a GPU is present, gloo otherwise
    (torch_utils/distributed.py:29-45 defaults MASTER_ADDR/PORT/RANK/WORLD_SIZE the same way)."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("LOCAL_RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    if torch.distributed.is_initialized():
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
    torch.distributed.init_process_group(backend=backend, init_method="env://")


def get_rank() -> int:
    return torch.distributed.get_rank() if torch.distributed.is_initialized() else 0


def get_world_size() -> int:
    return torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1


def rank_batches(num_seeds: int, max_batch_size: int, world_size: Optional[int] = None, rank: Optional[int] = None) -> List[np.ndarray]:
    """Index batches this rank processes (generate_images.py:199-200):
    num_batches = max((N-1)//(max_batch*W)+1, 1)*W; array_split(arange(N), num_batches)[rank::W]."""
    W = get_world_size() if world_size is None else world_size
    r = get_rank() if rank is None else rank
    num_batches = max((num_seeds - 1) // (max_batch_size * W) + 1, 1) * W
    return np.array_split(np.arange(num_seeds), num_batches)[r::W]
