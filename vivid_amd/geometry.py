"""Camera-geometry vector codec (host side, tiny).

Mirrors ``compose_geometry`` / ``decompose_geometry`` of the reference
(``training/utils.py:47-94``): a 3x4 ``tgt2src`` plus two (fx, fy, cx, cy)
intrinsics are flattened to a 20-vector and whitened with fixed dataset
statistics.  The statistics below are the data constants of
``training/utils.py:38-44``; the HIP depth-warp kernel de-normalises with the
same numbers (passed as kernel arguments, see ``engine.warp_features``).
"""
from __future__ import annotations

import torch

GEOM_MEAN = (9.6681e-01, -1.6038e-04, -3.7034e-05, -1.6904e-03, -8.7718e-05,
             9.9869e-01, 3.1288e-03, -1.0794e-03, 1.0653e-05, 3.0997e-03,
             9.6691e-01, 1.2561e-02, 5.7708e+01, 5.7704e+01, 3.2000e+01,
             3.2000e+01, 5.7708e+01, 5.7704e+01, 3.2000e+01, 3.2000e+01)
GEOM_STD = (0.1104, 0.0346, 0.2279, 0.4930, 0.0347, 0.0091, 0.0367, 0.2208, 0.2279,
            0.0368, 0.1088, 1.0751, 6.6464, 6.6511, 0.0000, 0.0000, 6.6464, 6.6511,
            0.0000, 0.0000)


def geometry_stats(imsize: int, dtype=torch.float32, device="cpu"):
    """(mean, std) for an image of side `imsize` (training/utils.py:77-78, 90-91):
    the intrinsics' mean scales with imsize/64 and their std with (imsize/64)**2."""
    mean = torch.tensor(GEOM_MEAN, dtype=dtype, device=device)
    std = torch.tensor(GEOM_STD, dtype=dtype, device=device)
    s = imsize / 64
    mean[12:] = mean[12:] * s
    std[12:] = std[12:] * s * s
    return mean, std


def compose_geometry(tgt2src: torch.Tensor, src_K: torch.Tensor, tgt_K: torch.Tensor, imsize: int = 64) -> torch.Tensor:
    """training/utils.py:64-81.  `src_K`/`tgt_K` are (fx, fy, cx, cy) 4-vectors."""
    mean, std = geometry_stats(imsize, tgt2src.dtype, tgt2src.device)
    flat = torch.cat((tgt2src.reshape(*tgt2src.shape[:-2], 12), src_K, tgt_K), -1)
    return torch.where(std > 0, (flat - mean) / std, torch.zeros_like(flat))


def decompose_geometry(t: torch.Tensor, imsize: int = 64):
    """training/utils.py:84-94 → (tgt2src [...,3,4], src_K [...,3,3], tgt_K [...,3,3])."""
    mean, std = geometry_stats(imsize, t.dtype, t.device)
    t = t * std + mean

    def K3(v):
        K = torch.zeros(v.shape[:-1] + (3, 3), dtype=v.dtype, device=v.device)
        K[..., 0, 0], K[..., 1, 1], K[..., 0, 2], K[..., 1, 2] = v.unbind(-1)
        K[..., 2, 2] = 1
        return K

    return t[..., :12].reshape(*t.shape[:-1], 3, 4), K3(t[..., 12:16]), K3(t[..., 16:])
