"""Golden-fixture case table and seeded input generators.

Shared by `make_fixtures.py` (which runs the reference) and by the tests
(which run the oracle / the HIP path on the SAME inputs).  Nothing here reads
`/root/reference`.
"""
from __future__ import annotations

import math

import torch

from vivid_amd.arch import NetConfig
from vivid_amd.geometry import compose_geometry

_T = dict(img_resolution=16, model_channels=64, extra_attn=1)

CASES = {
    # HEAD dual-source path, attention at every level (16x16 has 64 ch = 1 head), B=2 targets (4 rows)
    "tiny_dual": dict(cfg=NetConfig(**_T), seed=3, B=2, sigmas=[80.0, 5.0, 0.5, 0.05],
                      sampler=dict(num_steps=4)),
    # classifier-free guidance 1.5 with a vivid-uncond guidance net (adapter semantics)
    "tiny_guided": dict(cfg=NetConfig(**_T), gcfg=NetConfig(**_T, uncond=True), seed=5, B=1,
                        sampler=dict(num_steps=3, guidance=1.5)),
    # encoder run once with sigma=1 and reused (no_time_enc, generate_images.py:52-53); churn on
    "tiny_nte": dict(cfg=NetConfig(**_T, no_time_enc=True), seed=7, B=1, sigmas=[2.0],
                     sampler=dict(num_steps=3, S_churn=2.0)),
    # super-resolution variant: 7-channel first conv, D=32 heads in the UNet, D=64 in the encoder
    "tiny_sr": dict(cfg=NetConfig(img_resolution=32, model_channels=32, attn_resolutions=(), super_res=True),
                    seed=9, B=1, sigmas=[5.0, 0.5], sampler=dict(num_steps=2)),
    # depth-warp Fourier features (config 5): src carries a depth channel, 128 extra channels each side
    "tiny_warp": dict(cfg=NetConfig(img_resolution=16, model_channels=64, attn_resolutions=(8,),
                                    warp_depth_coor=True), seed=11, B=1, sigmas=[5.0, 0.3]),
    # depth map as a 4th source channel only (5-channel first encoder conv).  NB: depth_input and
    # warp_depth_coor together do not run in the reference (encoder conv expects 133 channels, gets 132).
    "tiny_depth": dict(cfg=NetConfig(img_resolution=16, model_channels=64, attn_resolutions=(8,),
                                     depth_input=True), seed=12, B=1, sigmas=[1.0]),
    # the all-zero-source shortcut of training/models.py:647-648: src[:, :3] == 0 everywhere -> zero warp grids
    "tiny_warp_zero": dict(cfg=NetConfig(img_resolution=16, model_channels=64, attn_resolutions=(8,),
                                         warp_depth_coor=True), seed=11, B=1, sigmas=[2.0], zero_src=True),
    # non-default constructor surface: noise / embedding widths (:340-341) and a 4-tap resampling filter (:139, :48-61)
    "tiny_opts": dict(cfg=NetConfig(img_resolution=16, model_channels=64, extra_attn=1, channel_mult_noise=2,
                                    channel_mult_emb=3, resample_filter=(1.0, 3.0, 3.0, 1.0)), seed=17, B=1, sigmas=[5.0, 0.3]),
    # upstream single-source variant kept by the reference under experiments/code
    "tiny_vanilla": dict(cfg=NetConfig(**_T, target_label_dim=20), gcfg=NetConfig(**_T, target_label_dim=20, uncond=True),
                         seed=13, B=2, sigmas=[5.0, 0.2], sampler=dict(num_steps=3, guidance=1.5), snapshot=True),
}


def subsample(t: torch.Tensor, n: int = 4096) -> torch.Tensor:
    """Deterministic sub-sample of a large activation: every k-th element of the flattened tensor."""
    f = t.detach().reshape(-1)
    k = max(1, f.numel() // n)
    return f[::k][:n].clone()


def make_inputs(case: dict) -> dict:
    cfg: NetConfig = case["cfg"]
    B, R = case["B"], cfg.img_resolution
    single = case.get("snapshot", False)
    rows = B if single else 2 * B
    g = torch.Generator("cpu").manual_seed(1000 + case["seed"])
    out = {}
    src = torch.rand(rows, 3, R, R, generator=g) * 2 - 1
    if case.get("zero_src"):
        src = torch.zeros_like(src)
    if cfg.depth_input or cfg.warp_depth_coor:
        depth = torch.rand(rows, 1, R, R, generator=g) * 4 + 1
        src = torch.cat([src, depth], dim=1)
    out["src"] = src
    out["img"] = torch.rand(rows, 3, R, R, generator=g) * 2 - 1
    eps = torch.randn(B, 3, R, R, generator=g)
    out["eps"] = eps if single else eps.repeat_interleave(2, dim=0)
    out["noise"] = out["eps"]
    if cfg.warp_depth_coor:
        # a plausible pose: small rotation about y, small translation, RE10K-like intrinsics
        th = 0.05 * torch.randn(rows, generator=g)
        Rm = torch.zeros(rows, 3, 3)
        Rm[:, 0, 0], Rm[:, 0, 2], Rm[:, 1, 1], Rm[:, 2, 0], Rm[:, 2, 2] = th.cos(), th.sin(), 1.0, -th.sin(), th.cos()
        t = 0.1 * torch.randn(rows, 3, 1, generator=g)
        K = torch.tensor([57.7, 57.7, 32.0, 32.0]) * (R / 64)
        out["geometry"] = compose_geometry(torch.cat([Rm, t], dim=2), K.expand(rows, 4), K.expand(rows, 4), imsize=R)
    else:
        geo = torch.randn(rows, 20, generator=g)
        geo[:, [14, 15, 18, 19]] = 0
        out["geometry"] = geo
    if cfg.super_res:
        low = torch.rand(B, 3, R // 4, R // 4, generator=g) * 2 - 1
        out["cond"] = torch.nn.functional.interpolate(low, size=(R, R), mode="bilinear", align_corners=False)
    return out


def x_for(inp: dict, sigma: float) -> torch.Tensor:
    """Noisy target fed to the denoiser at noise level sigma."""
    return inp["img"] + float(sigma) * inp["eps"]


def make_randn_like(seed: int):
    """A repeatable `randn_like` for the sampler's churn noise (CPU generator; results are moved
    to the input's device), so the reference, the oracle and the HIP path draw identical noise."""
    g = torch.Generator("cpu").manual_seed(777 + seed)

    def randn_like(x):
        return torch.randn(x.shape, generator=g, dtype=torch.float32).to(device=x.device, dtype=x.dtype)

    return randn_like
