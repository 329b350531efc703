"""FID / joint-FID / PSNR statistics over a stream of generated batches.

Mirrors ``calculate_stats_for_iterable_nvs`` and ``calculate_metrics_from_stats_nvs`` of the reference
(``calculate_metrics.py:134-248, 295-322``): per-detector fp64 first and second moments of generated and
ground-truth features (and of the joint [image | source] features), PSNR against the target, one
``all_reduce(SUM)`` of the accumulators on the last batch (the only data-carrying collective of the reference's
north-star path — RCCL over xGMI when the process group's backend is "nccl"), then the Fréchet distance on the host
with ``scipy.linalg.sqrtm``.

Detectors (Inception-v3 from an NGC pickle, DINOv2 from torch.hub; ``calculate_metrics.py:45-47,63``) need the
network and are out of scope: they are passed in as callables ``images_uint8[N,3,H,W] -> features[N,F]`` with a
``feature_dim`` attribute.  The fp64 F x F accumulation is a plain library GEMM (``torch.matmul``).
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, Sequence

import numpy as np
import scipy.linalg
import torch

from .generate import EasyDict

STAT_METRICS = ("fid", "fd_dinov2", "joint_fid", "joint_fd_dinov2")


def _all_reduce(x: torch.Tensor) -> torch.Tensor:
    x = x.clone()
    if torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        torch.distributed.all_reduce(x)
    return x


def psnr(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """calculate_metrics.py:147 — per image, on [0,255] values."""
    return 10 * torch.log10(255 ** 2 / ((x.float() - y.float()) ** 2).mean((1, 2, 3)))


def calculate_stats_for_iterable_nvs(image_iter: Iterable, detectors: Dict[str, Callable],
                                     metrics: Sequence[str] = ("fid", "joint_fid", "psnr"), device="cuda"):
    """Yields (r, ref) per batch like the reference; `.stats` is filled on the last batch of the iterable."""
    metrics = list(metrics)
    for m in metrics:
        if m.startswith("joint_"):
            assert m.replace("joint_", "") in metrics
    num_batches = len(image_iter)
    dets = {m: d for m, d in detectors.items() if m in metrics and m in ("fid", "fd_dinov2")}

    def new_state():
        out = []
        for m, d in dets.items():
            F = d.feature_dim
            s = EasyDict(metric=m, detector=d, cum_mu=torch.zeros(F, dtype=torch.float64, device=device),
                         cum_sigma=torch.zeros(F, F, dtype=torch.float64, device=device))
            if "joint_" + m in metrics:
                s.j_cum_mu = torch.zeros(2 * F, dtype=torch.float64, device=device)
                s.j_cum_sigma = torch.zeros(2 * F, 2 * F, dtype=torch.float64, device=device)
            out.append(s)
        return out

    def reduce(state, r):                                               # calculate_metrics.py:174-182
        for s in state:
            mu = _all_reduce(s.cum_mu) / r.num_images
            sigma = (_all_reduce(s.cum_sigma) - mu.ger(mu) * r.num_images) / (r.num_images - 1)
            r.stats[s.metric] = dict(mu=mu.cpu().numpy(), sigma=sigma.cpu().numpy())
            if "joint_" + s.metric in metrics:
                mu = _all_reduce(s.j_cum_mu) / r.num_images
                sigma = (_all_reduce(s.j_cum_sigma) - mu.ger(mu) * r.num_images) / (r.num_images - 1)
                r.stats["joint_" + s.metric] = dict(mu=mu.cpu().numpy(), sigma=sigma.cpu().numpy())

    def gen():
        state, ref_state = new_state(), new_state()
        cum_psnr = torch.zeros(1, dtype=torch.float64, device=device)
        cum_images = torch.zeros([], dtype=torch.int64, device=device)
        for batch_idx, data in enumerate(image_iter):
            images, tgt, src = (None if data.get(k) is None else torch.as_tensor(data[k]).to(device) for k in ("images", "tgt", "src"))
            if images is not None and tgt is not None:
                with torch.no_grad():
                    for s, sref in zip(state, ref_state):               # :158-172
                        f = s.detector(images).to(torch.float64)
                        s.cum_mu += f.sum(0)
                        s.cum_sigma += f.T @ f
                        ft = s.detector(tgt).to(torch.float64)
                        sref.cum_mu += ft.sum(0)
                        sref.cum_sigma += ft.T @ ft
                        if "joint_" + s.metric in metrics:
                            fs = s.detector(src).to(torch.float64)
                            j = torch.cat([f, fs], -1)
                            s.j_cum_mu += j.sum(0)
                            s.j_cum_sigma += j.T @ j
                            j = torch.cat([ft, fs], -1)
                            sref.j_cum_mu += j.sum(0)
                            sref.j_cum_sigma += j.T @ j
                if "psnr" in metrics:
                    cum_psnr += psnr(images, tgt).sum()
                cum_images += images.shape[0]
            r = EasyDict(stats=None, images=images, batch_idx=batch_idx, num_batches=num_batches)
            ref = EasyDict(stats=None, images=images, batch_idx=batch_idx, num_batches=num_batches)
            r.num_images = ref.num_images = int(_all_reduce(cum_images).cpu())    # :225,:228 (one scalar all_reduce per batch)
            if batch_idx == num_batches - 1:                            # :230
                assert r.num_images >= 2
                r.stats, ref.stats = dict(num_images=r.num_images), dict(num_images=r.num_images)
                reduce(state, r)
                reduce(ref_state, ref)
                if "psnr" in metrics:
                    r.stats["psnr"] = dict(val=(_all_reduce(cum_psnr) / r.num_images).cpu().numpy())
            yield r, ref

    class StatsIterable:
        def __len__(self):
            return num_batches

        def __iter__(self):
            return gen()

    return StatsIterable()


def calculate_metrics_from_stats_nvs(stats: dict, ref: dict, metrics: Sequence[str] = ("fid", "joint_fid", "psnr")) -> dict:
    """calculate_metrics.py:295-322: ||mu1-mu2||^2 + tr(S1 + S2 - 2 sqrtm(S1 S2)); PSNR is passed through."""
    out = {}
    for m in metrics:
        if m not in stats or (m in STAT_METRICS and m not in ref):
            continue
        if m in STAT_METRICS:
            d = np.square(stats[m]["mu"] - ref[m]["mu"]).sum()
            s, _ = scipy.linalg.sqrtm(np.dot(stats[m]["sigma"], ref[m]["sigma"]), disp=False)
            out[m] = float(np.real(d + np.trace(stats[m]["sigma"] + ref[m]["sigma"] - s * 2)))
        else:
            out[m] = float(stats[m]["val"])
    return out
