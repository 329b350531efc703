"""Sampling driver: seeds -> rank batches -> per-seed noise -> guided sampler -> (SR cascade) -> uint8 images.

Mirrors ``generate_images_nvs`` of the reference (``generate_images.py:139-343``), the first "next" row of
SURVEY.md 8(f): the same arguments where they make sense here, the same per-batch flow and the same result
records (``images, src, tgt, labels, noise, seeds, batch_idx, num_batches, indices``).  What is different, and why:

* networks are modules or paths of local ``network-snapshot-*.pkl`` files; a path is decoded by
  ``vivid_amd.snapshot`` — a restricted unpickler that never executes the source text those files embed
  (the reference ``pickle.load``s them, ``generate_images.py:164-174``).  URLs raise: there is no network here;
* data comes from any iterable of collated batches instead of ``CustomLitDataset``/``DataLoader`` over litdata
  chunks (``:211-226``, out of scope SURVEY 2.1 #11).  A batch is a dict with ``src_image``, ``tgt_image``
  (uint8-range ``[rows,3,H,W]``) and ``geometry`` (``[rows,20]``), rows interleaved ``[s1,s2,s1,s2,...]`` in
  dual-source mode exactly as ``DualSourceCollate`` emits them, plus ``sr_src_image``/``sr_tgt_image``/
  ``sr_geometry`` when an SR model is given;
* the depth model is a callable ``images -> depth map`` (DepthAnythingV2 itself is external);
* the SR hand-off ``torchvision.transforms.functional.resize`` (``:299-302,322``) is ``vh_resize_bilinear``;
* in dual-source mode the SR stage receives pair-duplicated rows like the base stage (at the reference's HEAD the
  SR stage is fed B rows where the dual-source forward expects 2B and cannot run; SURVEY 0.4 lists the same kind
  of breakage for guidance).
"""
from __future__ import annotations

import os
from typing import Callable, Iterable, Optional

import numpy as np
import torch

from . import _lib as L
from . import distributed as vdist
from .encoders import StandardRGBEncoder, add_depth
from .sampler import StackedRandomGenerator, _context, edm_sampler


class EasyDict(dict):
    """Attribute access to dict entries (dnnlib.EasyDict's behaviour, dnnlib/util.py)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value


def resize(x: torch.Tensor, size: int, antialias: bool = True) -> torch.Tensor:
    """Bilinear resize of [N,C,H,W] fp32 to size x size (align_corners=False), optionally anti-aliased."""
    if x.device.type != "cuda":
        raise RuntimeError("vivid_amd.generate.resize runs on the GPU")
    x = x.to(torch.float32).contiguous()
    N, Cc, H, W = x.shape
    out = torch.empty(N, Cc, size, size, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _context(x.device).call("vh_resize_bilinear", L.ResizeArgs(inp=x.data_ptr(), out=out.data_ptr(), planes=N * Cc, hin=H, win=W,
                                                                  hout=size, wout=size, antialias=1 if antialias else 0))
    return out


def generate_images_nvs(
    net,                                            # Main network (vivid_amd.NVPrecond or anything with its call protocol).
    gnet=None,                                      # Guidance network. None = same as main network.
    encoder=None,                                   # Pixel codec. None = StandardRGBEncoder.
    outdir: Optional[str] = None,                   # Where to save PNGs. None = do not save.
    subdirs: bool = False,                          # Create a sub-directory per 1000 seeds?
    seeds=range(16, 24),                            # Random seeds, one image each.
    max_batch_size: int = 32,
    verbose: bool = False,
    device=torch.device("cuda"),
    sampler_fn: Callable = edm_sampler,
    data: Optional[Iterable[dict]] = None,          # Iterable of collated batches (see module docstring).
    sr_model=None,                                  # Super-resolution network for the cascade, or None.
    depth_fn: Optional[Callable] = None,            # images [N,3,H,W] in [0,255] -> depth map [N,1,H,W]
    dual_source: Optional[bool] = None,             # None = take it from net.dual_source
    rng_device=None,                                # device of the per-seed generators (default: `device`, as the reference)
    **sampler_kwargs,
):
    def resolve(m, name):                                                                  # :164-196, without pickle.load
        if not isinstance(m, str):
            return m
        if "://" in m:
            raise NotImplementedError(f"{name}: fetching {m!r} needs the network (the reference's dnnlib.util.open_url); "
                                      "download the snapshot and pass its path")
        from .snapshot import load_network_pkl
        return load_network_pkl(m, dual_source=True if dual_source is None else dual_source).to(device)
    net, gnet, sr_model = resolve(net, "net"), resolve(gnet, "gnet"), resolve(sr_model, "sr_model")
    if data is None:
        raise ValueError("generate_images_nvs needs `data`: an iterable of collated batches (the litdata loader is out of scope)")
    device = torch.device(device)
    if gnet is None:
        gnet = net                                                                          # :180-181
    if encoder is None:
        encoder = StandardRGBEncoder()                                                      # :172-173
    encoder.init(device)
    dual = getattr(net, "dual_source", True) if dual_source is None else dual_source
    rng_device = device if rng_device is None else rng_device
    seeds = list(seeds)
    rank_batches = vdist.rank_batches(len(seeds), max_batch_size)                            # :199-200
    data_iterator = iter(data)
    super_res = (net.img_resolution == 256)                                                  # :229
    sr_sampler_kwargs = {k: v for k, v in sampler_kwargs.items() if k != "guidance"}         # :231-232 (no CFG in the SR model)
    barrier = torch.distributed.barrier if torch.distributed.is_initialized() else (lambda: None)

    class ImageIterable:
        def __len__(self):
            return len(rank_batches)

        def __iter__(self):
            for batch_idx, indices in enumerate(rank_batches):
                r = EasyDict(images=None, src=None, tgt=None, labels=None, noise=None, batch_idx=batch_idx,
                             num_batches=len(rank_batches), indices=indices)
                r.seeds = [seeds[idx] for idx in indices]
                if len(r.seeds) > 0:
                    try:
                        batch = next(data_iterator)
                        if batch is None:
                            continue
                    except StopIteration:
                        continue
                    step = 2 if dual else 1                                                  # :258 vs :270
                    base_src, r.tgt, geometry = (batch[k][::step] for k in ["src_image", "tgt_image", "geometry"])
                    n = min(len(r.seeds), base_src.shape[0])
                    if n == 0:
                        continue
                    r.seeds = r.seeds[:n]
                    r.src, r.tgt, geometry = base_src[:n], r.tgt[:n], geometry[:n]
                    rep = (lambda t: t.repeat_interleave(2, dim=0)) if dual else (lambda t: t)  # :279-280
                    src = encoder.encode_latents(rep(r.src).to(device))
                    r.labels = rep(geometry).to(device)

                    rnd = StackedRandomGenerator(rng_device, r.seeds)                         # :284-291
                    noise = rnd.randn([n] + list(src.shape[1:]), device=rng_device).to(device)
                    r.noise = rep(noise)
                    if depth_fn is not None:                                                  # :293-295
                        src_for_depth = r.src if not super_res else batch["sr_src_image"][::step][:n]
                        depth = depth_fn(rep(src_for_depth).to(device))
                        src = add_depth(depth, src, inv_norm=bool(getattr(net, "depth_input", False)))
                    kw = dict(sampler_kwargs)
                    if super_res:                                                             # :297-303
                        tgt_lat = encoder.encode_latents(r.tgt.to(device))
                        kw["conditioning_image"] = resize(resize(tgt_lat, tgt_lat.shape[-1] // 4), tgt_lat.shape[-1])

                    def randn_like(x, _rnd=rnd):
                        if x.shape[0] == n:
                            return _rnd.randn(list(x.shape), device=rng_device).to(device)
                        return rep(_rnd.randn([n] + list(x.shape[1:]), device=rng_device)).to(device)

                    latents = sampler_fn(net=net, src=src, noise=r.noise, labels=r.labels, gnet=gnet,
                                         randn_like=randn_like, **kw)                        # :305-307
                    r.images = encoder.decode(latents)

                    if sr_model is not None:                                                  # :310-327
                        r.src, r.tgt, sr_geometry = (batch["sr_" + k][::step][:n] for k in ["src_image", "tgt_image", "geometry"])
                        sr_src = encoder.encode_latents(rep(r.src).to(device))
                        rnd = StackedRandomGenerator(rng_device, r.seeds)
                        sr_noise = rnd.randn([n, sr_model.img_channels, sr_model.img_resolution, sr_model.img_resolution],
                                             device=rng_device).to(device)
                        r.noise = rep(sr_noise)
                        r.labels = rep(sr_geometry).to(device)
                        low_res = resize(latents, sr_src.shape[-1])                           # :322

                        def sr_randn_like(x, _rnd=rnd):
                            return rep(_rnd.randn([n] + list(x.shape[1:]), device=rng_device)).to(device)

                        sr_latents = sampler_fn(net=sr_model, src=sr_src, noise=r.noise, labels=r.labels, gnet=sr_model,
                                                conditioning_image=low_res, randn_like=sr_randn_like, **sr_sampler_kwargs)
                        r.images = encoder.decode(sr_latents)

                    if outdir is not None:                                                    # :329-338
                        import PIL.Image
                        for seed, _src, _tgt, image in zip(r.seeds,
                                                           r.src.clip(0, 255).to(torch.uint8).permute(0, 2, 3, 1).cpu().numpy(),
                                                           r.tgt.clip(0, 255).to(torch.uint8).permute(0, 2, 3, 1).cpu().numpy(),
                                                           r.images.permute(0, 2, 3, 1).cpu().numpy()):
                            image_dir = os.path.join(outdir, f"{seed // 1000 * 1000:06d}") if subdirs else outdir
                            os.makedirs(image_dir, exist_ok=True)
                            PIL.Image.fromarray(_src, "RGB").save(os.path.join(image_dir, f"src_{seed:06d}.png"))
                            PIL.Image.fromarray(_tgt, "RGB").save(os.path.join(image_dir, f"tgt_{seed:06d}.png"))
                            PIL.Image.fromarray(image, "RGB").save(os.path.join(image_dir, f"sample_{seed:06d}.png"))
                barrier()                                                                     # :340
                yield r

    return ImageIterable()
