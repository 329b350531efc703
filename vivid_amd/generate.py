"""Sampling driver: seeds -> rank batches -> per-seed noise -> guided sampler -> (SR cascade) -> uint8 images.

The first "next" row of SURVEY.md 8(f): what ``generate_images_nvs`` of the reference does (``generate_images.py:139-343``),
with the same arguments where they make sense here and the same per-batch records (``images, src, tgt, labels, noise, seeds,
batch_idx, num_batches, indices``).  The work of one batch is a short pipeline of stages (:class:`_Batch` carries the state):

    collate slice -> encode source -> per-seed noise -> [depth channel] -> [SR conditioning] -> sampler -> decode
                  -> [SR stage: resize latents -> second sampler with gnet = sr_model -> decode] -> [PNG dump]

What differs from the reference, and why:

* networks are modules or paths of local ``network-snapshot-*.pkl`` files; a path is decoded by ``vivid_amd.snapshot`` — a
  restricted unpickler that never executes the source text those files embed (the reference ``pickle.load``s them,
  ``generate_images.py:164-174``) — and its pixel codec is taken from the same file (:170-173).  URLs raise: no network here;
* rank 0 resolves the networks first and the other ranks follow (:160-161, :195-196), so that 8 ranks do not decode the same
  snapshot at once;
* data comes from any iterable of collated batches instead of ``CustomLitDataset``/``DataLoader`` over litdata chunks
  (``:211-226``, out of scope SURVEY 2.1 #11).  A batch is a dict with ``src_image``, ``tgt_image`` (uint8-range
  ``[rows,3,H,W]``) and ``geometry`` (``[rows,20]``), rows interleaved ``[s1,s2,s1,s2,...]`` in dual-source mode exactly as
  ``DualSourceCollate`` emits them, plus ``sr_src_image``/``sr_tgt_image``/``sr_geometry`` when an SR model is given;
* the depth model is a callable ``images -> depth map`` (DepthAnythingV2 itself is external);
* the SR hand-off ``torchvision.transforms.functional.resize`` (``:299-302,322``) is ``vh_resize_bilinear``;
* in dual-source mode the SR stage receives pair-duplicated rows like the base stage (at the reference's HEAD the SR stage is
  fed B rows where the dual-source forward expects 2B and cannot run; SURVEY 0.4 lists the same kind of breakage for guidance).
"""
from __future__ import annotations

import os
from typing import Callable, Iterable, List, Optional

import torch

from . import _lib as L
from . import distributed as vdist
from .encoders import StandardRGBEncoder, add_depth
from .sampler import StackedRandomGenerator, _context, edm_sampler

RECORD_FIELDS = ("images", "src", "tgt", "labels", "noise")


class EasyDict(dict):
    """Attribute access to dict entries (what callers of the reference's records expect, dnnlib/util.py)."""

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


def _resolve_networks(net, gnet, sr_model, encoder, device, dual_source):
    """Modules pass through; paths are decoded once each, rank 0 first (generate_images.py:160-196)."""
    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
    rank = vdist.get_rank()
    if dist_on and rank != 0:
        torch.distributed.barrier()                                   # wait until rank 0 has been through the files

    def load(m, what, want_codec=False):
        if not isinstance(m, str):
            return m, None
        if "://" in m:
            raise NotImplementedError(f"{what}: fetching {m!r} needs the network (the reference's dnnlib.util.open_url); "
                                      "download the snapshot and pass its path")
        from .snapshot import network_from_snapshot, read_snapshot, snapshot_encoder
        data = read_snapshot(m)
        module = network_from_snapshot(data, dual_source=True if dual_source is None else dual_source).to(device)
        return module, (snapshot_encoder(data) if want_codec else None)      # raises for a codec this build does not have

    try:
        net, codec = load(net, "net", want_codec=encoder is None)
        gnet, _ = load(gnet, "gnet")
        sr_model, _ = load(sr_model, "sr_model")
    finally:
        if dist_on and rank == 0:
            torch.distributed.barrier()                               # release the other ranks (also when loading failed)
    if encoder is None:
        encoder = codec if codec is not None else StandardRGBEncoder()       # :170-173
    return net, (net if gnet is None else gnet), sr_model, encoder           # :180-181


class _Batch:
    """One rank batch on its way through the stages; `rows()` applies the dual-source pair duplication (:279-280)."""

    def __init__(self, seeds: List[int], collated: dict, dual: bool, device, rng_device):
        self.dual, self.device, self.rng_device = dual, device, rng_device
        self.step = 2 if dual else 1                                  # DualSourceCollate interleaves [s1, s2]: keep every other row (:258)
        self.collated = collated
        first = collated["src_image"][::self.step]
        self.n = min(len(seeds), first.shape[0])                      # a short last batch of the loader shortens the seed list (:261-264)
        self.seeds = list(seeds[:self.n])

    def field(self, key: str) -> torch.Tensor:
        return self.collated[key][::self.step][:self.n]

    def rows(self, t: torch.Tensor) -> torch.Tensor:
        return t.repeat_interleave(2, dim=0) if self.dual else t

    def generator(self) -> StackedRandomGenerator:
        return StackedRandomGenerator(self.rng_device, self.seeds)    # noise is a function of the seed only (:284)

    def noise(self, rnd: StackedRandomGenerator, shape_per_seed) -> torch.Tensor:
        return self.rows(rnd.randn([self.n] + list(shape_per_seed), device=self.rng_device).to(self.device))

    def churn_noise_fn(self, rnd: StackedRandomGenerator) -> Callable:
        def randn_like(x):
            if x.shape[0] == self.n:
                return rnd.randn(list(x.shape), device=self.rng_device).to(self.device)
            return self.rows(rnd.randn([self.n] + list(x.shape[1:]), device=self.rng_device)).to(self.device)
        return randn_like


def _dump_pngs(outdir: str, subdirs: bool, seeds, src, tgt, images):
    """src_/tgt_/sample_%06d.png per seed, optionally in a directory per 1000 seeds (:329-338)."""
    import PIL.Image

    def hwc_u8(t):
        t = t if t.dtype == torch.uint8 else t.clip(0, 255).to(torch.uint8)
        return t.permute(0, 2, 3, 1).cpu().numpy()

    planes = {"src": hwc_u8(src), "tgt": hwc_u8(tgt), "sample": hwc_u8(images)}
    for i, seed in enumerate(seeds):
        where = os.path.join(outdir, f"{seed // 1000 * 1000:06d}") if subdirs else outdir
        os.makedirs(where, exist_ok=True)
        for stem, arr in planes.items():
            PIL.Image.fromarray(arr[i], "RGB").save(os.path.join(where, f"{stem}_{seed:06d}.png"))


def generate_images_nvs(
    net,                                            # Main network (vivid_amd.NVPrecond or anything with its call protocol), or a snapshot path.
    gnet=None,                                      # Guidance network. None = same as main network.
    encoder=None,                                   # Pixel codec. None = the snapshot's, else StandardRGBEncoder.
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
    resize_antialias: bool = True,                  # the three torchvision resizes of the cascade (:299-302, :322), see below
    **sampler_kwargs,
):
    """``resize_antialias``: the reference calls ``torchvision.transforms.functional.resize(tensor, size)`` without naming
    ``antialias``; which filter that is depends on the torchvision version ("pytorch > 2" admits both): torchvision >= 0.17
    defaults to antialias=True for tensors, 0.15-0.16 to False with a warning.  This build follows current torchvision (True);
    pass False to reproduce a run made with an older torchvision.  It matters only for the 4x DOWN-scale of the SR net's
    conditioning image (:299-302); for the up-scales the two filters coincide."""
    device = torch.device(device)
    net, gnet, sr_model, encoder = _resolve_networks(net, gnet, sr_model, encoder, device, dual_source)
    if data is None:
        raise ValueError("generate_images_nvs needs `data`: an iterable of collated batches (the litdata loader is out of scope)")
    encoder.init(device)
    dual = getattr(net, "dual_source", True) if dual_source is None else dual_source
    rng_device = device if rng_device is None else rng_device
    seeds = list(seeds)
    my_batches = vdist.rank_batches(len(seeds), max_batch_size)                               # :199-200
    loader = iter(data)
    net_is_sr = net.img_resolution == 256                                                    # :229
    sr_kwargs = {k: v for k, v in sampler_kwargs.items() if k != "guidance"}                 # :231-232: no CFG in the SR stage
    sync = torch.distributed.barrier if (torch.distributed.is_available() and torch.distributed.is_initialized()) else (lambda: None)

    def base_stage(b: _Batch, r: EasyDict):
        r.src, r.tgt = b.field("src_image"), b.field("tgt_image")
        src = encoder.encode_latents(b.rows(r.src).to(device))
        r.labels = b.rows(b.field("geometry")).to(device)
        rnd = b.generator()
        r.noise = b.noise(rnd, src.shape[1:])                                                 # :284-291
        if depth_fn is not None:                                                              # :293-295
            views = r.src if not net_is_sr else b.field("sr_src_image")
            src = add_depth(depth_fn(b.rows(views).to(device)), src, inv_norm=bool(getattr(net, "depth_input", False)))
        kw = dict(sampler_kwargs)
        if net_is_sr:                                                                         # :297-303: blurred target as conditioning
            tgt_lat = encoder.encode_latents(r.tgt.to(device))
            kw["conditioning_image"] = resize(resize(tgt_lat, tgt_lat.shape[-1] // 4, resize_antialias), tgt_lat.shape[-1], resize_antialias)
        latents = sampler_fn(net=net, src=src, noise=r.noise, labels=r.labels, gnet=gnet,
                             randn_like=b.churn_noise_fn(rnd), **kw)                         # :305-307
        r.images = encoder.decode(latents)
        return latents

    def sr_stage(b: _Batch, r: EasyDict, latents: torch.Tensor):                              # :310-327
        r.src, r.tgt = b.field("sr_src_image"), b.field("sr_tgt_image")
        sr_src = encoder.encode_latents(b.rows(r.src).to(device))
        rnd = b.generator()                                                                   # the same seeds start over (:319)
        r.noise = b.noise(rnd, [sr_model.img_channels, sr_model.img_resolution, sr_model.img_resolution])
        r.labels = b.rows(b.field("sr_geometry")).to(device)
        low_res = resize(latents, sr_src.shape[-1], resize_antialias)                         # :322
        sr_latents = sampler_fn(net=sr_model, src=sr_src, noise=r.noise, labels=r.labels, gnet=sr_model,
                                conditioning_image=low_res, randn_like=b.churn_noise_fn(rnd), **sr_kwargs)
        r.images = encoder.decode(sr_latents)

    def run_batch(batch_idx, indices) -> EasyDict:
        r = EasyDict({k: None for k in RECORD_FIELDS})
        r.update(batch_idx=batch_idx, num_batches=len(my_batches), indices=indices, seeds=[seeds[i] for i in indices])
        if not r.seeds:
            return r
        collated = next(loader, None)
        if collated is None:
            return r
        b = _Batch(r.seeds, collated, dual, device, rng_device)
        if b.n == 0:
            return r
        r.seeds = b.seeds
        latents = base_stage(b, r)
        if sr_model is not None:
            sr_stage(b, r, latents)
        if outdir is not None:
            _dump_pngs(outdir, subdirs, r.seeds, r.src, r.tgt, r.images)
        return r

    class ImageIterable:
        def __len__(self):
            return len(my_batches)

        def __iter__(self):
            for batch_idx, indices in enumerate(my_batches):
                with torch.no_grad():
                    r = run_batch(batch_idx, indices)
                sync()                                                                        # one barrier per batch (:340)
                yield r

    return ImageIterable()
