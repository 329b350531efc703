"""Pixel codec and depth-channel assembly around the sampler, on the GPU.

``StandardRGBEncoder`` mirrors the reference's class of the same name
(``training/encoders.py:50-62``): raw uint8 pixels <-> latents in [-1, 1].
``add_depth`` mirrors ``training/utils.py:129-139`` but takes the depth MAP
(the monocular depth model that produces it, DepthAnythingV2, is external and
out of scope — SURVEY 2.1 #5); it returns ``src`` with the normalised depth
appended as a 4th channel, the input format of ``depth_input`` /
``warp_depth_coor`` networks.
"""
from __future__ import annotations

import torch

from . import _lib as L
from .sampler import _context


class StandardRGBEncoder:
    def init(self, device):            # the reference's Encoder.init(device) hook; nothing to set up
        return None

    def encode_pixels(self, x):
        return x

    def encode_latents(self, x: torch.Tensor) -> torch.Tensor:
        """raw pixels (uint8 or float in [0,255]) -> x/127.5 - 1   (training/encoders.py:58-59)"""
        if x.device.type != "cuda":
            raise RuntimeError("vivid_amd.StandardRGBEncoder runs on the GPU")
        if x.dtype != torch.uint8:
            return x.to(torch.float32) / 127.5 - 1
        x = x.contiguous()
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _context(x.device).call("vh_codec", L.CodecArgs(inp=x.data_ptr(), out=out.data_ptr(), n=x.numel(), decode=0))
        return out

    def decode(self, x: torch.Tensor) -> torch.Tensor:
        """latents -> uint8 pixels: (x*127.5+128).clip(0,255).to(uint8)   (training/encoders.py:61-62)"""
        if x.device.type != "cuda":
            raise RuntimeError("vivid_amd.StandardRGBEncoder runs on the GPU")
        x = x.to(torch.float32).contiguous()
        out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            _context(x.device).call("vh_codec", L.CodecArgs(inp=x.data_ptr(), out=out.data_ptr(), n=x.numel(), decode=1))
        return out


def add_depth(depth: torch.Tensor, src: torch.Tensor, inv_norm: bool) -> torch.Tensor:
    """src [N,C,H,W] + depth map [N,1,H,W] -> [N,C+1,H,W]   (training/utils.py:133-139 given the depth map)."""
    if src.device.type != "cuda":
        raise RuntimeError("vivid_amd.add_depth runs on the GPU")
    N, Cc, H, W = src.shape
    assert tuple(depth.shape) == (N, 1, H, W), "depth must be [N,1,H,W] at the source resolution"
    src = src.to(torch.float32).contiguous()
    depth = depth.to(torch.float32).contiguous()
    out = torch.empty(N, Cc + 1, H, W, dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        _context(src.device).call("vh_add_depth", L.AddDepthArgs(src=src.data_ptr(), c=Cc, depth=depth.data_ptr(), rows=N, h=H, w=W,
                                                                 inv_norm=1 if inv_norm else 0, out=out.data_ptr()))
    return out


def _resize(x: torch.Tensor, size, mode: str, align_corners: bool, ch_scale=None, ch_bias=None) -> torch.Tensor:
    if x.device.type != "cuda":
        raise RuntimeError("vivid_amd resize kernels run on the GPU")
    x = x.to(torch.float32).contiguous()
    N, Cc, H, W = x.shape
    ho, wo = (size, size) if isinstance(size, int) else tuple(size)
    out = torch.empty(N, Cc, ho, wo, dtype=torch.float32, device=x.device)
    sc = bi = None
    if ch_scale is not None:
        sc = torch.as_tensor(ch_scale, dtype=torch.float32, device=x.device).contiguous()
        bi = torch.as_tensor(ch_bias, dtype=torch.float32, device=x.device).contiguous()
        assert sc.numel() == Cc and bi.numel() == Cc
    with torch.cuda.device(x.device):
        _context(x.device).call("vh_resize", L.ResizeArgs(
            inp=x.data_ptr(), out=out.data_ptr(), planes=N * Cc, hin=H, win=W, hout=ho, wout=wo, antialias=0,
            mode={"bilinear": 0, "bicubic": 1}[mode], align_corners=1 if align_corners else 0,
            ch_scale=sc.data_ptr() if sc is not None else None, ch_bias=bi.data_ptr() if bi is not None else None, channels=Cc))
    return out


_IMAGENET_MEAN = (0.485, 0.456, 0.406)
_IMAGENET_STD = (0.229, 0.224, 0.225)


def depth_prepare(x: torch.Tensor) -> torch.Tensor:
    """Image in [0,255] -> the input of a DepthAnythingV2 model (training/utils.py:107-115): /255, bicubic resize to 518
    with align_corners (kornia.geometry.transform.resize = F.interpolate of the same arguments), ImageNet normalisation,
    fp16.  One kernel: the affine (y/255 - mean)/std is applied to the interpolated value."""
    assert x.shape[1] == 3, "depth_prepare expects RGB"
    H, W = x.shape[-2:]
    size = (518, 518) if H == W else ((518, int(518 * W / H)) if H < W else (int(518 * H / W), 518))   # kornia side='short'
    sc = [1.0 / (255.0 * s) for s in _IMAGENET_STD]
    bi = [-m / s for m, s in zip(_IMAGENET_MEAN, _IMAGENET_STD)]
    return _resize(x, size, "bicubic", True, sc, bi).to(torch.float16)


def get_depth(depth_model, image: torch.Tensor, shape=None) -> torch.Tensor:
    """depth_model(depth_prepare(image)) resized to `shape` (bilinear, align_corners) as [N,1,h,w]   (training/utils.py:118-126).
    `depth_model` is any callable returning [N, h', w'] (the monocular depth network itself is external)."""
    shape = tuple(image.shape[-2:]) if shape is None else tuple(shape[-2:])
    with torch.no_grad():
        depth = depth_model(depth_prepare(image)).to(torch.float32)[:, None]
        return _resize(depth, shape, "bilinear", True)


def add_depth_from_model(depth_model, image: torch.Tensor, src: torch.Tensor, inv_norm: bool) -> torch.Tensor:
    """The reference's add_depth(depth_model, image, src, inv_norm) (training/utils.py:129-139)."""
    return add_depth(get_depth(depth_model, image, src.shape[-2:]), src, inv_norm)
