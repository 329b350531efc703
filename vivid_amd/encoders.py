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
