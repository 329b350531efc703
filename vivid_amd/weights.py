"""state_dict layout of the reference ``NVPrecond`` and a seeded synthetic
weight generator.

Key names and shapes reproduce what ``training/models.py`` registers
(``MPConv.weight`` :112, ``Block.emb_gain`` :157, ``UNet.out_gain`` :345,
``MPFourier.freqs/phases`` :93-94, ``NVPrecond.logvar_*`` :623-624), so a
reference checkpoint's state_dict loads unchanged and ours loads into the
reference with ``strict=True`` (checked when the golden fixtures are made,
``tests/golden/make_fixtures.py``).

Trained VIVID checkpoints are fetched from a CDN by the reference
(``generate_images.py:36-40``) and are unavailable offline, so benchmarks and
parity tests use :func:`synth_state_dict`.  ``emb_gain`` / ``out_gain``
initialise to 0 in the reference (which makes F_x == 0); the generator sets
them non-zero so that parity tests are not vacuous.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Tuple

import torch

from .arch import NetConfig, UNetSpec, unet_spec


def _unet_entries(spec: UNetSpec, prefix: str, out: "OrderedDict[str, Tuple[int, ...]]"):
    out[f"{prefix}out_gain"] = ()
    out[f"{prefix}emb_fourier.freqs"] = (spec.cnoise,)
    out[f"{prefix}emb_fourier.phases"] = (spec.cnoise,)
    out[f"{prefix}emb_noise.weight"] = (spec.cemb, spec.cnoise)
    if spec.label_dim:
        out[f"{prefix}emb_label.weight"] = (spec.cemb, spec.label_dim)
    for group, blocks in (("enc", spec.enc), ("dec", spec.dec)):
        for b in blocks:
            if not b.live:
                continue
            p = f"{prefix}{group}.{b.name}."
            if b.kind == "conv":
                out[p + "weight"] = (b.cout, b.cin, 3, 3)
                continue
            out[p + "emb_gain"] = ()
            c0_in = b.cout if b.flavor == "enc" else b.cin
            out[p + "conv_res0.weight"] = (b.cout, c0_in, 3, 3)
            out[p + "emb_linear.weight"] = (b.cout, spec.cemb)
            out[p + "conv_res1.weight"] = (b.cout, b.cout, 3, 3)
            if b.cin != b.cout:
                out[p + "conv_skip.weight"] = (b.cout, b.cin, 1, 1)
            if b.heads:
                out[p + "attn_qkv.weight"] = (3 * b.cout, b.cout, 1, 1)
                if b.xattn:
                    out[p + "x_attn_kv.weight"] = (2 * b.cout, b.cout, 1, 1)
                out[p + "attn_proj.weight"] = (b.cout, b.cout, 1, 1)
    if spec.out_channels:
        out[f"{prefix}out_conv.weight"] = (spec.out_channels, spec.last_ch, 3, 3)


def state_dict_shapes(cfg: NetConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """All state_dict keys of NVPrecond(cfg) with their shapes (registration order
    differs from the reference's; names and shapes are what matter)."""
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    if not cfg.uncond:
        enc = unet_spec(cfg, role="encoder")
        _unet_entries(enc, "encoder.", out)
        del out["encoder.out_gain"]                     # UNetEncoder sets out_gain=None (:528)
    _unet_entries(unet_spec(cfg, role="unet"), "unet.", out)
    out["logvar_fourier.freqs"] = (cfg.logvar_channels,)
    out["logvar_fourier.phases"] = (cfg.logvar_channels,)
    out["logvar_linear.weight"] = (1, cfg.logvar_channels)
    return out


def synth_state_dict(cfg: NetConfig, seed: int = 0, device="cpu",
                     emb_gain_std: float = 0.5, out_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights: N(0,1) conv/linear weights (the reference's own
    init, :112), freqs = 2*pi*N(0,1), phases = 2*pi*U(0,1) (:93-94),
    emb_gain ~ N(0, emb_gain_std^2), out_gain constant.  Each tensor draws from
    its own generator seeded by (seed, key index) so the result does not depend
    on enumeration order or device."""
    sd: Dict[str, torch.Tensor] = OrderedDict()
    for i, (k, shape) in enumerate(state_dict_shapes(cfg).items()):
        g = torch.Generator("cpu").manual_seed((seed * 1000003 + i * 7919 + 12345) % (1 << 31))
        if k.endswith("freqs"):
            t = 2 * math.pi * torch.randn(shape, generator=g)
        elif k.endswith("phases"):
            t = 2 * math.pi * torch.rand(shape, generator=g)
        elif k.endswith("emb_gain"):
            t = emb_gain_std * torch.randn(shape, generator=g)
        elif k.endswith("out_gain"):
            t = torch.full(shape, float(out_gain))
        else:
            t = torch.randn(shape, generator=g)
        sd[k] = t.to(device)
    return sd


def param_count(cfg: NetConfig) -> int:
    n = 0
    for k, s in state_dict_shapes(cfg).items():
        if k.endswith(("freqs", "phases")):
            continue
        m = 1
        for d in s:
            m *= d
        n += m
    return n
