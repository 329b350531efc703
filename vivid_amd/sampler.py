"""Guided multi-step EDM sampler with the reference's signature.

Mirrors ``edm_sampler`` of the reference (``generate_images.py:43-118``): rho
time-step discretisation (:68-70), optional churn (:78-84), Euler step and
Heun correction (:87-114), classifier-free guidance ``ref.lerp(D, guidance)``
(:58-62), dual-source row handling — the net returns half the rows, the update
is applied to row 2i and copied to row 2i+1 (:90-98, :106-111) — and encoder
feature reuse for ``no_time_enc`` nets (:52-53).  The per-step arithmetic runs in
one HIP kernel (``vh_sampler_step``); the denoiser calls go to whatever ``net`` /
``gnet`` are (normally :class:`vivid_amd.NVPrecond`).

``StackedRandomGenerator`` keeps the contract of ``generate_images.py:120-134`` (per-seed streams: ``randn``, ``randn_like``,
``randint``).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib as L

_ctx = {}


def _context(device) -> L.Context:
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    stream = torch.cuda.current_stream(device).cuda_stream
    if key not in _ctx:
        _ctx[key] = L.Context(stream)
    else:
        _ctx[key].set_stream(stream)
    return _ctx[key]


def _step(ctx, x_hat, x_probe, D, Dref, guidance, d_cur, t_hat, t_next, x_next):
    rows = D.shape[0]
    row_mul = x_hat.shape[0] // rows
    a = L.SamplerStepArgs(x_hat=x_hat.data_ptr(), x_probe=x_probe.data_ptr() if x_probe is not None else None,
                          d_cond=D.data_ptr(), d_ref=Dref.data_ptr() if Dref is not None else None, guidance=float(guidance),
                          d_cur=d_cur.data_ptr(), t_hat=float(t_hat), t_next=float(t_next), rows=rows, row_mul=row_mul,
                          row_elems=D[0].numel(), x_next=x_next.data_ptr())
    ctx.call("vh_sampler_step", a)


_side_streams = {}


def _side_stream(dev, name):
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), name)
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=dev)
    return st

# Guidance overlap: up to this many input pixels per evaluation (rows x H x W) the two networks of a guided evaluation run on two
# HIP streams.  Small evaluations cannot fill 256 CUs (the reference's own base@64 preset at batch 1: every launch uses <= 64
# workgroups and ~20 us of fixed costs), so the guidance net's launches fit beside the main net's.  Measured on MI355X
# (profiles/r03_guidance_overlap.txt): base@64 batch 1 / 4 / 8 / 16 / 32 -> 1.38x / 1.27x / 1.21x / 1.12-1.17x / 1.09x; 256x256 batch 1 / 2 -> 1.09x /
# 1.05x; the headline 256x256 batch 16 -> 1.005x (the chip is full; kept serial there, which also keeps per-kernel timings meaningful).
# VIVID_GUIDANCE_OVERLAP=0/1 forces it off / on.
GUIDANCE_OVERLAP_MAX_PIXELS = 32 * 2 * 64 * 64


def guided_denoise(net, gnet, src, x, tt, labels=None, conditioning_image=None, features=None, guidance=1, overlap=None):
    """The sampler's `denoise` closure (generate_images.py:55-62): D = net(src, x, t, labels, cond, inject_features=features) and, when
    guidance != 1, ref = gnet(src, x, t).  Returns (D, ref or None); the CFG combination itself is part of vh_sampler_step.
    `overlap` (None = by size, see above): evaluate gnet on a side stream while net runs on the current one."""
    if guidance == 1 or gnet is None:
        return net(src, x, tt, labels, conditioning_image, inject_features=features).to(torch.float32).contiguous(), None
    if overlap is None:
        env = os.environ.get("VIVID_GUIDANCE_OVERLAP")
        overlap = (env == "1") if env in ("0", "1") else x.shape[0] * x.shape[-1] * x.shape[-2] <= GUIDANCE_OVERLAP_MAX_PIXELS
    if not overlap or gnet is net:
        Dx = net(src, x, tt, labels, conditioning_image, inject_features=features).to(torch.float32).contiguous()
        return Dx, gnet(src, x, tt).to(torch.float32).contiguous()
    dev = x.device
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev, "guidance")
    side.wait_stream(main)                                   # x and tt were produced on the main stream
    with torch.cuda.stream(side):
        ref = gnet(src, x, tt).to(torch.float32).contiguous()
    Dx = net(src, x, tt, labels, conditioning_image, inject_features=features).to(torch.float32).contiguous()
    main.wait_stream(side)
    ref.record_stream(main)                                  # allocated on the side stream, consumed on the main one
    return Dx, ref


class _FeaturePipeline:
    """Encoder features for the sampler's sequence of noise levels.

    The encoder half of NVPrecond sees (src, sigma, geometry) only - never the noisy image (training/models.py:664-667) - and
    the sampler's noise levels are known before its first call (generate_images.py:68-70, :78-84).  Hence:
      * calls at the same level share ONE encoder evaluation.  Without churn the Heun probe of step i (at t_next) and the Euler
        call of step i+1 (t_hat = t_cur = that t_next) are such a pair: 32 encoder evaluations per 32-step run instead of 63 -
        the reference recomputes identical features there;
      * the features of the next level are computed on a side stream while this level's UNet (and guidance net) run.
    Results are bit-identical to calling net(src, x, t, labels, cond) every time: the same kernels on the same inputs."""

    def __init__(self, net, src, labels, cond, levels, dtype, dev):
        self.net, self.src, self.labels, self.cond, self.levels, self.dtype, self.dev = net, src, labels, cond, levels, dtype, dev
        self.stream = _side_stream(dev, "enc")
        self.cur = None            # (level, handle)
        self.ahead = None          # (level, handle, event): launched, not yet consumed
        self.slot = 0
        self.encoder_evals = 0

    def _launch(self, level):
        main = torch.cuda.current_stream(self.dev)
        self.stream.wait_stream(main)        # everything enqueued so far - incl. the UNet call that still reads this slot's previous features
        slot, self.slot = self.slot, self.slot ^ 1
        with torch.cuda.stream(self.stream):
            tt = torch.full((self.src.shape[0],), float(level), dtype=self.dtype, device=self.dev)
            h = self.net.encode_features(self.src, tt, self.labels, self.cond, slot=slot)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.encoder_evals += 1
        return h, ev

    def features(self, k):
        """Handle for call k (levels[k]); the main stream is made to wait for its encoder."""
        level = self.levels[k]
        if self.cur is None or self.cur[0] != level:
            if self.ahead is not None and self.ahead[0] == level:
                _, h, ev = self.ahead
            else:
                h, ev = self._launch(level)
            torch.cuda.current_stream(self.dev).wait_event(ev)
            self.cur, self.ahead = (level, h), None
        if self.ahead is None:
            for j in range(k + 1, len(self.levels)):
                if self.levels[j] != level:
                    self.ahead = (self.levels[j],) + self._launch(self.levels[j])
                    break
        return self.cur[1]


def _pipeline_capable(net, src, gnet=None, guidance=1) -> bool:
    env = os.environ.get("VIVID_FEATURE_PIPELINE")
    if env == "0":
        return False
    if gnet is net and guidance != 1:
        # the guidance call `gnet(src, x, t)` would be a whole ('full') evaluation of the SAME engine on the main stream while that
        # engine's look-ahead encoder runs on the side stream.  NVPrecond orders the two (its whole evaluations wait for an outstanding
        # look-ahead), so results would stay right, but nothing is gained: keep the plain call pattern.
        return False
    return hasattr(net, "encode_features") and not getattr(net, "uncond", None) and not getattr(net, "no_time_enc", None)


def edm_sampler(
    net, src, noise, labels=None, gnet=None, conditioning_image=None,
    num_steps=32, sigma_min=0.002, sigma_max=80, rho=7, guidance=1,
    S_churn=0, S_min=0, S_max=float('inf'), S_noise=1,
    dtype=torch.float32, randn_like=torch.randn_like,
):
    if dtype != torch.float32:
        raise NotImplementedError("the HIP sampler computes in float32 (the reference's default dtype)")
    dev = noise.device
    if dev.type != "cuda":
        raise RuntimeError("vivid_amd.edm_sampler runs on the GPU; inputs must be on a cuda (ROCm) device")
    with torch.cuda.device(dev), torch.no_grad():
        ctx = _context(dev)
        features = None
        if getattr(net, "no_time_enc", None):                                                   # :52-53
            features = net(src, torch.zeros_like(src), torch.ones(src.shape[0], dtype=dtype, device=dev), labels,
                           conditioning_image, return_features=True)

        # Time step discretisation (:68-70), in fp32 like the reference.
        idx = torch.arange(num_steps, dtype=dtype)
        t_steps = (sigma_max ** (1 / rho) + idx / (num_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
        t_steps = torch.cat([t_steps, torch.zeros_like(t_steps[:1])])

        # The noise level of every denoiser call of this run, in call order (Euler call at t_hat, Heun probe at t_next; :78-84, :104)
        pipe, calls = None, [0]
        if features is None and _pipeline_capable(net, src, gnet, guidance):
            levels = []
            for i, (t_cur, t_next) in enumerate(zip(t_steps[:-1], t_steps[1:])):
                churn = S_churn > 0 and S_min <= t_cur <= S_max
                levels.append(float(t_cur + min(S_churn / num_steps, np.sqrt(2) - 1) * t_cur) if churn else float(t_cur))
                if i < num_steps - 1:
                    levels.append(float(t_next))
            pipe = _FeaturePipeline(net, src, labels, conditioning_image, levels, dtype, dev)

        def denoise(x, t):
            tt = torch.full((x.shape[0],), float(t), dtype=dtype, device=dev)
            f = features
            if pipe is not None:
                assert pipe.levels[calls[0]] == float(t), "noise-level schedule out of step with the sampler loop"
                f = pipe.features(calls[0])
                calls[0] += 1
            return guided_denoise(net, gnet, src, x, tt, labels, conditioning_image, f, guidance)   # :55-62

        x_next = (noise.to(dtype) * t_steps[0].item()).contiguous()
        dual = False
        for i, (t_cur, t_next) in enumerate(zip(t_steps[:-1], t_steps[1:])):
            x_cur = x_next
            if S_churn > 0 and S_min <= t_cur <= S_max:                                           # :78-84
                gamma = min(S_churn / num_steps, np.sqrt(2) - 1)
                t_hat = t_cur + gamma * t_cur
                x_hat = (x_cur + (t_hat ** 2 - t_cur ** 2).sqrt().item() * S_noise * randn_like(x_cur)).contiguous()
            else:
                t_hat, x_hat = t_cur, x_cur
            D, ref = denoise(x_hat, t_hat)
            dual = D.shape[0] != x_hat.shape[0]                                                    # :90
            d_cur = torch.empty_like(D)
            x_next = torch.empty_like(x_hat)
            _step(ctx, x_hat, None, D, ref, guidance, d_cur, t_hat, t_next, x_next)              # :93-98
            if i < num_steps - 1:                                                                  # :104-111
                Dp, refp = denoise(x_next, t_next)
                x_corr = torch.empty_like(x_hat)
                _step(ctx, x_hat, x_next, Dp, refp, guidance, d_cur, t_hat, t_next, x_corr)
                x_next = x_corr
        return x_next[::2] if dual else x_next                                                    # :116-118


class StackedRandomGenerator:
    """Per-seed random streams stacked along the batch axis: row i of every draw comes from a generator seeded with
    ``seeds[i] % 2**32`` and from nothing else, so a sample's noise does not depend on which batch or rank it lands in - the
    contract of the reference's class of the same name (generate_images.py:120-134), which the driver and ``edm_sampler``'s
    ``randn_like`` hook rely on."""

    def __init__(self, device, seeds):
        self.device = torch.device(device)
        self.streams = []
        for seed in seeds:
            g = torch.Generator(self.device)
            g.manual_seed(int(seed) % (1 << 32))
            self.streams.append(g)

    def _stacked(self, draw, size):
        """One draw of shape size[1:] per stream, stacked; size[0] must be the number of seeds."""
        size = tuple(size)
        if size[0] != len(self.streams):
            raise ValueError(f"leading dimension {size[0]} != number of seeds {len(self.streams)}")
        rows = [draw(size[1:], g) for g in self.streams]
        return torch.stack(rows) if rows else torch.empty((0,) + size[1:], device=self.device)

    def randn(self, size, **kwargs):
        return self._stacked(lambda shape, g: torch.randn(shape, generator=g, **kwargs), size)

    def randn_like(self, input):
        return self.randn(input.shape, dtype=input.dtype, layout=input.layout, device=input.device)

    def randint(self, *args, size, **kwargs):
        """Per-seed integer draws (generate_images.py:132-134; EDM2-style scripts draw class labels with it)."""
        return self._stacked(lambda shape, g: torch.randint(*args, size=shape, generator=g, **kwargs), size)
