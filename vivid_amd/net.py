"""Drop-in ``NVPrecond`` backed by the HIP engine.

Same constructor arguments, call protocol, attributes and state_dict key names
as the reference's ``training.models.NVPrecond`` (training/models.py:589-749;
SURVEY.md 8(b)), so ``edm_sampler`` / ``generate_images_nvs`` style drivers can
use it unchanged:

    net(src, dst, sigma, geometry=None, conditioning_image=None, force_fp32=False,
        return_logvar=False, return_features=False, inject_features=None)

Differences, all deliberate:
  * arithmetic is fp32-grade on gfx950 MFMA: ``precision="bf16x3"`` (default; fp32 emulated by a bf16
    hi/lo split, 3 MFMAs per product, fp32 accumulate, ~1e-5 rel-L2 per call) or ``precision="fp32"``
    (exact fp32 MFMA, ~1e-6); ``use_fp16`` / ``force_fp32`` are accepted and ignored — the reference's
    fp16 mode is outside the 1e-3 parity budget;
  * inference only (no autograd through the HIP path);
  * the reference selects dual-source vs single-source with a module global
    (``custom_litdata_loader.VANILLA_MODE``); here it is the ``dual_source`` argument;
  * an ``uncond`` net accepts ``geometry=None`` and needs no features — the meaning of
    the reference's zero-feature branch (:727-736), which HEAD's dual-source forward
    cannot reach;
  * there is no CPU fallback: calling the net on a non-GPU tensor raises.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from .arch import NetConfig
from .engine import Engine
from .weights import state_dict_shapes


class _Node(torch.nn.Module):
    """Parameter container reproducing the reference's module tree (for state_dict key names).  Every node of one net shares
    `_epoch` (a one-element list): any tensor assigned, registered or deleted anywhere in the tree bumps it, which is what tells
    NVPrecond that the weights it prepared may be stale (EMA swaps by assignment, parametrisations, ...)."""

    def __init__(self, epoch=None):
        super().__init__()
        object.__setattr__(self, "_epoch", epoch if epoch is not None else [0])

    def child(self, name: str) -> "_Node":
        if name not in self._modules:
            self.add_module(name, _Node(self._epoch))
        return self._modules[name]

    def __setattr__(self, name, value):
        if isinstance(value, torch.Tensor) or name in self._parameters or name in self._buffers:
            self._epoch[0] += 1
        super().__setattr__(name, value)

    def __delattr__(self, name):
        self._epoch[0] += 1
        super().__delattr__(name)

    def register_parameter(self, name, param):
        self._epoch[0] += 1
        super().register_parameter(name, param)

    def register_buffer(self, name, tensor, persistent=True):
        self._epoch[0] += 1
        super().register_buffer(name, tensor, persistent=persistent)


class FeatureHandle:
    """Encoder features that stay inside the engine (NVPrecond.encode_features): no NCHW copies, no re-split for the matrix
    kernels.  Valid until the same net encodes into the same `slot` again."""

    def __init__(self, net, slot, B, version):
        self.net, self.slot, self.B, self.version = net, slot, B, version


class NVPrecond(torch.nn.Module):
    def __init__(self, img_resolution, img_channels, source_label_dim, target_label_dim,
                 use_fp16=True, sigma_data=0.5, logvar_channels=128, super_res=False, no_time_enc=None,
                 depth_input=False, warp_depth_coor=False, uncond=None, noisy_sr=0.25,
                 dual_source=True, precision=None, **unet_kwargs):
        super().__init__()
        import os
        if precision is None:
            precision = os.environ.get("VIVID_PRECISION", "bf16x3")
        allowed = {"model_channels", "channel_mult", "num_blocks", "attn_resolutions", "extra_attn",
                   "label_balance", "concat_balance", "res_balance", "attn_balance", "clip_act", "dropout",
                   "epipolar_attention_bias", "channel_mult_noise", "channel_mult_emb", "resample_filter"}
        for k in unet_kwargs:
            if k not in allowed:
                raise TypeError(f"NVPrecond: unexpected keyword {k!r}")
        if unet_kwargs.get("epipolar_attention_bias"):
            raise NotImplementedError("epipolar attention bias is dead code at HEAD (SURVEY 2.1 #13) and not built")
        if unet_kwargs.get("dropout"):
            raise NotImplementedError("dropout is a training-time option (training/models.py:178); this is the inference path")
        kw = {k: v for k, v in unet_kwargs.items() if k in ("model_channels", "num_blocks", "extra_attn", "label_balance",
                                                             "concat_balance", "res_balance", "attn_balance", "clip_act",
                                                             "channel_mult_noise", "channel_mult_emb")}
        if "resample_filter" in unet_kwargs:
            kw["resample_filter"] = tuple(float(v) for v in unet_kwargs["resample_filter"])
        if "channel_mult" in unet_kwargs:
            kw["channel_mult"] = tuple(unet_kwargs["channel_mult"])
        if "attn_resolutions" in unet_kwargs:
            kw["attn_resolutions"] = tuple(unet_kwargs["attn_resolutions"])
        kw.setdefault("model_channels", 192)            # UNet default, training/models.py:326
        self.cfg = NetConfig(img_resolution=img_resolution, img_channels=img_channels,
                             source_label_dim=source_label_dim, target_label_dim=target_label_dim,
                             sigma_data=sigma_data, logvar_channels=logvar_channels, super_res=bool(super_res),
                             no_time_enc=no_time_enc, depth_input=bool(depth_input), warp_depth_coor=bool(warp_depth_coor),
                             uncond=uncond, noisy_sr=noisy_sr if noisy_sr is not None else 0.0, use_fp16=use_fp16, **kw)
        # attributes the reference's callers read
        self.img_resolution = img_resolution
        self.img_channels = img_channels
        self.use_fp16 = use_fp16
        self.sigma_data = sigma_data
        self.super_res = super_res
        self.no_time_enc = no_time_enc
        self.depth_input = depth_input
        self.warp_depth_coor = warp_depth_coor
        self.uncond = uncond
        self.noisy_sr = noisy_sr
        self.dual_source = dual_source

        # parameters / buffers under the reference's names, with the reference's initialisation
        self._tree_epoch = [0]
        for key, shape in state_dict_shapes(self.cfg).items():
            *path, leaf = key.split(".")
            node = self
            for name in path:
                if isinstance(node, _Node):
                    node = node.child(name)
                else:
                    if name not in node._modules:
                        node.add_module(name, _Node(self._tree_epoch))
                    node = node._modules[name]
            if leaf == "freqs":
                node.register_buffer(leaf, 2 * np.pi * torch.randn(shape))
            elif leaf == "phases":
                node.register_buffer(leaf, 2 * np.pi * torch.rand(shape))
            elif leaf in ("emb_gain", "out_gain"):
                node.register_parameter(leaf, torch.nn.Parameter(torch.zeros(shape)))
            else:
                node.register_parameter(leaf, torch.nn.Parameter(torch.randn(shape)))
        self.requires_grad_(False)
        self.precision = precision
        self._engine = Engine(self.cfg, dual_source=dual_source, precision=precision)
        self._prepared_fp = None
        self._tensors = None            # flat list of parameters and buffers (rebuilt after _apply / load_state_dict)
        self._weights_epoch = 0
        self._enc_event = None          # end of the last encode_features launch (it may still be running on another stream)

    @classmethod
    def from_config(cls, cfg: NetConfig, dual_source: bool = True, precision=None) -> "NVPrecond":
        return cls(img_resolution=cfg.img_resolution, img_channels=cfg.img_channels,
                   source_label_dim=cfg.source_label_dim, target_label_dim=cfg.target_label_dim,
                   use_fp16=cfg.use_fp16, sigma_data=cfg.sigma_data, logvar_channels=cfg.logvar_channels,
                   super_res=cfg.super_res, no_time_enc=cfg.no_time_enc, depth_input=cfg.depth_input,
                   warp_depth_coor=cfg.warp_depth_coor, uncond=cfg.uncond, noisy_sr=cfg.noisy_sr,
                   dual_source=dual_source, precision=precision, model_channels=cfg.model_channels, channel_mult=cfg.channel_mult,
                   num_blocks=cfg.num_blocks, attn_resolutions=cfg.attn_resolutions, extra_attn=cfg.extra_attn,
                   label_balance=cfg.label_balance, concat_balance=cfg.concat_balance,
                   res_balance=cfg.res_balance, attn_balance=cfg.attn_balance, clip_act=cfg.clip_act,
                   channel_mult_noise=cfg.channel_mult_noise, channel_mult_emb=cfg.channel_mult_emb,
                   resample_filter=cfg.resample_filter)

    # -- weights ---------------------------------------------------------------------------
    def _apply(self, fn, *args, **kwargs):          # .to() / .cuda() / .float(): the tensors are replaced
        self._tensors = None
        self._weights_epoch += 1
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._tensors = None
        self._weights_epoch += 1
        return super().load_state_dict(*args, **kwargs)

    def _fingerprint(self):
        """Changes whenever the weights may have: an epoch bumped by load_state_dict / _apply, the tree epoch bumped by any tensor
        assignment / registration / deletion in the parameter tree (_Node), plus the autograd version counters of the ~400 tensors
        (in-place edits), summed over a flat list that is rebuilt only when one of the epochs moved."""
        epoch = (self._weights_epoch, self._tree_epoch[0])
        if self._tensors is None or self._tensors[0] != epoch:
            self._tensors = (epoch, list(self.parameters()) + list(self.buffers()))
        ts = self._tensors[1]
        return (epoch, sum(t._version for t in ts), ts[0].data_ptr() if ts else 0)

    def _prepare(self, device):
        fp = self._fingerprint()
        if self._prepared_fp == fp and self._engine.device == device:
            self._engine._ensure_ctx(device)
            return
        params: Dict[str, torch.Tensor] = {}
        for k, v in self.state_dict().items():
            if v.device != device or v.dtype != torch.float32 or not v.is_contiguous():
                raise RuntimeError(f"NVPrecond: parameter {k} must be a contiguous fp32 tensor on {device} "
                                   f"(got {v.dtype} on {v.device}); call net.to(device).float() first")
            params[k] = v
        self._engine.prepare_weights(params, device)
        self._prepared_fp = fp

    # -- split evaluation -------------------------------------------------------------------
    @torch.no_grad()
    def encode_features(self, src, sigma, geometry=None, conditioning_image=None, slot: int = 0) -> FeatureHandle:
        """The encoder half of forward() (training/models.py:664-667: `features = self.encoder(src, c_noise * ..., geometry)`) on the
        current stream, leaving the features in the engine.  Pass the handle as `inject_features` to evaluate the UNet half on
        them in place.  The encoder sees (src, sigma, geometry) only - never the noisy image - so a sampler can compute the
        features of its NEXT noise level on a side stream while this level's UNet runs, and share one encoder evaluation between
        the calls it makes at the same level (vivid_amd.sampler).  Two slots: one can be filled while the other is read."""
        cfg, eng = self.cfg, self._engine
        if cfg.uncond:
            raise RuntimeError("an uncond net has no encoder")
        dev = src.device
        if dev.type != "cuda":
            raise RuntimeError("vivid_amd.NVPrecond runs on MI355X only: inputs must be on a GPU device")
        with torch.cuda.device(dev):
            self._prepare(dev)
            rm = 2 if self.dual_source else 1
            rows = src.shape[0]
            if rows % rm:
                raise ValueError(f"dual-source input needs an even number of rows, got {rows}")
            B, R = rows // rm, cfg.img_resolution
            src_c = 3 + int(cfg.depth_input or cfg.warp_depth_coor)
            if cfg.warp_depth_coor:
                assert src.shape[1] == 4, "warp_depth_coor requires depth channel in src"      # :644
            if geometry is None:
                raise TypeError("geometry is required (the reference multiplies None by an int here, :631)")
            prog = eng.program("features", B, bool(cfg.super_res), False, slot=slot)
            self._order_after_encoder(dev)
            self._put(prog, "sigma", sigma.reshape(-1), (rows,))
            self._put(prog, "geometry", geometry.reshape(rows, -1), (rows, cfg.source_label_dim))
            self._put(prog, "src", src, (rows, src_c, R, R))
            prog.plan.run()
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            self._enc_event = ev
            return FeatureHandle(self, slot, B, self._prepared_fp)

    def _order_after_encoder(self, dev):
        """Programs that run the ENCODER ('full', 'features') share its split-K scratch (Engine.scratch_enc) and the engine's one
        context: a whole evaluation issued on this stream while a look-ahead encode_features is still running on another one would
        interleave partial sums with it.  Make this stream wait for the last encoder launch first (free when it already finished or
        ran on this stream); 'bound' / 'inject' / 'uncond' programs touch only the UNet's scratch and need no ordering."""
        if self._enc_event is not None:
            torch.cuda.current_stream(dev).wait_event(self._enc_event)

    @staticmethod
    def _put(prog, name, t, shape):
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name} shape {tuple(t.shape)} != {tuple(shape)}")
        prog.view(name).copy_(t.to(torch.float32))

    # -- forward ---------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, src, dst, sigma, geometry=None, conditioning_image=None, force_fp32=False,
                return_logvar=False, return_features=False, inject_features=None, **unet_kwargs):
        if unet_kwargs:
            # the reference passes these on to UNetEncoder.forward / XAttnUNet.forward (:667,:679), neither of which takes any
            # keyword (:483,:536): there, too, any extra keyword ends in a TypeError
            raise TypeError(f"NVPrecond.forward: unexpected keywords {sorted(unet_kwargs)}")
        dev = dst.device
        if dev.type != "cuda":
            raise RuntimeError("vivid_amd.NVPrecond runs on MI355X only: inputs must be on a GPU device "
                               "(no CPU path; use oracle/vivid_ref.py for a CPU reference)")
        cfg, eng = self.cfg, self._engine
        with torch.cuda.device(dev):
            self._prepare(dev)
            rm = 2 if self.dual_source else 1
            rows = dst.shape[0]
            if rows % rm:
                raise ValueError(f"dual-source input needs an even number of rows, got {rows}")
            B = rows // rm
            R = cfg.img_resolution
            src_c = 3 + int(cfg.depth_input or cfg.warp_depth_coor)
            if tuple(dst.shape) != (rows, cfg.img_channels, R, R):
                raise ValueError(f"dst shape {tuple(dst.shape)} != {(rows, cfg.img_channels, R, R)}")
            if cfg.warp_depth_coor:
                assert src.shape[1] == 4, "warp_depth_coor requires depth channel in src"      # :644
            if cfg.super_res:
                assert conditioning_image is not None, "super_res mode requires a conditioning_image"   # :656
            has_cond = bool(cfg.super_res)
            if geometry is None:
                if not cfg.uncond and not (not self.dual_source):
                    raise TypeError("geometry is required (the reference multiplies None by an int here, :631)")
                geometry = torch.zeros(rows, cfg.source_label_dim, device=dev)
            handle = inject_features if isinstance(inject_features, FeatureHandle) else None
            if handle is not None:
                if handle.net is not self or handle.B != B or handle.version != self._prepared_fp:
                    raise ValueError("inject_features: this handle belongs to another net, batch size or weight version")
                if return_features:
                    prog = eng.program("features", B, has_cond, False, slot=handle.slot)
                    return [v.clone().permute(0, 3, 1, 2) for v in prog.view("features_out")]
            if return_features:
                if inject_features is not None:
                    return list(inject_features)
                if cfg.uncond:
                    mode = None
                else:
                    mode = "features"
            elif handle is not None:
                mode = "bound"
            elif inject_features is not None:
                mode = "inject"
            elif cfg.uncond:
                mode = "uncond"
            else:
                mode = "full"
            if mode is None:     # uncond net asked for features: the zero list of :727-736
                return [torch.zeros(rows, c, r, r, device=dev) for (c, r) in eng._feature_shapes()]
            prog = eng.program(mode, B, has_cond, bool(return_logvar), slot=handle.slot if handle is not None else 0)
            if mode in ("full", "features"):
                self._order_after_encoder(dev)

            def put(name, t, shape):
                self._put(prog, name, t, shape)

            put("sigma", sigma.reshape(-1), (rows,))
            put("geometry", geometry.reshape(rows, -1), (rows, cfg.source_label_dim))
            if "src" in prog.io:
                put("src", src, (rows, src_c, R, R))
            if "x" in prog.io:
                put("x", dst, (rows, cfg.img_channels, R, R))
            if "cond" in prog.io:
                cond = conditioning_image.to(torch.float32)
                if self.noisy_sr:                    # read per call like the reference's self.noisy_sr (:658): callers may set it
                    cond = cond + self.noisy_sr * torch.randn_like(cond)
                put("cond", cond, (B, cfg.img_channels, R, R))
            if mode == "inject":
                # copied on every call, like the reference's deepcopy (:665): a cache keyed on addresses goes stale when the
                # allocator hands a freed feature buffer's address to the next batch's features (a few MB against a whole forward)
                views = prog.view("features_in")
                if len(views) != len(inject_features):
                    raise ValueError(f"expected {len(views)} feature maps, got {len(inject_features)}")
                for v, f in zip(views, inject_features):
                    if tuple(f.shape) != (v.shape[0], v.shape[3], v.shape[1], v.shape[2]):
                        raise ValueError(f"feature shape {tuple(f.shape)} does not match {tuple(v.shape)} (NHWC)")
                    v.copy_(f.permute(0, 2, 3, 1))
            prog.plan.run()
            if mode == "features":
                # NCHW-shaped views over fresh NHWC storage (values and shapes as the reference's list)
                return [v.clone().permute(0, 3, 1, 2) for v in prog.view("features_out")]
            D = prog.view("D").clone()
            if return_logvar:
                return D, prog.view("logvar").clone().reshape(-1, 1, 1, 1)
            return D

    # debugging aid for tests: run once un-recorded, calling hook(name, tensor) after every block
    @torch.no_grad()
    def trace(self, src, dst, sigma, geometry, conditioning_image=None, hook=None):
        dev = dst.device
        with torch.cuda.device(dev):
            self._prepare(dev)
            eng = self._engine
            rm = 2 if self.dual_source else 1
            rows = dst.shape[0]
            B = rows // rm

            def fill(prog):
                prog.view("sigma").copy_(sigma.reshape(-1))
                prog.view("geometry").copy_(geometry.reshape(rows, -1))
                if "src" in prog.io:
                    prog.view("src").copy_(src)
                prog.view("x").copy_(dst)
                if "cond" in prog.io:
                    prog.view("cond").copy_(conditioning_image)
                torch.cuda.synchronize()

            eng.hook = hook
            try:
                prog = eng.program("uncond" if self.cfg.uncond else "full", B, bool(self.cfg.super_res), False, fill=fill)
                torch.cuda.synchronize()
                return prog.view("D").clone()
            finally:
                eng.hook = None
