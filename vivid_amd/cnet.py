"""The C-level whole-network entry points (``vh_net_*`` of include/vivid_hip.h) driven from Python.

``vh_net`` is the evaluation of ``NVPrecond.forward`` (training/models.py:628-749) for hosts that do not run Python: the library
itself generates the architecture, names the parameters with the reference's state_dict keys, prepares the weights, records the
evaluation and replays it.  :class:`CNet` is the thin binding a host would write - here used by the tests, which hold it bit for
bit against :class:`vivid_amd.NVPrecond` (whose Python engine emits the same op sequence) - and a worked example of the call order:
create -> bind every parameter -> prepare -> record per batch size -> run.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L
from .arch import NetConfig
from .geometry import geometry_stats


def c_config(cfg: NetConfig, dual_source: bool = True) -> L.NetConfigC:
    c = L.NetConfigC(img_resolution=cfg.img_resolution, img_channels=cfg.img_channels, source_label_dim=cfg.source_label_dim,
                     target_label_dim=cfg.target_label_dim, model_channels=cfg.model_channels, num_levels=len(cfg.channel_mult),
                     num_blocks=cfg.num_blocks, num_attn_resolutions=len(cfg.attn_resolutions),
                     extra_attn=-1 if cfg.extra_attn is None else cfg.extra_attn,
                     channel_mult_noise=cfg.channel_mult_noise or 0, channel_mult_emb=cfg.channel_mult_emb or 0,
                     label_balance=cfg.label_balance, concat_balance=cfg.concat_balance, res_balance=cfg.res_balance,
                     attn_balance=cfg.attn_balance, clip_act=cfg.clip_act if cfg.clip_act is not None else 0.0, sigma_data=cfg.sigma_data,
                     logvar_channels=cfg.logvar_channels, super_res=int(cfg.super_res), no_time_enc=int(bool(cfg.no_time_enc)),
                     depth_input=int(cfg.depth_input), warp_depth_coor=int(cfg.warp_depth_coor), uncond=int(bool(cfg.uncond)),
                     dual_source=int(dual_source))
    for i, v in enumerate(cfg.channel_mult):
        c.channel_mult[i] = v
    for i, v in enumerate(cfg.attn_resolutions):
        c.attn_resolutions[i] = v
    mean, std = geometry_stats(cfg.img_resolution)
    for i in range(20):
        c.geom_mean[i], c.geom_std[i] = float(mean[i]), float(std[i])
    if tuple(float(v) for v in cfg.resample_filter) != (1.0, 1.0):
        raise ValueError("vh_net implements the default resample_filter [1, 1] only")
    return c


class CNet:
    def __init__(self, cfg: NetConfig, dual_source: bool = True, stream: int = 0):
        self.cfg, self.dual = cfg, dual_source
        self._L = L.lib()
        self.ctx = L.Context(stream)
        h = C.c_void_p()
        self._cfg_c = c_config(cfg, dual_source)
        L.check(self._L.vh_net_create(self.ctx.handle, C.byref(self._cfg_c), C.byref(h)), "vh_net_create")
        self.handle = h
        self._keep: List[torch.Tensor] = []
        self._ws: Dict[int, torch.Tensor] = {}

    def params(self) -> List[Tuple[str, Tuple[int, ...]]]:
        """(state_dict key, shape) of every parameter / buffer the net reads, as the library names them."""
        out = []
        name, ndim, shape = C.c_char_p(), C.c_int(), (C.c_int * 4)()
        for i in range(self._L.vh_net_num_params(self.handle)):
            L.check(self._L.vh_net_param_info(self.handle, i, C.byref(name), C.byref(ndim), shape), "vh_net_param_info")
            out.append((name.value.decode(), tuple(shape[k] for k in range(ndim.value))))
        return out

    def load_state_dict(self, sd: Dict[str, torch.Tensor], device="cuda"):
        self._keep = []
        for key, shape in self.params():
            t = sd[key].detach().to(device=device, dtype=torch.float32).contiguous()
            if tuple(t.shape) != shape:
                raise ValueError(f"{key}: shape {tuple(t.shape)} != {shape}")
            self._keep.append(t)
            L.check(self._L.vh_net_bind_param(self.handle, key.encode(), C.c_void_p(t.data_ptr())), "vh_net_bind_param")
        nbytes = self._L.vh_net_prepared_bytes(self.handle)
        self._prepared = torch.empty(nbytes // 4 + 4, dtype=torch.float32, device=device)
        # the `.to(device)` copies above ran on torch's current stream: prepare on that stream, not on the one the context was created with
        self.ctx.set_stream(torch.cuda.current_stream(torch.device(device)).cuda_stream)
        L.check(self._L.vh_net_prepare(self.handle, C.c_void_p(self._prepared.data_ptr()), nbytes), "vh_net_prepare")
        self._ws = {}

    def workspace_bytes(self, batch: int) -> int:
        n = self._L.vh_net_workspace_bytes(self.handle, batch)
        if n == 0:
            L.check(-1, "vh_net_workspace_bytes")
        return n

    def __call__(self, src, x, sigma, geometry=None, conditioning_image=None) -> torch.Tensor:
        rm = 2 if self.dual else 1
        B = x.shape[0] // rm
        if B not in self._ws:
            nbytes = self.workspace_bytes(B)
            self._ws[B] = torch.empty(nbytes // 4 + 4, dtype=torch.float32, device=x.device)
            L.check(self._L.vh_net_record(self.handle, B, C.c_void_p(self._ws[B].data_ptr()), nbytes), "vh_net_record")
        R = self.cfg.img_resolution
        out = torch.empty(B, 3, R, R, dtype=torch.float32, device=x.device)

        def p(t):
            return None if t is None else C.c_void_p(t.to(torch.float32).contiguous().data_ptr())
        ts = [None if t is None else t.to(torch.float32).contiguous() for t in (src, x, sigma, geometry, conditioning_image)]
        self.ctx.set_stream(torch.cuda.current_stream(x.device).cuda_stream)
        L.check(self._L.vh_net_run(self.handle, B, *[None if t is None else C.c_void_p(t.data_ptr()) for t in ts], C.c_void_p(out.data_ptr())), "vh_net_run")
        for t in ts:
            if t is not None:
                t.record_stream(torch.cuda.current_stream(x.device))
        return out

    # -- split evaluation (vh_net_encode / vh_net_run_bound): what vivid_amd.sampler's feature pipeline does, from the C side
    FULL, FEATURES, BOUND = 0, 1, 2

    def _record(self, mode: int, slot: int, B: int, device):
        key = (mode, slot, B)
        if key not in self._ws:
            nbytes = self._L.vh_net_workspace_bytes_mode(self.handle, mode, B)
            if nbytes == 0:
                L.check(-1, "vh_net_workspace_bytes_mode")
            self._ws[key] = torch.empty(nbytes // 4 + 4, dtype=torch.float32, device=device)
            L.check(self._L.vh_net_record_mode(self.handle, mode, slot, B, C.c_void_p(self._ws[key].data_ptr()), nbytes), "vh_net_record_mode")

    def encode(self, slot: int, src, sigma, geometry):
        B = src.shape[0] // (2 if self.dual else 1)
        self._record(self.FEATURES, slot, B, src.device)
        ts = [t.to(torch.float32).contiguous() for t in (src, sigma, geometry)]
        self.ctx.set_stream(torch.cuda.current_stream(src.device).cuda_stream)
        L.check(self._L.vh_net_encode(self.handle, slot, B, *[C.c_void_p(t.data_ptr()) for t in ts]), "vh_net_encode")

    def run_bound(self, slot: int, src, x, sigma, geometry=None, conditioning_image=None) -> torch.Tensor:
        B = x.shape[0] // (2 if self.dual else 1)
        self._record(self.FEATURES, slot, B, x.device)
        self._record(self.BOUND, slot, B, x.device)
        R = self.cfg.img_resolution
        out = torch.empty(B, 3, R, R, dtype=torch.float32, device=x.device)
        ts = [None if t is None else t.to(torch.float32).contiguous() for t in (src, x, sigma, geometry, conditioning_image)]
        self.ctx.set_stream(torch.cuda.current_stream(x.device).cuda_stream)
        L.check(self._L.vh_net_run_bound(self.handle, slot, B, *[None if t is None else C.c_void_p(t.data_ptr()) for t in ts], C.c_void_p(out.data_ptr())),
                "vh_net_run_bound")
        return out

    def close(self):
        if getattr(self, "handle", None):
            self._L.vh_net_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
