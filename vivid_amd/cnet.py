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


def c_config(cfg: NetConfig, dual_source: bool = True, precision: str = "bf16x3") -> L.NetConfigC:
    c = L.NetConfigC(img_resolution=cfg.img_resolution, img_channels=cfg.img_channels, source_label_dim=cfg.source_label_dim,
                     target_label_dim=cfg.target_label_dim, model_channels=cfg.model_channels, num_levels=len(cfg.channel_mult),
                     num_blocks=cfg.num_blocks, num_attn_resolutions=len(cfg.attn_resolutions),
                     extra_attn=-1 if cfg.extra_attn is None else cfg.extra_attn,
                     channel_mult_noise=cfg.channel_mult_noise or 0, channel_mult_emb=cfg.channel_mult_emb or 0,
                     label_balance=cfg.label_balance, concat_balance=cfg.concat_balance, res_balance=cfg.res_balance,
                     attn_balance=cfg.attn_balance, clip_act=cfg.clip_act if cfg.clip_act is not None else 0.0, sigma_data=cfg.sigma_data,
                     logvar_channels=cfg.logvar_channels, super_res=int(cfg.super_res), no_time_enc=int(bool(cfg.no_time_enc)),
                     depth_input=int(cfg.depth_input), warp_depth_coor=int(cfg.warp_depth_coor), uncond=int(bool(cfg.uncond)),
                     dual_source=int(dual_source), noisy_sr=float(cfg.noisy_sr or 0.0) if cfg.super_res else 0.0)
    for i, v in enumerate(cfg.channel_mult):
        c.channel_mult[i] = v
    for i, v in enumerate(cfg.attn_resolutions):
        c.attn_resolutions[i] = v
    mean, std = geometry_stats(cfg.img_resolution)
    for i in range(20):
        c.geom_mean[i], c.geom_std[i] = float(mean[i]), float(std[i])
    f = [float(v) for v in cfg.resample_filter]
    if len(f) % 2 or not 2 <= len(f) <= 8:
        raise ValueError(f"resample_filter must have 2, 4, 6 or 8 taps (the reference asserts an even length, training/models.py:52); got {f}")
    c.resample_ntaps = len(f)
    for i, v in enumerate(f):
        c.resample_filter[i] = v
    if precision not in ("fp32", "bf16x3"):
        raise ValueError(f"precision must be 'fp32' or 'bf16x3', got {precision!r}")
    c.fp32 = int(precision == "fp32")
    return c


def _memcpy_d2d(dst, src, nbytes, stream):
    """hipMemcpyAsync(device -> device) on `stream` through the HIP runtime torch already loaded."""
    hip = _hip()
    rc = hip.hipMemcpyAsync(C.c_void_p(dst), C.c_void_p(src), C.c_size_t(nbytes), 3, C.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(f"hipMemcpyAsync failed ({rc})")


_HIP = []


def _hip():
    if not _HIP:
        import os
        for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6", os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "libamdhip64.so")):
            try:
                _HIP.append(C.CDLL(name))
                break
            except OSError:
                continue
        if not _HIP:
            raise RuntimeError("libamdhip64.so not found")
    return _HIP[0]


class CNet:
    def __init__(self, cfg: NetConfig, dual_source: bool = True, stream: int = 0, precision: str = "bf16x3"):
        self.cfg, self.dual, self.precision = cfg, dual_source, precision
        self._L = L.lib()
        self.ctx = L.Context(stream)
        h = C.c_void_p()
        self._cfg_c = c_config(cfg, dual_source, precision)
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

    def __call__(self, src, x, sigma, geometry=None, conditioning_image=None, cond_noise=None) -> torch.Tensor:
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
        ts = [None if t is None else t.to(torch.float32).contiguous() for t in (src, x, sigma, geometry, conditioning_image, cond_noise)]
        self.ctx.set_stream(torch.cuda.current_stream(x.device).cuda_stream)
        L.check(self._L.vh_net_run(self.handle, B, *[None if t is None else C.c_void_p(t.data_ptr()) for t in ts], C.c_void_p(out.data_ptr())), "vh_net_run")
        for t in ts:
            if t is not None:
                t.record_stream(torch.cuda.current_stream(x.device))
        return out

    # -- split evaluation (vh_net_encode / vh_net_run_bound): what vivid_amd.sampler's feature pipeline does, from the C side
    FULL, FEATURES, BOUND, INJECT = 0, 1, 2, 3

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

    def run_bound(self, slot: int, src, x, sigma, geometry=None, conditioning_image=None, cond_noise=None) -> torch.Tensor:
        B = x.shape[0] // (2 if self.dual else 1)
        self._record(self.FEATURES, slot, B, x.device)
        self._record(self.BOUND, slot, B, x.device)
        R = self.cfg.img_resolution
        out = torch.empty(B, 3, R, R, dtype=torch.float32, device=x.device)
        ts = [None if t is None else t.to(torch.float32).contiguous() for t in (src, x, sigma, geometry, conditioning_image, cond_noise)]
        self.ctx.set_stream(torch.cuda.current_stream(x.device).cuda_stream)
        L.check(self._L.vh_net_run_bound(self.handle, slot, B, *[None if t is None else C.c_void_p(t.data_ptr()) for t in ts], C.c_void_p(out.data_ptr())),
                "vh_net_run_bound")
        return out

    # -- the rest of NVPrecond.forward's protocol and the sampler, from the C side
    def feature_shapes(self) -> List[Tuple[int, int]]:
        out, c, r = [], C.c_int(), C.c_int()
        for i in range(self._L.vh_net_num_features(self.handle)):
            L.check(self._L.vh_net_feature_shape(self.handle, i, C.byref(c), C.byref(r)), "vh_net_feature_shape")
            out.append((c.value, r.value))
        return out

    def features(self, src, sigma, geometry) -> List[torch.Tensor]:
        """`net(src, ., sigma, geometry, return_features=True)`: the encoder's feature list as NCHW tensors (training/models.py:669-670)."""
        rows = src.shape[0]
        B = rows // (2 if self.dual else 1)
        self._record(self.FEATURES, 0, B, src.device)
        outs = [torch.empty(rows, c, r, r, dtype=torch.float32, device=src.device) for c, r in self.feature_shapes()]
        ptrs = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
        ts = [t.to(torch.float32).contiguous() for t in (src, sigma, geometry)]
        self.ctx.set_stream(torch.cuda.current_stream(src.device).cuda_stream)
        L.check(self._L.vh_net_features(self.handle, B, *[C.c_void_p(t.data_ptr()) for t in ts], ptrs), "vh_net_features")
        return outs

    def run_inject(self, src, x, sigma, geometry, features, conditioning_image=None, cond_noise=None) -> torch.Tensor:
        """`net(src, x, sigma, geometry, cond, inject_features=features)` (training/models.py:664-665)."""
        B = x.shape[0] // (2 if self.dual else 1)
        self._record(self.INJECT, 0, B, x.device)
        R = self.cfg.img_resolution
        out = torch.empty(B, 3, R, R, dtype=torch.float32, device=x.device)
        fs = [f.to(torch.float32).contiguous() for f in features]
        ptrs = (C.c_void_p * len(fs))(*[f.data_ptr() for f in fs])
        ts = [None if t is None else t.to(torch.float32).contiguous() for t in (src, x, sigma, geometry, conditioning_image, cond_noise)]
        self.ctx.set_stream(torch.cuda.current_stream(x.device).cuda_stream)
        L.check(self._L.vh_net_run_inject(self.handle, B, *[None if t is None else C.c_void_p(t.data_ptr()) for t in ts], ptrs, C.c_void_p(out.data_ptr())),
                "vh_net_run_inject")
        return out

    def logvar(self, sigma) -> torch.Tensor:
        B = sigma.shape[0] // (2 if self.dual else 1)
        out = torch.empty(B, dtype=torch.float32, device=sigma.device)
        s = sigma.to(torch.float32).contiguous()
        self.ctx.set_stream(torch.cuda.current_stream(sigma.device).cuda_stream)
        L.check(self._L.vh_net_logvar(self.handle, B, C.c_void_p(s.data_ptr()), C.c_void_p(out.data_ptr())), "vh_net_logvar")
        return out.reshape(-1, 1, 1, 1)

    def edm_sampler(self, src, noise, labels=None, gnet: Optional["CNet"] = None, conditioning_image=None, num_steps=32, sigma_min=0.002, sigma_max=80,
                    rho=7, guidance=1, S_churn=0, S_min=0, S_max=float("inf"), S_noise=1, randn=None, t_steps=None, pipeline=True,
                    guidance_overlap=-1) -> torch.Tensor:
        """vh_edm_sampler: vivid_amd.edm_sampler from the C side.  randn(n) -> a CUDA fp32 tensor of n standard-normal draws (on the current
        stream); t_steps: a host float tensor of num_steps + 1 levels to use instead of the library's own schedule arithmetic."""
        dev = noise.device
        rows = noise.shape[0]
        B = rows // (2 if self.dual else 1)
        if not self.cfg.uncond and (pipeline or self.cfg.no_time_enc):
            for slot in ((0,) if self.cfg.no_time_enc else (0, 1)):
                self._record(self.FEATURES, slot, B, dev)
                self._record(self.BOUND, slot, B, dev)
        if (not pipeline and not self.cfg.no_time_enc) or self.cfg.uncond:
            self._record(self.FULL, 0, B, dev)
        if gnet is not None and gnet is not self:
            gnet._record(gnet.FULL, 0, B, dev)
        keep = []

        def _randn(user, dst, n, stream):
            # draws made on torch's current stream (== the stream the library names: the contexts are set to it below), copied into the library's buffer there
            t = randn(n)
            keep.append(t)
            _memcpy_d2d(dst, t.data_ptr(), n * 4, stream)

        cfgc = L.SamplerConfigC(num_steps=num_steps, sigma_min=sigma_min, sigma_max=sigma_max, rho=rho, guidance=guidance, S_churn=S_churn, S_min=S_min,
                                S_max=S_max, S_noise=S_noise, guidance_overlap=guidance_overlap)
        ts_keep = None
        if t_steps is not None:
            ts_keep = (C.c_float * (num_steps + 1))(*[float(v) for v in t_steps])
            cfgc.t_steps = C.cast(ts_keep, C.POINTER(C.c_float))
        cb = L.RANDN_FN(_randn) if randn is not None else L.RANDN_FN()
        cfgc.randn = cb
        nbytes = self._L.vh_edm_sampler_workspace_bytes(self.handle, B)
        ws = torch.empty(nbytes // 4 + 4, dtype=torch.float32, device=dev)
        R = self.cfg.img_resolution
        out = torch.empty(B, 3, R, R, dtype=torch.float32, device=dev)
        ts = [None if t is None else t.to(torch.float32).contiguous() for t in (src, noise, labels, conditioning_image)]
        stream = torch.cuda.current_stream(dev).cuda_stream
        self.ctx.set_stream(stream)
        if gnet is not None:
            gnet.ctx.set_stream(stream)
        L.check(self._L.vh_edm_sampler(self.handle, gnet.handle if gnet is not None else None, C.byref(cfgc), B,
                                       *[None if t is None else C.c_void_p(t.data_ptr()) for t in ts], C.c_void_p(ws.data_ptr()), nbytes,
                                       C.c_void_p(out.data_ptr())), "vh_edm_sampler")
        torch.cuda.synchronize(dev)            # (ws, keep: the run reads them until it is done)
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
