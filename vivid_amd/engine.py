"""Host-side executor of the VIVID denoiser on libvivid_hip.so.

Walks the block table of ``vivid_amd/arch.py`` in the order the reference's
forward passes do (``UNetEncoder.forward`` training/models.py:536-570,
``XAttnUNet.forward`` :483-518, ``NVPrecond._forward_dualsource`` :628-689 and
the single-source forward :691-749) and emits C-ABI ops.  One denoiser
evaluation for a given (mode, batch) is emitted ONCE into a ``vh_plan`` over a
private workspace and replayed on later calls; only inputs are copied in.

PyTorch is used for device memory and the stream only.  All activations live
in one fp32 workspace tensor as NHWC buffers handed out by :class:`Arena`.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from .arch import BlockSpec, NetConfig, UNetSpec, unet_spec
from .geometry import geometry_stats

LOG2E = 1.4426950408889634


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class Arena:
    """First-fit allocator over one flat fp32 buffer (offsets in floats, 64-float granules).
    Execution is stream-ordered, so a released range may be reused by the next op."""

    GRAN = 64

    def __init__(self):
        self.free: List[List[int]] = [[0, 1 << 62]]
        self.peak = 0
        self.base_ptr = 0          # set once the backing tensor exists

    def alloc(self, n: int) -> int:
        n = _round_up(max(n, 1), self.GRAN)
        for i, (off, size) in enumerate(self.free):
            if size >= n:
                if size == n:
                    self.free.pop(i)
                else:
                    self.free[i] = [off + n, size - n]
                self.peak = max(self.peak, off + n)
                return off
        raise MemoryError("arena exhausted")

    def release(self, off: int, n: int):
        n = _round_up(max(n, 1), self.GRAN)
        self.free.append([off, n])
        self.free.sort()
        merged = [self.free[0]]
        for o, s in self.free[1:]:
            if merged[-1][0] + merged[-1][1] == o:
                merged[-1][1] += s
            else:
                merged.append([o, s])
        self.free = merged


@dataclass
class Buf:
    """A workspace range viewed as [rows, h, w, c] (NHWC) or any flat shape."""
    off: int
    shape: Tuple[int, ...]
    arena: Arena

    @property
    def numel(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n

    @property
    def ptr(self) -> int:
        return self.arena.base_ptr + 4 * self.off

    def view(self, backing: torch.Tensor) -> torch.Tensor:
        return backing[self.off:self.off + self.numel].view(self.shape)


class Ghost:
    """A tensor that was never materialised in fp32: its only reader takes the S8 forms its producer wrote (vh_s8_sink)."""

    def __init__(self, shape):
        self.shape = tuple(shape)
        self.off = None


@dataclass
class Weight:
    wt: torch.Tensor
    cin_pad: int
    k_pad: int
    cout: int
    taps: int


class Program:
    """A recorded denoiser evaluation: plan + workspace + named I/O buffers."""

    def __init__(self, plan, backing, io: Dict[str, object]):
        self.plan, self.backing, self.io = plan, backing, io

    def view(self, name):
        b = self.io[name]
        if isinstance(b, list):
            return [x.view(self.backing) for x in b]
        return b.view(self.backing)


class Engine:
    """Executor bound to one NVPrecond instance (its config and parameters)."""

    def __init__(self, cfg: NetConfig, dual_source: bool = True, precision: str = "fp32"):
        if precision not in ("fp32", "bf16x3"):
            raise ValueError(f"precision must be 'fp32' or 'bf16x3', got {precision!r}")
        self.precision = precision
        self.x3 = precision == "bf16x3"      # convs/attention on the bf16 hi/lo split MFMA path
        import os
        self.glds = self.x3 and os.environ.get("VIVID_CONV_KERNEL", "glds256") != "tile128"
        # replay each recorded evaluation as one hipGraph: measured null on MI355X (15.1 vs 15.2 ms at batch 1, 376.7 vs
        # 377.4 ms at the headline workload: replay is GPU-bound, not launch-bound), so opt-in only
        self.use_graph = os.environ.get("VIVID_HIPGRAPH", "0") == "1"
        self.conv_stagger = 0                 # vh_conv scheduling hint: 0 = library default (the 512x128 tile staggers, the others do not)
        # attn_qkv / x_attn_kv write q, k, v^T from their own epilogue (glds kernel)
        self.fuse_qkv = os.environ.get("VIVID_FUSE_QKV", "1") != "0"
        # decoder blocks: conv_res1 and conv_skip as one GEMM (a 1-tap tail segment of the 3x3 K loop)
        self.fuse_skip = os.environ.get("VIVID_FUSE_SKIP", "1") != "0"
        # plain encoder blocks: residual of conv_res1 as x * scale[pixel] instead of a stored pixel-normalised copy
        self.scale_residual = os.environ.get("VIVID_SCALE_RESIDUAL", "1") != "0"
        # decoder blocks: the two halves of `mp_silu(mp_cat(x, skip))` (training/models.py:78-84, :174) written in S8 form by the convolutions that
        # PRODUCE x and skip (vh_s8_sink, patch-resident kernel) instead of by a vh_split pass over their fp32 results
        # (VIVID_FUSE_CONCAT: 2 both halves, 1 the x half only, 0 never; unset = the library's knob default, 0: measured +0.35 % (C2) / +1.4 % (C4)
        #  whole-step at -3..5 % of the convolutions' rate - the bytes move into epilogues that are exposed.  The knob is process-wide and is what
        #  the C-level walk vh_net_* reads, so the two walks stay identical.)
        self.fuse_concat = int(os.environ["VIVID_FUSE_CONCAT"]) if "VIVID_FUSE_CONCAT" in os.environ else None
        self._cat: Dict[int, dict] = {}
        # sampler's split evaluation: cross-attention K / V computed with the features (once per noise level) instead of in every UNet call
        self.hoist_kv = os.environ.get("VIVID_HOIST_KV", "1") != "0"
        self.cfg = cfg
        self.std_filter = tuple(float(v) for v in cfg.resample_filter) == (1.0, 1.0)
        self.dual = dual_source
        self.nsrc = 2 if dual_source else 1
        self.enc_spec: Optional[UNetSpec] = None if cfg.uncond else unet_spec(cfg, role="encoder")
        self.unet_spec: UNetSpec = unet_spec(cfg, role="unet")
        if self.x3:
            for sp in (self.enc_spec, self.unet_spec):
                for blk in (sp.enc + sp.dec if sp is not None else []):
                    if blk.cout % 32:
                        raise ValueError(f"precision='bf16x3' needs channel counts that are multiples of 32 "
                                         f"({blk.name} has {blk.cout}); use precision='fp32'")
        self.W: Dict[str, Weight] = {}
        self.embW: Dict[str, Tuple[torch.Tensor, Dict[str, int], int]] = {}
        self.bufs: Dict[str, torch.Tensor] = {}
        self.programs: Dict[tuple, Program] = {}
        self.ctx: Optional[L.Context] = None
        self.device = None
        self._A: Optional[Arena] = None
        self._emit = False
        self.hook = None
        self.oplog: List[str] = []

    # ------------------------------------------------------------------ context / weights
    def _ensure_ctx(self, device):
        if self.ctx is None or self.device != device:
            self.device = device
            self.ctx = L.Context(torch.cuda.current_stream(device).cuda_stream)
        else:
            self.ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)

    def prepare_weights(self, params: Dict[str, torch.Tensor], device):
        """K1 once per weight version: normalise, scale by gain/sqrt(fan_in), re-lay out
        (the reference repeats this on every forward, training/models.py:115-120)."""
        self._ensure_ctx(device)
        self.zeros = torch.zeros(16384, dtype=torch.float32, device=device)     # 64 KiB zero page for vh_conv
        # split-K partial sums (64 MiB each): one for the encoder's launches, one for the UNet's - the two halves of a split evaluation
        # (NVPrecond.encode_features / forward(inject_features=handle)) run on different streams at the same time
        self.scratch_enc = torch.empty(16 << 20, dtype=torch.float32, device=device) if self.glds else None
        self.scratch_unet = torch.empty(16 << 20, dtype=torch.float32, device=device) if self.glds else None
        self.scratch = self.scratch_unet
        self.W.clear()
        self.embW.clear()
        self.programs.clear()
        self._params = params
        for prefix, spec in (("encoder.", self.enc_spec), ("unet.", self.unet_spec)):
            if spec is None:
                continue
            self._prep_linear(prefix + "emb_noise.weight")
            if spec.label_dim:
                self._prep_linear(prefix + "emb_label.weight")
            blocks = [b for b in spec.live_blocks() if b.kind == "block"]
            total = sum(b.cout for b in blocks)
            kpad = _round_up(spec.cemb, 32)
            wt = torch.zeros(kpad // 4 * total * 4, dtype=torch.float32, device=device)
            cols, c0 = {}, 0
            for grp, b in [("enc", b) for b in spec.enc] + [("dec", b) for b in spec.dec]:
                if not b.live:
                    continue
                p = f"{prefix}{grp}.{b.name}."
                if b.kind == "conv":
                    self._prep_conv(p + "weight", 9)
                    continue
                self._prep_conv(p + "conv_res0.weight", 9)
                if self._fused_skip(b):
                    self._prep_fused_res1_skip(p, b)
                else:
                    self._prep_conv(p + "conv_res1.weight", 9)
                    if b.cin != b.cout:
                        self._prep_conv(p + "conv_skip.weight", 1)
                if b.heads:
                    fused = self._qkv_fused(b)
                    self._prep_conv(p + "attn_qkv.weight", 1, qkv_perm=(3, b.cout // b.heads) if fused else None)
                    self._prep_conv(p + "attn_proj.weight", 1)
                    if b.xattn:
                        self._prep_conv(p + "x_attn_kv.weight", 1, qkv_perm=(2, b.cout // b.heads) if fused else None)
                w = params[p + "emb_linear.weight"]
                a = L.PrepWeightArgs(w=w.data_ptr(), cout=b.cout, cin=spec.cemb, taps=1, cin_pad=_round_up(spec.cemb, 4),
                                     k_pad=kpad, gain_ptr=params[p + "emb_gain"].data_ptr(), gain_value=1.0,
                                     wt=wt.data_ptr(), dst_col0=c0, dst_cols=total, split=0)
                self.ctx.call("vh_prep_weight", a)
                cols[p] = c0
                c0 += b.cout
            self.embW[prefix] = (wt, cols, total)
            if spec.out_channels:
                self._prep_conv(prefix + "out_conv.weight", 9, gain=params[prefix + "out_gain"])
        self._prep_linear("logvar_linear.weight")

    def _qkv_fused(self, b: BlockSpec) -> bool:
        """attn_qkv / x_attn_kv write the attention operands from their own epilogue (VH_EPI_QKV) instead of an fp32
        tensor that vh_qkv_split_x3 reads back: bf16x3 glds path, 64- or 32-channel heads, 32 | pixels per image."""
        if not (self.x3 and self.glds and self.fuse_qkv and b.heads):
            return False
        return b.cout % 32 == 0 and b.cin % 32 == 0 and b.cout // b.heads in (32, 64) and (b.res * b.res) % 32 == 0

    def _fused_skip(self, b: BlockSpec) -> bool:
        """Decoder blocks with a skip convolution, bf16x3 glds path: `x = mp_sum(conv_skip(x_cat), conv_res1(y), t)` (training/models.py:184-186) is
        ONE GEMM - K = 9*Cout (y, 3x3) + Cin (x_cat, 1 tap), the mp_sum coefficients folded into the two weights - instead of a 1x1 launch, its
        fp32 output and a residual read (vh_conv_args.src1 as a tail segment)."""
        return (self.x3 and self.glds and self.fuse_skip and b.kind == "block" and b.flavor == "dec" and b.cin != b.cout
                and b.cout % 32 == 0 and b.cin % 32 == 0)

    def _tail_f32(self, b: BlockSpec, rows: int, srcs) -> bool:
        """Does the fused conv_res1 + conv_skip launch of decoder block b read its tail segment from the fp32 tensors x and skip themselves
        (vh_conv_args.tail_f32)?  The library answers (vh_conv_takes_patch: the patch-resident kernel's size rule); then vh_split writes the
        mp_silu form of the concat only - a third of its bytes less."""
        if not (self._fused_skip(b) and all(s_.shape[-1] % 32 == 0 for s_, _ in srcs)):
            return False
        q = L.ConvArgs(src0=16, src1=16, c0=b.cout, c1=srcs[0][0].shape[-1], src2=16 if len(srcs) > 1 else None, c2=srcs[1][0].shape[-1] if len(srcs) > 1 else 0,
                       tail_f32=1, rows=rows, h=b.res, w=b.res, up=0, taps=9, cout=b.cout, prec=1, kernel=1, epi=L_EPI_STORE, out=16)
        return L.lib().vh_conv_takes_patch(C.byref(q)) == 1

    def _src_f32(self, b: BlockSpec, rows: int, srcs, up: int) -> bool:
        """Does conv_res0 of decoder block b stage its patches from the fp32 tensors themselves (vh_conv_args.src_f32: mp_cat weights, mp_silu and the
        bf16 split applied in the kernel - no vh_split pass)?  The library answers (vh_conv_takes_patch)."""
        if not (self.x3 and self.glds and self.hook is None and all(s_.shape[-1] % 32 == 0 for s_, _ in srcs)):
            return False
        q = L.ConvArgs(src0=16, src1=16 if len(srcs) > 1 else None, c0=srcs[0][0].shape[-1], c1=srcs[1][0].shape[-1] if len(srcs) > 1 else 0, src_f32=1,
                       rows=rows, h=b.res, w=b.res, up=up, taps=9, pro=L_PRO_SILU, cout=b.cout, prec=1, kernel=1, epi=L_EPI_SCALE_SILU, out_s8=16)
        return L.lib().vh_conv_takes_patch(C.byref(q)) == 1

    def _prep_fused_res1_skip(self, p: str, b: BlockSpec):
        ta, tb = self._mp_sum_coeffs(self.cfg.res_balance)
        w1, ws = self._params[p + "conv_res1.weight"], self._params[p + "conv_skip.weight"]
        cout, c1 = b.cout, _round_up(b.cin, 32)
        k_pad = 9 * cout + c1
        wt = torch.empty(k_pad // 4 * cout * 4, dtype=torch.float32, device=w1.device)
        if w1.numel():
            for w, taps, cin, cin_pad, gain, k_off in ((w1, 9, cout, cout, tb, 0), (ws, 1, b.cin, c1, ta, 9 * cout)):
                self.ctx.call("vh_prep_weight", L.PrepWeightArgs(w=w.data_ptr(), cout=cout, cin=cin, taps=taps, cin_pad=cin_pad, k_pad=taps * cin_pad,
                                                                 gain_ptr=None, gain_value=gain, wt=wt.data_ptr(), dst_col0=0, dst_cols=cout, split=2,
                                                                 k_off=k_off, k_stride=k_pad))
        self.W[p + "conv_res1+skip"] = Weight(wt, cout, k_pad, cout, 9)

    def _prep_conv(self, key: str, taps: int, gain: Optional[torch.Tensor] = None, qkv_perm: Optional[Tuple[int, int]] = None):
        w = self._params[key]
        if qkv_perm and w.numel():
            # output channel (head*D + d)*nj + j  ->  (head*nj + j)*D + d: one (head, j) per D-column accumulator slab
            nj, D = qkv_perm
            heads = w.shape[0] // (D * nj)
            w = w.view(heads, D, nj, *w.shape[1:]).transpose(1, 2).contiguous().view(w.shape)
        cout, cin = w.shape[0], w.shape[1]
        # 2-D (linear) weights feed embed_k/linear_k: never split.  bf16x3 convs: split 2 = [cout][K] for the glds kernel
        split = (2 if self.glds else 1) if (self.x3 and w.ndim == 4) else 0
        cin_pad = _round_up(cin, 32 if w.ndim == 4 else 4)       # conv K-tiles are 32 channels of one tap
        k_pad = _round_up(taps * cin_pad, 32)
        wt = torch.empty(k_pad // 4 * cout * 4, dtype=torch.float32, device=w.device)
        a = L.PrepWeightArgs(w=w.data_ptr(), cout=cout, cin=cin, taps=taps, cin_pad=cin_pad, k_pad=k_pad,
                             gain_ptr=gain.data_ptr() if gain is not None else None, gain_value=1.0,
                             wt=wt.data_ptr(), dst_col0=0, dst_cols=cout, split=split)
        self.ctx.call("vh_prep_weight", a)
        self.W[key] = Weight(wt, cin_pad, k_pad, cout, taps)

    def _prep_linear(self, key: str):
        self._prep_conv(key, 1)

    def measure_workspace(self, mode: str, B: int, has_cond: bool = False, want_logvar: bool = False) -> int:
        """Workspace bytes one recorded evaluation needs (dry walk, no GPU, no launches)."""
        saved = (self.W, self.embW, getattr(self, "_params", None))
        if not self.W:
            from .weights import state_dict_shapes
            shapes = state_dict_shapes(self.cfg)
            dummy = torch.empty(0)
            self._params = {k: dummy for k in shapes}
            self.W = {}
            for k, shp in shapes.items():
                if k.endswith("weight"):
                    taps = 9 if len(shp) == 4 and shp[-1] == 3 else 1
                    cin_pad = _round_up(shp[1], 32 if len(shp) == 4 else 4)
                    self.W[k] = Weight(dummy, cin_pad, _round_up(taps * cin_pad, 32), shp[0], taps)
            self.embW = {}
            for prefix, spec in (("encoder.", self.enc_spec), ("unet.", self.unet_spec)):
                if spec is None:
                    continue
                cols, c0 = {}, 0
                for grp, b in [("enc", b) for b in spec.enc] + [("dec", b) for b in spec.dec]:
                    if b.live and b.kind == "block":
                        cols[f"{prefix}{grp}.{b.name}."] = c0
                        c0 += b.cout
                        if self._fused_skip(b):
                            self.W[f"{prefix}{grp}.{b.name}.conv_res1+skip"] = Weight(dummy, b.cout, 9 * b.cout + _round_up(b.cin, 32), b.cout, 9)
                self.embW[prefix] = (dummy, cols, c0)
        try:
            self._A, self._emit, self._backing = Arena(), False, None
            self._walk(mode, B, has_cond, want_logvar)
            return 4 * self._A.peak
        finally:
            self.W, self.embW, self._params = saved

    # ------------------------------------------------------------------ emission helpers
    def _alloc(self, *shape) -> Buf:
        n = 1
        for s in shape:
            n *= s
        return Buf(self._A.alloc(n), tuple(shape), self._A)

    def _free(self, b: Optional[Buf]):
        if b is not None and not isinstance(b, Ghost):
            self._A.release(b.off, b.numel)

    def _call(self, name, args, desc: str = ""):
        if self._emit:
            self.ctx.call(name, args)
            self.oplog.append(f"{name[3:]} {desc}")

    def _tap(self, name: str, buf: Buf):
        if self._emit and self.hook is not None and self._backing is not None:
            torch.cuda.synchronize()
            self.hook(name, buf.view(self._backing).clone())

    def _conv(self, srcs: Sequence[Tuple[Buf, float]], W: Weight, rows, h, w, *, up=0, pro=0, epi=0,
              cvec: Optional[Tuple[int, int]] = None, res: Optional[Buf] = None, res_up=0, res_scale: Optional[Buf] = None,
              ta=0.0, tb=0.0, clip=0.0, out: Optional[Buf] = None, prec=0, s8_only=False, also_s8=False, qkv=None,
              sink_plan: Optional[Tuple[int, str]] = None, fp32_optional=False, tail_f32=False, src_f32=False):
        """prec=1: srcs[0] is an S8 (bf16 hi/lo) buffer.  tail_f32: srcs[1:] are the fp32 tensors of the 1-tap tail segment with their mp_cat weights
        (vh_conv_args.tail_f32: the bf16 split happens while the tail is staged; no raw S8 concat exists).  src_f32 (with prec=1): srcs are the 1-2
        fp32 tensors of the input concat with their mp_cat weights and `pro` applies, as in fp32 mode - the patch-resident kernel splits them while
        it stages its patches (vh_conv_args.src_f32).  s8_only: the result is written only as S8;
        also_s8: fp32 and S8 copies are both written and (out, out_s8) is returned.
        qkv (L.QkvEpilogue): the result goes straight into attention operand buffers (VH_EPI_QKV); nothing is returned."""
        out_s8 = None
        if qkv is not None:
            epi = L_EPI_QKV
        s0, sc0 = srcs[0]
        s1, sc1 = srcs[1] if len(srcs) > 1 else (None, 1.0)
        s2, sc2 = srcs[2] if len(srcs) > 2 else (None, 0.0)
        assert s2 is None or tail_f32
        assert not (src_f32 and tail_f32)
        # sink_plan = (dec block index, "x" | "skip"): this result is one half of that block's concat input.  If the launch takes the patch-resident
        # kernel (the library's own rule: vh_conv_takes_patch), it writes the S8 forms itself (fp32_optional: and nothing else reads the fp32 form)
        sinks = None
        if sink_plan is not None and prec and self.glds and self.hook is None and qkv is None:
            q = L.ConvArgs(src0=16, src1=16 if s1 is not None else None, c0=s0.shape[-1], c1=s1.shape[-1] if s1 is not None else 0, rows=rows, h=h, w=w,
                           up=up, taps=W.taps, cout=W.cout, prec=prec, kernel=1, epi=epi, res_up=res_up, tail_f32=int(tail_f32),
                           src2=16 if s2 is not None else None, c2=s2.shape[-1] if s2 is not None else 0)
            if L.lib().vh_conv_takes_patch(C.byref(q)) == 1:
                sinks = self._cat_sinks(*sink_plan)
        skip_fp32 = sinks is not None and fp32_optional and not also_s8 and not s8_only
        if (s8_only or also_s8) and qkv is None:
            out_s8 = self._alloc(rows, h, w, W.cout)
        if not s8_only and out is None and qkv is None and not skip_fp32:
            out = self._alloc(rows, h, w, W.cout)
        a = L.ConvArgs(src0=s0.ptr, src1=s1.ptr if s1 is not None else None,
                       c0=s0.shape[-1], c1=s1.shape[-1] if s1 is not None else 0,
                       scale0=sc0, scale1=sc1, rows=rows, h=h, w=w, up=up, taps=W.taps, pro=pro,
                       wt=W.wt.data_ptr(), cin_pad=W.cin_pad, k_pad=W.k_pad,
                       zeros=self.zeros.data_ptr() if getattr(self, "zeros", None) is not None else None, zeros_bytes=65536, cout=W.cout,
                       scratch=self.scratch.data_ptr() if getattr(self, "scratch", None) is not None else None,
                       scratch_floats=self.scratch.numel() if getattr(self, "scratch", None) is not None else 0,
                       out=out.ptr if out is not None else None, out_s8=out_s8.ptr if out_s8 is not None else None,
                       out_s8_c=W.cout if out_s8 is not None else 0, prec=prec, kernel=1 if (prec and self.glds) else 0, epi=epi,
                       cvec=cvec[0] if cvec else None, cvec_ld=cvec[1] if cvec else 0,
                       res=res.ptr if res is not None else None, res_up=res_up, res_scale=res_scale.ptr if res_scale is not None else None,
                       ta=ta, tb=tb, clip=clip,
                       qkv=C.addressof(qkv) if qkv is not None else None,
                       stagger=self.conv_stagger if (prec and self.glds) else 0,
                       tail_f32=int(tail_f32), src2=s2.ptr if s2 is not None else None, c2=s2.shape[-1] if s2 is not None else 0, scale2=sc2,
                       src_f32=int(src_f32))
        for i, (buf, ct, off, scale, silu) in enumerate(sinks or []):
            a.sink[i] = L.S8Sink(ptr=buf.ptr, c_total=ct, c_off=off, scale=scale, silu=silu)
        self._call("vh_conv", a, f"{W.taps}tap rows={rows} {h}x{w} cin={a.c0 + a.c1 + a.c2} cout={W.cout} up={up} pro={pro} epi={epi} prec={prec}"
                   + (f" sinks={len(sinks)}" if sinks else "") + (" tail=fp32" if tail_f32 else "") + (" src=fp32" if src_f32 else ""))
        if skip_fp32:
            return Ghost((rows, h, w, W.cout))
        if qkv is not None:
            return None
        if also_s8:
            return out, out_s8
        return out_s8 if s8_only else out

    # ---- decoder concat inputs written by their producers -------------------------------------------------------------------------
    FUSE_CONCAT_DEFAULT = 0          # == the library's default of knob "fuse_concat" (csrc/api.hip); see DESIGN.md 3 for the A/B that set it

    def _fuse_mode(self) -> int:
        if self.fuse_concat is None:
            return self.FUSE_CONCAT_DEFAULT
        L.set_knob("fuse_concat", self.fuse_concat)
        return self.fuse_concat

    def _cat_plan(self, spec: UNetSpec, rows: int):
        """Per decoder block that takes a skip: which encoder entry it pops (UNet.forward's skip stack, training/models.py:507-510), the two
        channel counts, the mp_cat weights (:78-84) and whether conv_skip needs the raw form too.  Returns (plan by dec index, consumer dec
        index by enc entry index)."""
        plan, consumer = {}, {}
        k = len(spec.enc) - 1
        cprev = spec.enc[-1].cout
        t = self.cfg.concat_balance
        for j, b in enumerate(spec.dec):
            if not b.live:
                break
            if b.takes_skip:
                Nb = spec.enc[k].cout
                Na = b.cin - Nb
                Cc = math.sqrt((Na + Nb) / ((1 - t) ** 2 + t ** 2))
                plan[j] = dict(rows=rows, R=b.res, Na=Na, Nb=Nb, sc0=Cc / math.sqrt(Na) * (1 - t), sc1=Cc / math.sqrt(Nb) * t,
                               raw=b.cin != b.cout, cs=None, craw=None, x_done=False, skip_done=False,
                               ok=Na % 32 == 0 and Nb % 32 == 0 and b.resample != "up" and Na == cprev)
                consumer[k] = j
                k -= 1
            cprev = b.cout
        return plan, consumer

    def _cat_sinks(self, j: int, half: str):
        """The sink list of the convolution that produces half `half` of dec block j's concat input; allocates the S8 concat tensors."""
        st = self._cat[j]
        Ct = st["Na"] + st["Nb"]
        if st["cs"] is None:
            st["cs"] = self._alloc(st["rows"], st["R"], st["R"], Ct)
            if st["raw"]:
                st["craw"] = self._alloc(st["rows"], st["R"], st["R"], Ct)
        off, scale = (0, st["sc0"]) if half == "x" else (st["Na"], st["sc1"])
        st[half + "_done"] = True
        out = [(st["cs"], Ct, off, scale, 1)]
        if st["raw"]:
            out.append((st["craw"], Ct, off, scale, 0))
        return out

    def _split_half(self, src: Buf, scale: float, st: dict, off: int):
        """vh_split of ONE half of a concat input into its channel range of the S8 concat tensors (the other half came from a sink)."""
        rows, h, w, c = src.shape
        Ct = st["Na"] + st["Nb"]
        self._call("vh_split", L.SplitArgs(src0=src.ptr, src1=None, c0=c, c1=0, scale0=scale, scale1=1.0, pro=L_PRO_SILU, npix=rows * h * w, c_pad=c,
                                          out=st["cs"].ptr, out_raw=st["craw"].ptr if st["craw"] is not None else None, out_c_total=Ct, out_c_off=off),
                   f"rows={rows} {h}x{w} c={c} raw={int(st['craw'] is not None)} half@{off}/{Ct}")

    def _split(self, srcs: Sequence[Tuple[Buf, float]], pro: int, raw_too: bool = False):
        """fp32 NHWC (1-2 sources, mp_cat weights) -> S8 at the sources' resolution, channels padded to 32.
        pro = prologue of the main output; raw_too additionally returns the un-activated split (conv_skip's input)."""
        s0, sc0 = srcs[0]
        s1, sc1 = srcs[1] if len(srcs) > 1 else (None, 1.0)
        rows, h, w = s0.shape[:3]
        ctot = s0.shape[-1] + (s1.shape[-1] if s1 is not None else 0)
        cpad = _round_up(ctot, 32)
        out = self._alloc(rows, h, w, cpad)
        raw = self._alloc(rows, h, w, cpad) if raw_too else None
        self._call("vh_split", L.SplitArgs(src0=s0.ptr, src1=s1.ptr if s1 is not None else None, c0=s0.shape[-1],
                                          c1=s1.shape[-1] if s1 is not None else 0, scale0=sc0, scale1=sc1, pro=pro,
                                          npix=rows * h * w, c_pad=cpad, out=out.ptr, out_raw=raw.ptr if raw is not None else None),
                   f"rows={rows} {h}x{w} c={ctot} raw={int(raw_too)}")
        return (out, raw) if raw_too else out

    def _resample(self, x: Buf, up: bool) -> Buf:
        """resample() with a non-default filter (training/models.py:48-61) as its own launch; the default [1,1] is fused
        into vh_pixnorm (down) and vh_conv (up) instead."""
        rows, h, w, c = x.shape
        f = [float(v) for v in self.cfg.resample_filter]
        if len(f) % 2 or not 2 <= len(f) <= 8:
            raise ValueError(f"resample_filter must have 2, 4, 6 or 8 taps (the reference asserts an even length, :52); got {f}")
        out = self._alloc(rows, h * 2, w * 2, c) if up else self._alloc(rows, h // 2, w // 2, c)
        a = L.ResampleArgs(inp=x.ptr, out=out.ptr, rows=rows, h=h, w=w, c=c, up=1 if up else 0, ntaps=len(f))
        tot = sum(f)
        for i, v in enumerate(f):
            a.taps[i] = v / tot
        self._call("vh_resample", a, f"rows={rows} {h}x{w} c={c} up={int(up)}")
        return out

    def _mp_sum_coeffs(self, t: float):
        n = math.sqrt((1 - t) ** 2 + t ** 2)
        return (1 - t) / n, t / n

    # ------------------------------------------------------------------ one block
    def _block(self, prefix: str, grp: str, b: BlockSpec, rows: int, x: Buf, skip: Optional[Buf],
               cvec_all: Buf, cols: Dict[str, int], total_cols: int,
               feat: Optional[Buf], feat_s8: Optional[Buf], n_zero: float, want_s8: bool = False,
               cat_j: Optional[int] = None, out_sink: Optional[Tuple[int, str]] = None, fp32_optional: bool = False,
               kv_pre: Optional[Tuple[Buf, Buf]] = None, s8_final: bool = False):
        """Block.forward :165-206 / XAttnBlock.forward :251-315.  x is the block input (before
        resampling); returns (block output fp32, its S8 copy or None).  Neither x nor skip is released here.
        In bf16x3 mode every conv reads an S8 (bf16 hi/lo) tensor written by the op that produced it."""
        cfg = self.cfg
        p = f"{prefix}{grp}.{b.name}."
        R = b.res
        cv = (cvec_all.ptr + 4 * cols[p], total_cols)
        ta, tb = self._mp_sum_coeffs(cfg.res_balance)
        clip = float(cfg.clip_act) if cfg.clip_act is not None else 0.0
        clip_res = 0.0 if b.heads else clip
        has_skip_conv = b.cin != b.cout
        x3 = self.x3 and b.cout % 32 == 0 and b.cin % 32 == 0
        P1 = 1 if x3 else 0
        C = b.cout
        D = C // b.heads if b.heads else 0
        ax3 = x3 and D in (32, 64)                      # attention (and its 1x1 convs) on the bf16x3 path
        out_s8 = None
        res1_s8 = bool(b.heads) and ax3                 # conv_res1's result also feeds attn_qkv -> S8 copy
        fin_s8 = want_s8 and x3 and not b.heads         # no attention: conv_res1's result is the block output
        # the UNet's last block: out_conv is its only reader and reads S8 - conv_res1 writes that form alone (no fp32 tensor, no vh_split pass)
        fin_only = s8_final and x3 and not b.heads and b.flavor == "dec" and self.hook is None
        if b.heads or not x3:
            out_sink, fp32_optional = None, False       # (the block's last op is attn_proj, a 1x1 convolution: no sinks there)
        if b.flavor == "enc":
            xs = self._alloc(rows, R, R, C) if x3 else None      # S8 of mp_silu(xn): conv_res0's input
            xs_ptr = xs.ptr if xs is not None else None
            res_scale = res_src = None
            if b.resample == "down" and self.std_filter:
                xn = self._alloc(rows, R, R, C)
                self._call("vh_pixnorm", L.PixnormArgs(inp=x.ptr, out=xn.ptr, rows=rows, h=R, w=R, c=C, pool=1, norm=1, out_s8=xs_ptr))
            elif b.resample == "down":                           # general FIR filter (:48-59), then the plain pixel norm
                xn = self._resample(x, up=False)
                self._call("vh_pixnorm", L.PixnormArgs(inp=xn.ptr, out=xn.ptr, rows=rows, h=R, w=R, c=C, pool=0, norm=1, out_s8=xs_ptr))
            elif has_skip_conv:
                if x3:
                    xr = self._split([(x, 1.0)], 0)
                    xn = self._conv([(xr, 1.0)], self.W[p + "conv_skip.weight"], rows, R, R, prec=1)
                    self._free(xr)
                else:
                    xn = self._conv([(x, 1.0)], self.W[p + "conv_skip.weight"], rows, R, R)
                self._call("vh_pixnorm", L.PixnormArgs(inp=xn.ptr, out=xn.ptr, rows=rows, h=R, w=R, c=C, pool=0, norm=1, out_s8=xs_ptr))
            elif x3 and self.scale_residual:
                # the normalised tensor is never materialised: its S8 mp_silu form feeds conv_res0, and its other reader - the residual of
                # conv_res1 - takes x * scale[pixel] (4 of this pass's 12 bytes per element saved)
                xn, res_src = None, x
                res_scale = self._alloc(rows, R, R, 1)
                self._call("vh_pixnorm", L.PixnormArgs(inp=x.ptr, out=None, rows=rows, h=R, w=R, c=C, pool=0, norm=1, out_s8=xs_ptr, scale_out=res_scale.ptr))
            else:
                xn = self._alloc(rows, R, R, C)
                self._call("vh_pixnorm", L.PixnormArgs(inp=x.ptr, out=xn.ptr, rows=rows, h=R, w=R, c=C, pool=0, norm=1, out_s8=xs_ptr))
            if x3:
                y = self._conv([(xs, 1.0)], self.W[p + "conv_res0.weight"], rows, R, R, epi=L_EPI_SCALE_SILU, cvec=cv, prec=1, s8_only=True)
                self._free(xs)
            else:
                y = self._conv([(xn, 1.0)], self.W[p + "conv_res0.weight"], rows, R, R, pro=L_PRO_SILU, epi=L_EPI_SCALE_SILU, cvec=cv)
            r = self._conv([(y, 1.0)], self.W[p + "conv_res1.weight"], rows, R, R, epi=L_EPI_MPSUM, res=xn if xn is not None else res_src,
                           res_scale=res_scale, ta=ta, tb=tb, clip=clip_res, prec=P1, also_s8=res1_s8 or fin_s8, sink_plan=out_sink)
            self._free(y)
            self._free(xn)
            self._free(res_scale)
        else:
            up = 1 if b.resample == "up" else 0
            xup = None
            if up and not self.std_filter:                         # general FIR filter (:60-61): materialise the upsampled input
                x = xup = self._resample(x, up=True)
                up = 0
            if skip is not None:                                   # mp_cat :78-84
                t = cfg.concat_balance
                Na, Nb = x.shape[-1], skip.shape[-1]
                Cc = math.sqrt((Na + Nb) / ((1 - t) ** 2 + t ** 2))
                srcs = [(x, Cc / math.sqrt(Na) * (1 - t)), (skip, Cc / math.sqrt(Nb) * t)]
            else:
                srcs = [(x, 1.0)]
            craw, tail32, src32 = None, False, False
            st = self._cat.get(cat_j) if (cat_j is not None and skip is not None) else None
            if x3 and st is not None and (st["x_done"] or st["skip_done"]):
                # at least one half of mp_silu(mp_cat(x, skip)) was written by its producer (vh_s8_sink); vh_split fills in the other, if any
                if not st["x_done"]:
                    self._split_half(x, st["sc0"], st, 0)
                if not st["skip_done"]:
                    self._split_half(skip, st["sc1"], st, st["Na"])
                cs, craw = st["cs"], st["craw"]
            elif x3:
                # mp_silu(mp_cat(...)) once per element as S8; the raw split is conv_skip's input - unless the fused launch reads x and skip as fp32
                # (tail_f32), and conv_res0 stages its patches from the fp32 tensors itself (src_f32): then no vh_split pass at all
                tail32 = has_skip_conv and not up and self._tail_f32(b, rows, srcs)
                src32 = self._src_f32(b, rows, srcs, up)
                cs = None
                if has_skip_conv and not tail32:
                    if src32:
                        craw = self._split(srcs, 0)
                    else:
                        cs, craw = self._split(srcs, L_PRO_SILU, raw_too=True)
                elif not src32:
                    cs = self._split(srcs, L_PRO_SILU)
            if x3 and src32:
                y = self._conv(srcs, self.W[p + "conv_res0.weight"], rows, R, R, up=up, pro=L_PRO_SILU,
                               epi=L_EPI_SCALE_SILU, cvec=cv, prec=1, s8_only=True, src_f32=True)
            elif x3:
                y = self._conv([(cs, 1.0)], self.W[p + "conv_res0.weight"], rows, R, R, up=up,
                               epi=L_EPI_SCALE_SILU, cvec=cv, prec=1, s8_only=True)
                self._free(cs)
            else:
                y = self._conv(srcs, self.W[p + "conv_res0.weight"], rows, R, R, up=up, pro=L_PRO_SILU,
                               epi=L_EPI_SCALE_SILU, cvec=cv)
            if has_skip_conv and x3 and self._fused_skip(b):
                # conv_res1 + conv_skip as one GEMM: the raw concat enters as the 1-tap tail of the K loop, ta / tb are in the weights
                r = self._conv([(y, 1.0)] + (list(srcs) if tail32 else [(craw, 1.0)]), self.W[p + "conv_res1+skip"], rows, R, R, epi=L_EPI_STORE, clip=clip_res,
                               prec=1, also_s8=(res1_s8 or fin_s8) and not fin_only, s8_only=fin_only, sink_plan=out_sink, fp32_optional=fp32_optional,
                               tail_f32=tail32)
                self._free(craw)
                xsk = None
            else:
                if has_skip_conv:
                    if x3:
                        xsk = self._conv([(craw, 1.0)], self.W[p + "conv_skip.weight"], rows, R, R, up=up, prec=1)
                        self._free(craw)
                    else:
                        xsk = self._conv(srcs, self.W[p + "conv_skip.weight"], rows, R, R, up=up)
                    res, res_up = xsk, 0
                else:
                    assert skip is None
                    xsk, res, res_up = None, x, up
                r = self._conv([(y, 1.0)], self.W[p + "conv_res1.weight"], rows, R, R, epi=L_EPI_MPSUM, res=res,
                               res_up=res_up, ta=ta, tb=tb, clip=clip_res, prec=P1, also_s8=(res1_s8 or fin_s8) and not fin_only, s8_only=fin_only,
                               sink_plan=out_sink, fp32_optional=fp32_optional)
            self._free(y)
            self._free(xsk)
            self._free(xup)
        if fin_only:
            out, r_s8 = Ghost((rows, R, R, C)), r
        elif res1_s8 or fin_s8:
            out, r_s8 = r
        else:
            out, r_s8 = r, None
        if fin_s8 or fin_only:
            out_s8 = r_s8
        if not isinstance(out, Ghost):
            self._tap(p + "res", out)
        if b.heads:
            S = R * R
            use_feat = b.xattn and feat is not None
            kl = S * (1 + self.nsrc) if use_feat else S
            nz = n_zero * S if (b.xattn and not use_feat) else 0.0
            fused = ax3 and self._qkv_fused(b)
            klp = _round_up(kl, 64) if ax3 else kl            # bf16x3: K as S8, V transposed, keys padded to 64
            split_op, attn_op = ("vh_qkv_split_x3", "vh_attention_x3") if ax3 else ("vh_qkv_split", "vh_attention")
            q = self._alloc(rows, b.heads, S, D)
            if kv_pre is not None:
                # the cross keys / values of this block were written when the features were (the 'features' program of this slot: they depend
                # on the features only, training/models.py:279-297); attn_qkv adds the self keys at offset 0 of the same tensors
                assert use_feat and fused
                k, v = kv_pre
            else:
                k = self._alloc(rows, b.heads, klp, D)
                v = self._alloc(rows, b.heads, klp, D)
            if fused:
                self._conv([(r_s8, 1.0)], self.W[p + "attn_qkv.weight"], rows, R, R, prec=1,
                           qkv=L.QkvEpilogue(q=q.ptr, k=k.ptr, v=v.ptr, heads=b.heads, nj=3, rows_per_b=1, koff=0, kl=kl,
                                             qscale=LOG2E / math.sqrt(D)))
                self._free(r_s8)
            else:
                if ax3:
                    qkv = self._conv([(r_s8, 1.0)], self.W[p + "attn_qkv.weight"], rows, R, R, prec=1)
                    self._free(r_s8)
                else:
                    qkv = self._conv([(out, 1.0)], self.W[p + "attn_qkv.weight"], rows, R, R)
                self._call(split_op, L.QkvSplitArgs(inp=qkv.ptr, rows=rows, s=S, heads=b.heads, d=D, nj=3, rows_per_b=1,
                                                    koff=0, kl=kl, qscale=LOG2E / math.sqrt(D), q=q.ptr, k=k.ptr, v=v.ptr))
                self._free(qkv)
            if use_feat and kv_pre is None:
                if ax3:
                    fs, own = feat_s8, False
                    if fs is None:
                        fs, own = self._split([(feat, 1.0)], 0), True
                    if fused:
                        self._conv([(fs, 1.0)], self.W[p + "x_attn_kv.weight"], rows * self.nsrc, R, R, prec=1,
                                   qkv=L.QkvEpilogue(q=None, k=k.ptr, v=v.ptr, heads=b.heads, nj=2, rows_per_b=self.nsrc,
                                                     koff=S, kl=kl, qscale=1.0))
                        kv = None
                    else:
                        kv = self._conv([(fs, 1.0)], self.W[p + "x_attn_kv.weight"], rows * self.nsrc, R, R, prec=1)
                    if own:
                        self._free(fs)
                else:
                    kv = self._conv([(feat, 1.0)], self.W[p + "x_attn_kv.weight"], rows * self.nsrc, R, R)
                if kv is not None:
                    self._call(split_op, L.QkvSplitArgs(inp=kv.ptr, rows=rows * self.nsrc, s=S, heads=b.heads, d=D, nj=2,
                                                        rows_per_b=self.nsrc, koff=S, kl=kl, qscale=1.0, q=None, k=k.ptr, v=v.ptr))
                    self._free(kv)
            att = self._alloc(rows, R, R, C)
            self._call(attn_op, L.AttentionArgs(q=q.ptr, k=k.ptr, v=v.ptr, b=rows, heads=b.heads, s=S, kl=kl, d=D,
                                                n_zero_keys=nz, out=att.ptr, out_s8=1 if ax3 else 0,
                                                logit_bound=LOG2E * math.sqrt(D) * 1.001),   # q, k are RMS-normalised head vectors
                       f"b={rows} h={b.heads} S={S} KL={kl} D={D} nz={nz} x3={int(ax3)}")
            self._free(q)
            if kv_pre is None:
                self._free(k); self._free(v)
            ta2, tb2 = self._mp_sum_coeffs(cfg.attn_balance)
            emit = want_s8 and ax3
            r2 = self._conv([(att, 1.0)], self.W[p + "attn_proj.weight"], rows, R, R, epi=L_EPI_MPSUM, res=out,
                            ta=ta2, tb=tb2, clip=clip, out=out, prec=1 if ax3 else 0, also_s8=emit)
            if emit:
                out_s8 = r2[1]
            self._free(att)
        if not isinstance(out, Ghost):
            self._tap(p + "out", out)
        return out, out_s8

    # ------------------------------------------------------------------ embeddings
    def _embedding(self, prefix: str, spec: UNetSpec, rows: int, sigma: Buf, sigma_stride: int, time_scale: float,
                   geometry: Optional[Buf], label_dim: int) -> Tuple[Buf, Buf]:
        P = self._params
        emb = self._alloc(rows, spec.cemb)
        wn = self.W[prefix + "emb_noise.weight"]
        wl = self.W.get(prefix + "emb_label.weight")
        has_label = wl is not None and geometry is not None
        self._call("vh_embed", L.EmbedArgs(
            sigma=sigma.ptr, sigma_stride=sigma_stride, time_scale=time_scale,
            geometry=geometry.ptr if has_label else None, label_dim=label_dim if has_label else 0,
            geometry_scale=0.0 if self.cfg.uncond else 1.0,
            freqs=P[prefix + "emb_fourier.freqs"].data_ptr(), phases=P[prefix + "emb_fourier.phases"].data_ptr(),
            cnoise=spec.cnoise, w_noise=wn.wt.data_ptr(), w_noise_kpad=wn.k_pad,
            w_label=wl.wt.data_ptr() if has_label else None, w_label_kpad=wl.k_pad if has_label else 0,
            label_balance=self.cfg.label_balance, rows=rows, cemb=spec.cemb, raw=0, emb=emb.ptr))
        wt, cols, total = self.embW[prefix]
        cvec = self._alloc(rows, total)
        self._call("vh_linear", L.LinearArgs(emb=emb.ptr, rows=rows, cemb=spec.cemb, wt=wt.data_ptr(),
                                            k_pad=_round_up(spec.cemb, 32), cols=total, bias=1.0, out=cvec.ptr))
        self._free(emb)
        return cvec, None

    # ------------------------------------------------------------------ networks
    def _run_unet(self, prefix: str, spec: UNetSpec, rows: int, x_in: Buf, cvec: Buf,
                  feats: Optional[List[Tuple[Buf, Optional[Buf]]]], collect: bool, n_zero: float):
        """Shared walk of UNetEncoder.forward (collect=True) and XAttnUNet.forward (feats given or zero).
        feats / the returned feature list hold (fp32 NHWC, S8 copy or None) pairs."""
        _, cols, total = self.embW[prefix]
        skips: List[Buf] = []
        out_feats: List[Tuple[Buf, Optional[Buf]]] = []
        fi = 0
        x = x_in
        self._cat, consumer = {}, {}
        fuse = self._fuse_mode()
        if self.x3 and self.glds and fuse > 0 and self.hook is None:
            self._cat, consumer = self._cat_plan(spec, rows)

        def skip_sink(ei):
            j = consumer.get(ei)
            return (j, "skip") if fuse >= 2 and j is not None and self._cat[j]["ok"] else None

        def kept(buf):
            return any(buf is f[0] for f in out_feats)

        def next_feat(b):
            nonlocal fi
            f = (None, None, None)
            if b.xattn:
                if feats is not None:
                    f = tuple(feats[fi]) + (None,) * (3 - len(feats[fi]))
                fi += 1
            return f

        for ei, b in enumerate(spec.enc):
            if b.kind == "conv":
                if self.x3:
                    xs8 = self._split([(x, 1.0)], 0)
                    nx = self._conv([(xs8, 1.0)], self.W[f"{prefix}enc.{b.name}.weight"], rows, b.res, b.res, prec=1, sink_plan=skip_sink(ei))
                    self._free(xs8)
                else:
                    nx = self._conv([(x, 1.0)], self.W[f"{prefix}enc.{b.name}.weight"], rows, b.res, b.res)
                self._tap(f"{prefix}enc.{b.name}.out", nx)
                self._free(x)
            else:
                f32, f8, fkv = next_feat(b)
                nx, nx8 = self._block(prefix, "enc", b, rows, x, None, cvec, cols, total, f32, f8, n_zero,
                                      want_s8=collect and b.heads > 0, out_sink=skip_sink(ei), kv_pre=fkv)
                if collect and b.heads > 0:
                    out_feats.append((nx, nx8))
                # x was the output of entry ei-1; if its skip half already sits in its consumer's concat tensors, this block was its last fp32 reader
                pj = consumer.get(ei - 1)
                if pj is not None and self._cat[pj]["skip_done"] and not kept(x) and skips and skips[-1] is x:
                    self._free(x)
                    skips[-1] = Ghost(x.shape)
            skips.append(nx)
            x = nx
        n_live = sum(1 for b in spec.dec if b.live) if (spec.dec and spec.dec[0].live) else 0
        x8 = None
        for j, b in enumerate(spec.dec):
            if not b.live:
                break
            skip = skips.pop() if b.takes_skip else None
            f32, f8, fkv = next_feat(b)
            # this block's result is the x half of the NEXT block's concat input, and nothing else reads it
            nb_ = spec.dec[j + 1] if j + 1 < len(spec.dec) and spec.dec[j + 1].live else None
            xs_ok = nb_ is not None and nb_.takes_skip and (j + 1) in self._cat and self._cat[j + 1]["ok"] and not b.heads
            nx, nx8 = self._block(prefix, "dec", b, rows, x, skip, cvec, cols, total, f32, f8, n_zero,
                                  want_s8=collect and b.heads > 0, cat_j=j if j in self._cat else None,
                                  out_sink=(j + 1, "x") if xs_ok else None, fp32_optional=xs_ok, kv_pre=fkv,
                                  s8_final=not collect and j == n_live - 1 and spec.out_channels > 0)
            x8 = nx8 if isinstance(nx, Ghost) and j == n_live - 1 else None
            for old in (x, skip):
                if old is not None and not kept(old) and all(old is not s_ for s_ in skips):
                    self._free(old)
            if collect and b.heads > 0:
                out_feats.append((nx, nx8))
            x = nx
        for s_ in skips:                              # skips the trimmed encoder-decoder never consumed
            if not kept(s_) and s_ is not x:
                self._free(s_)
        return x, out_feats, x8

    # ------------------------------------------------------------------ program construction
    def program(self, mode: str, B: int, has_cond: bool, want_logvar: bool, fill=None, slot: int = 0) -> Program:
        """mode: 'full' (encoder + unet), 'features' (encoder only), 'inject' (unet with supplied
        features), 'uncond' (unet, zero features in closed form), 'bound' (unet reading the feature buffers of THIS engine's
        'features' program of the same `slot` in place - fp32 and S8 copies, no transfer: the split evaluation of
        NVPrecond.encode_features / forward(inject_features=<handle>)).  `slot` (0 / 1) selects one of two independent
        'features' / 'bound' program pairs, so that the encoder can fill one feature set while a UNet evaluation reads the other."""
        key = (mode, B, has_cond, want_logvar, slot)
        if key in self.programs:
            return self.programs[key]
        ext = None
        if mode == "bound":
            ext = self.program("features", B, has_cond, False, slot=slot).io["features_pairs"]
        peak = 0
        prog = None
        for emit in (False, True):
            self._A = Arena()
            self._emit = emit
            self._backing = None
            if emit:
                self.oplog = []
                # the two 'bound' programs of a batch size differ only in the feature pointers they read and never run at the same time
                # (both are replayed on the caller's stream): they share one workspace
                twin = self.programs.get((mode, B, has_cond, want_logvar, slot ^ 1)) if mode == "bound" else None
                if twin is not None and twin.backing.numel() >= peak:
                    self._backing = twin.backing
                else:
                    self._backing = torch.empty(peak, dtype=torch.float32, device=self.device)
                self._A.base_ptr = self._backing.data_ptr()
                if self.hook is None:
                    self.ctx.plan_begin()
            try:
                io = self._walk(mode, B, has_cond, want_logvar, fill if emit else None, ext)
            except BaseException:
                # an op was refused (unsupported shape, alignment): leave the engine and the context usable and the real error
                # visible - state first, then the abort, whose own failure (e.g. a sticky HIP error) must not replace the cause
                recording = emit and self.hook is None
                self._emit, self._A, self._backing = False, None, None
                if recording:
                    try:
                        self.ctx.plan_abort()
                    except Exception:
                        pass
                raise
            if not emit:
                peak = self._A.peak
            else:
                plan = self.ctx.plan_end() if self.hook is None else None
                if plan is not None and self.use_graph:
                    plan.capture_graph()
                prog = Program(plan, self._backing, io)
                prog.oplog = list(self.oplog)
        self._emit = False
        if self.hook is None:
            self.programs[key] = prog
        return prog

    def _walk(self, mode: str, B: int, has_cond: bool, want_logvar: bool, fill=None, ext_feats=None) -> Dict[str, object]:
        cfg = self.cfg
        R = cfg.img_resolution
        rm = self.nsrc if self.dual else 1          # rows per target sample in src/x/sigma/geometry
        rows_all = B * rm
        io: Dict[str, object] = {}
        src_c = 3 + int(cfg.depth_input or cfg.warp_depth_coor)
        need_enc = mode in ("full", "features")
        need_unet = mode != "features"
        io["sigma"] = self._alloc(rows_all)
        io["geometry"] = self._alloc(rows_all, cfg.source_label_dim)
        if need_enc or cfg.warp_depth_coor:
            io["src"] = self._alloc(rows_all, src_c, R, R)
        if need_unet:
            io["x"] = self._alloc(rows_all, cfg.img_channels, R, R)
            io["D"] = self._alloc(B, cfg.img_channels, R, R)
            if has_cond:
                io["cond"] = self._alloc(B, cfg.img_channels, R, R)
        if want_logvar:
            io["logvar"] = self._alloc(B)
        feats: Optional[List[Tuple[Buf, Optional[Buf]]]] = None
        if mode == "inject":
            fin = [self._alloc(rows_all, r, r, c) for (c, r) in self._feature_shapes()]
            io["features_in"] = fin
            feats = [(f, None) for f in fin]
        if mode == "bound":
            feats = list(ext_feats)           # (fp32, S8) buffers that live in the 'features' program's workspace: read in place, never freed here
        if fill is not None:                  # debug (immediate) mode: inputs must be in place before ops run
            fill(Program(None, self._backing, io))

        # depth-warp Fourier features :643-652
        sgrid = dgrid = None
        if cfg.warp_depth_coor:
            sgrid = self._alloc(rows_all, R, R, 128)
            dgrid = self._alloc(rows_all, R, R, 128)
            mean, std = geometry_stats(R)
            # `if torch.all(src[:, :3] == 0): zero grids` (:647-648) decided on the device: a flag kernel, read by the warp kernel
            flag = self._alloc(1)
            self._call("vh_nonzero_flag", L.NonzeroArgs(inp=io["src"].ptr, rows=rows_all, c_used=3, c_total=src_c, hw=R * R, flag=flag.ptr))
            wa = L.WarpArgs(depth=io["src"].ptr, src_c=src_c, depth_ch=3, geometry=io["geometry"].ptr,
                            freqs=self._params["logvar_fourier.freqs"].data_ptr(),
                            phases=self._params["logvar_fourier.phases"].data_ptr(),
                            rows=rows_all, s=R, grid_feat=sgrid.ptr, warp_feat=dgrid.ptr, nonzero_flag=flag.ptr)
            for i in range(20):
                wa.mean[i] = float(mean[i])
                wa.std[i] = float(std[i])
            self._call("vh_warp_features", wa)
            self._free(flag)

        if need_enc:
            spec = self.enc_spec
            self.scratch = getattr(self, "scratch_enc", None)
            segs = [(io["src"], 0, 3 if cfg.warp_depth_coor else src_c, src_c, 1, 0)]
            if cfg.warp_depth_coor:
                segs.append((sgrid, 1, 128, 128, 1, 0))
            xin = self._assemble(segs, rows_all, R, _round_up(spec.in_channels, 8 if self.x3 else 4), io["sigma"])
            self._free(sgrid)
            sgrid = None
            cvec, _ = self._embedding("encoder.", spec, rows_all, io["sigma"], 1, 0.0 if cfg.no_time_enc else 1.0,
                                      io["geometry"], cfg.source_label_dim)
            last, feats, _ = self._run_unet("encoder.", spec, rows_all, xin, cvec, None, True, 0.0)
            if all(last is not f[0] for f in feats):
                self._free(last)
            self._free(cvec)
            io["features_out"] = [f[0] for f in feats]
            if mode == "features" and self.hoist_kv:
                feats = self._cross_kv(feats, B)
            io["features_pairs"] = list(feats)
        self._free(sgrid)

        if need_unet:
            spec = self.unet_spec
            self.scratch = getattr(self, "scratch_unet", None)
            segs = [(io["x"], 0, cfg.img_channels, cfg.img_channels, rm, 1)]
            if cfg.warp_depth_coor:
                segs.append((dgrid, 1, 128, 128, rm, 0))
            if cfg.super_res:
                assert has_cond, "super_res needs a conditioning image (:656)"
                segs.append((io["cond"], 0, cfg.img_channels, cfg.img_channels, 1, 0))
            xin = self._assemble(segs, B, R, _round_up(spec.in_channels, 8 if self.x3 else 4), io["sigma"])
            self._free(dgrid)
            dgrid = None
            label_dim = cfg.target_label_dim
            cvec, _ = self._embedding("unet.", spec, B, io["sigma"], rm, 1.0, io["geometry"], label_dim)
            n_zero = float(self.nsrc) if (mode == "uncond") else 0.0
            last, _, last8 = self._run_unet("unet.", spec, B, xin, cvec, feats if mode != "uncond" else None, False, n_zero)
            if self.x3:
                ls8 = last8 if last8 is not None else self._split([(last, 1.0)], 0)
                F = self._conv([(ls8, 1.0)], self.W["unet.out_conv.weight"], B, R, R, prec=1)
                self._free(ls8)
            else:
                F = self._conv([(last, 1.0)], self.W["unet.out_conv.weight"], B, R, R)
            self._free(last)
            self._free(cvec)
            self._call("vh_precond_out", L.PrecondOutArgs(x=io["x"].ptr, row_mul=rm, f=F.ptr, fc=F.shape[-1], sigma=io["sigma"].ptr,
                                                         sigma_data=cfg.sigma_data, rows=B, c=cfg.img_channels, h=R, w=R, out=io["D"].ptr))
            self._free(F)
            if want_logvar:
                wl = self.W["logvar_linear.weight"]
                self._call("vh_embed", L.EmbedArgs(
                    sigma=io["sigma"].ptr, sigma_stride=rm, time_scale=1.0, geometry=None, label_dim=0, geometry_scale=0.0,
                    freqs=self._params["logvar_fourier.freqs"].data_ptr(), phases=self._params["logvar_fourier.phases"].data_ptr(),
                    cnoise=cfg.logvar_channels, w_noise=wl.wt.data_ptr(), w_noise_kpad=wl.k_pad, w_label=None, w_label_kpad=0,
                    label_balance=0.0, rows=B, cemb=1, raw=1, emb=io["logvar"].ptr))
        self._free(dgrid)
        return io

    def _cross_kv(self, feats, B: int):
        """The cross-attention keys / values of every XAttnBlock of the UNet (x_attn_kv + normalize, training/models.py:279-293) depend on
        the encoder features only - not on x, not on the UNet's own embedding - and the sampler evaluates the features once per noise level:
        compute them HERE, in the 'features' program, into per-block K / V tensors that stay in its workspace; the 'bound' program's attn_qkv
        adds the self keys at offset 0 and the attention reads them.  (The reference recomputes them in every UNet call, :491-492, :279-297.)
        Returns the feature list as (fp32, S8, (K, V) or None) triples."""
        out = []
        blocks = [("enc", b) for b in self.unet_spec.enc] + [("dec", b) for b in self.unet_spec.dec]
        xb = [(g, b) for g, b in blocks if b.kind == "block" and b.xattn]
        assert len(xb) == len(feats)
        for (g, b), (f32, f8) in zip(xb, feats):
            C_, D = b.cout, b.cout // b.heads
            ok = self.x3 and b.cout % 32 == 0 and b.cin % 32 == 0 and D in (32, 64) and self._qkv_fused(b) and f8 is not None
            if not ok:
                out.append((f32, f8, None))
                continue
            S, R = b.res * b.res, b.res
            kl = S * (1 + self.nsrc)
            klp = _round_up(kl, 64)
            k = self._alloc(B, b.heads, klp, D)
            v = self._alloc(B, b.heads, klp, D)
            self._conv([(f8, 1.0)], self.W[f"unet.{g}.{b.name}.x_attn_kv.weight"], B * self.nsrc, R, R, prec=1,
                       qkv=L.QkvEpilogue(q=None, k=k.ptr, v=v.ptr, heads=b.heads, nj=2, rows_per_b=self.nsrc, koff=S, kl=kl, qscale=1.0))
            out.append((f32, f8, (k, v)))
        return out

    def _feature_shapes(self):
        return [(b.cout, b.res) for b in self.unet_spec.enc + self.unet_spec.dec if b.kind == "block" and b.xattn]

    def _assemble(self, segs, rows, R, c_pad, sigma: Buf) -> Buf:
        out = self._alloc(rows, R, R, c_pad)
        a = L.AssembleArgs(nseg=len(segs), sigma=sigma.ptr, sigma_data=self.cfg.sigma_data, rows=rows, h=R, w=R,
                           c_pad=c_pad, out=out.ptr)
        for i, (buf, kind, c, c_src, row_mul, scale) in enumerate(segs):
            a.seg[i] = L.Segment(ptr=buf.ptr, kind=kind, c=c, c_src=c_src, row_mul=row_mul, scale_cin=scale)
        self._call("vh_assemble", a)
        return out


L_PRO_SILU = 1
L_EPI_STORE = 0
L_EPI_SCALE_SILU = 1
L_EPI_MPSUM = 2
L_EPI_QKV = 3
