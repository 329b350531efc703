#!/usr/bin/env python3
"""Interleaved A/B timing of library builds / knob settings on one device, in one process (cdna guide 5.4 rule 24).

  python tools/ab.py attn  B HEADS S KL [D]            -- vh_attention_x3 as the engine calls it (bounded logits)
  python tools/ab.py pixnorm ROWS H W C                -- vh_pixnorm (scale + S8 of silu(normalised x)); "TF/s" column = GB/s
  python tools/ab.py split ROWS H W C0 C1 RAW          -- vh_split (concat + silu -> S8; RAW 1: also the raw S8 form); "TF/s" column = GB/s
  python tools/ab.py conv  ROWS H W CIN COUT [TAPS] [EPI] [C1]  -- vh_conv, glds kernel (EPI 0 store, 1 cvec + silu, 2 residual mp_sum, 3 the q/k/v
                                                           epilogue of attn_qkv: COUT = 3 * heads * 64, self keys only; C1 > 0: a second S8 source
                                                           of C1 channels as the 1-tap tail segment - conv_res1 + conv_skip as one GEMM)
Variants come from VARIANTS="name=lib[:knob=val[,knob=val]];..." where lib is a suffix of vivid_amd/libvivid_hip[_<suffix>].so
("" = the product build), e.g.  VARIANTS="base=;dyn=attn_dyn;noxcd=:attn_xcd=0".
Prints median and min ms per launch over ROUNDS (default 7) interleaved rounds of N (default 10) launches."""
import ctypes as ct
import math
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from vivid_amd import _lib as L  # noqa: E402


KNOB_DEFAULTS = {"attn_xcd": 1, "attn_m16": 1, "conv_korder": -1, "conv_stagger": -1, "attn_pipe": 1, "attn_nomax": 1, "conv_slim2": -1, "conv_korder_mb": 60, "conv_ksplit": 0, "conv_patch": -1, "conv_patch_delay": 0, "fuse_concat": 0, "conv_patch96": 1, "conv_patch_tail": 2, "conv_tail_f32": 1, "conv_src_f32": 2}


def load(suffix):
    path = os.path.join(ROOT, "vivid_amd", f"libvivid_hip{'_' + suffix if suffix else ''}.so")
    L._lib, L.LIB_PATH = None, path
    return L.Context(torch.cuda.current_stream().cuda_stream)


def parse_variants():
    out = []
    for item in os.environ.get("VARIANTS", "base=").split(";"):
        name, rest = item.split("=", 1)
        lib, _, knobs = rest.partition(":")
        kv = {}
        for k in filter(None, knobs.split(",")):
            a, b = k.split("=")
            kv[a] = int(b)
        out.append((name, lib, kv))
    return out


def main():
    kind = sys.argv[1]
    args = [int(x) for x in sys.argv[2:]]
    rounds, n = int(os.environ.get("ROUNDS", "7")), int(os.environ.get("N", "10"))
    g = torch.Generator().manual_seed(0)
    runs = []
    if kind == "attn":
        b, heads, S, KL = args[:4]
        D = args[4] if len(args) > 4 else 64
        C = heads * D
        qkv = torch.randn(b, S, 3 * C, generator=g).cuda()
        kv = torch.randn(b, max(KL - S, 1), 2 * C, generator=g).cuda()
        klp = (KL + 63) // 64 * 64
        flops = 4.0 * b * heads * S * KL * D
        for name, lib, knobs in parse_variants():
            ctx = load(lib)
            Q = torch.zeros(b * heads * S * D, device="cuda"); K = torch.zeros(b * heads * klp * D, device="cuda"); V = torch.zeros(b * heads * klp * D, device="cuda")
            out = torch.empty(b, S, C, device="cuda")
            ctx.call("vh_qkv_split_x3", L.QkvSplitArgs(inp=qkv.data_ptr(), rows=b, s=S, heads=heads, d=D, nj=3, rows_per_b=1, koff=0, kl=KL,
                                                        qscale=1.4426950408889634 / math.sqrt(D), q=Q.data_ptr(), k=K.data_ptr(), v=V.data_ptr()))
            if KL > S:
                ctx.call("vh_qkv_split_x3", L.QkvSplitArgs(inp=kv.data_ptr(), rows=b, s=KL - S, heads=heads, d=D, nj=2, rows_per_b=1, koff=S, kl=KL,
                                                            qscale=1.0, q=None, k=K.data_ptr(), v=V.data_ptr()))
            a = L.AttentionArgs(q=Q.data_ptr(), k=K.data_ptr(), v=V.data_ptr(), b=b, heads=heads, s=S, kl=KL, d=D, n_zero_keys=0.0, out=out.data_ptr(),
                                out_s8=0, logit_bound=1.4426950408889634 * math.sqrt(D) * 1.001)
            runs.append((name, ctx, "vh_attention_x3", a, knobs, (Q, K, V, out)))
    elif kind == "split":                                  # ROWS H W C0 C1 RAW: the decoder's concat + silu -> S8 (+ raw S8 for the fused skip tail)
        rows, h, w, c0, c1, raw = args[:6]
        M, cp = rows * h * w, (c0 + c1 + 31) // 32 * 32
        x0 = torch.randn(M, c0, generator=g).cuda()
        x1 = torch.randn(M, c1, generator=g).cuda() if c1 else None
        flops = 4.0 * M * (c0 + c1 + cp * (2 if raw else 1)) * 1e3   # printed as "TF/s": read it as GB/s
        for name, lib, knobs in parse_variants():
            ctx = load(lib)
            o = torch.empty(M * cp, device="cuda")
            r = torch.empty(M * cp, device="cuda") if raw else None
            a = L.SplitArgs(src0=x0.data_ptr(), src1=x1.data_ptr() if c1 else None, c0=c0, c1=c1, scale0=0.8, scale1=1.1, pro=1, npix=M, c_pad=cp,
                            out=o.data_ptr(), out_raw=r.data_ptr() if raw else None)
            runs.append((name, ctx, "vh_split", a, knobs, (x0, x1, r, o)))
    elif kind == "pixnorm":                                # ROWS H W C: x -> per-pixel scale + S8 of silu(normalised x), as the encoder blocks call it
        rows, h, w, c = args[:4]
        M = rows * h * w
        x = torch.randn(M, c, generator=g).cuda()
        flops = 8.0 * M * c * 1e3                           # printed as "TF/s": read it as GB/s
        for name, lib, knobs in parse_variants():
            ctx = load(lib)
            o = torch.empty(M * c, device="cuda")
            sc = torch.empty(M, device="cuda")
            a = L.PixnormArgs(inp=x.data_ptr(), out=None, rows=rows, h=h, w=w, c=c, pool=0, norm=1, out_s8=o.data_ptr(), scale_out=sc.data_ptr())
            runs.append((name, ctx, "vh_pixnorm", a, knobs, (x, sc, o)))
    else:
        rows, h, w, cin, cout = args[:5]
        taps = args[5] if len(args) > 5 else 9
        epi = args[6] if len(args) > 6 else 0
        c1 = args[7] if len(args) > 7 else 0
        x = torch.randn(rows, h, w, cin, generator=g).cuda()
        wgt = torch.randn(cout, cin, *([3, 3] if taps == 9 else [1, 1]), generator=g).cuda()
        zeros = torch.zeros(16384, device="cuda")
        scr = torch.empty(16 << 20, device="cuda")
        k_pad, M = taps * cin + c1, rows * h * w
        flops = 2.0 * M * cout * (cin * taps + c1)
        x1 = torch.randn(rows, h, w, c1, generator=g).cuda() if c1 else None
        wgt1 = torch.randn(cout, c1, 1, 1, generator=g).cuda() if c1 else None
        res = torch.randn(M, cout, generator=g).cuda() if epi == 2 else None     # (knob resup=1: read as the half-resolution residual of an `up` block)
        cvec = (torch.randn(rows, cout, generator=g) * 0.3 + 1).cuda() if epi == 1 else None
        heads, S = cout // 192, h * w
        for name, lib, knobs in parse_variants():
            ctx = load(lib)
            s8 = torch.empty(M * cin, device="cuda")
            ctx.call("vh_split", L.SplitArgs(src0=x.data_ptr(), src1=None, c0=cin, c1=0, scale0=1.0, scale1=1.0, pro=0, npix=M, c_pad=cin, out=s8.data_ptr(), out_raw=None))
            wt = torch.zeros(k_pad // 4 * cout * 4, device="cuda")
            ctx.call("vh_prep_weight", L.PrepWeightArgs(w=wgt.data_ptr(), cout=cout, cin=cin, taps=taps, cin_pad=cin, k_pad=taps * cin, gain_ptr=None,
                                                        gain_value=1.0, wt=wt.data_ptr(), dst_col0=0, dst_cols=cout, split=2, k_off=0, k_stride=k_pad if c1 else 0))
            s81 = None
            if c1:
                s81 = torch.empty(M * c1, device="cuda")
                ctx.call("vh_split", L.SplitArgs(src0=x1.data_ptr(), src1=None, c0=c1, c1=0, scale0=1.0, scale1=1.0, pro=0, npix=M, c_pad=c1, out=s81.data_ptr(), out_raw=None))
                ctx.call("vh_prep_weight", L.PrepWeightArgs(w=wgt1.data_ptr(), cout=cout, cin=c1, taps=1, cin_pad=c1, k_pad=c1, gain_ptr=None,
                                                            gain_value=1.0, wt=wt.data_ptr(), dst_col0=0, dst_cols=cout, split=2, k_off=taps * cin, k_stride=k_pad))
            out = torch.empty(M, cout, device="cuda")
            qkv = None
            if epi == 3:
                Q, K, V = (torch.zeros(rows * heads * S * 64, device="cuda") for _ in range(3))
                qkv = L.QkvEpilogue(q=Q.data_ptr(), k=K.data_ptr(), v=V.data_ptr(), heads=heads, nj=3, rows_per_b=1, koff=0, kl=S, qscale=0.18)
                out = torch.cat([Q, K, V])                  # (a copy: the comparison below is then trivially 0; timing only)
            tailf = knobs.pop("tailf32", 0) if c1 else 0    # 1: the tail segment read from the fp32 tensor (vh_conv_args.tail_f32)
            srcf = knobs.pop("srcf32", 0)                   # 1: the main loop reads the fp32 tensor through mp_silu (vh_conv_args.src_f32): no S8 input
            s8mode = knobs.pop("s8", 0)                     # 0: fp32 output, 1: S8 only, 2: both
            o8 = torch.empty(M * cout, device="cuda") if s8mode else None
            a = L.ConvArgs(src0=x.data_ptr() if srcf else s8.data_ptr(), src_f32=srcf, src1=(x1.data_ptr() if tailf else s81.data_ptr()) if c1 else None, tail_f32=tailf, c0=cin, c1=c1, scale0=1.0, scale1=1.0, rows=rows, h=h, w=w, up=knobs.pop("up", 0), taps=taps, pro=1 if srcf == 1 else 0,
                           wt=wt.data_ptr(), cin_pad=cin, k_pad=k_pad, zeros=zeros.data_ptr(), zeros_bytes=65536, scratch=scr.data_ptr(),
                           scratch_floats=scr.numel(), cout=cout, out=out.data_ptr() if (s8mode != 1 and epi != 3) else None, out_s8=o8.data_ptr() if o8 is not None else None,
                           out_s8_c=cout if o8 is not None else 0, prec=1, kernel=1, epi=epi,
                           cvec=cvec.data_ptr() if cvec is not None else None, cvec_ld=cout if cvec is not None else 0, res=res.data_ptr() if res is not None else None, res_up=knobs.pop("resup", 0), ta=0.7, tb=0.3, clip=256.0 if epi == 2 else 0.0,
                           korder=knobs.pop("korder", 0), tile=knobs.pop("tile", 0), stagger=knobs.pop("stagger", 0),
                           qkv=ct.addressof(qkv) if qkv is not None else None)
            runs.append((name, ctx, "vh_conv", a, knobs, (s8, s81, wt, qkv, (Q, K, V) if epi == 3 else None, o8, out, o8 if s8mode == 1 else out)))   # (every buffer the launch writes stays referenced)

    def launch(r, k):
        name, ctx, op, a, knobs, _ = r
        for kk, vv in knobs.items():
            ctx._L.vh_set_knob(kk.encode(), vv)
        for _ in range(k):
            ctx.call(op, a)
        for kk in knobs:                       # back to the defaults of the next variant's library
            ctx._L.vh_set_knob(kk.encode(), KNOB_DEFAULTS.get(kk, 0))

    for r in runs:
        launch(r, 2)
    torch.cuda.synchronize()
    ref = runs[0][5][-1].clone()
    times = {r[0]: [] for r in runs}
    for _ in range(rounds):
        for r in runs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch(r, n)
            e1.record()
            torch.cuda.synchronize()
            times[r[0]].append(e0.elapsed_time(e1) / n)
    base = statistics.median(times[runs[0][0]])
    for r in runs:
        t = times[r[0]]
        med = statistics.median(t)
        err = float((r[5][-1] - ref).norm() / ref.norm())
        print(f"{r[0]:12s} median {med:8.4f} ms  min {min(t):8.4f} ms  {flops / med / 1e9:7.1f} TF/s  vs first {base / med:6.3f}x  diff-vs-first {err:.1e}")


if __name__ == "__main__":
    main()
