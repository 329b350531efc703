#!/usr/bin/env python3
"""Pricing of Winograd F(2x2, 3x3) for the bf16x3 convolutions.

F(2x2,3x3) computes a 2x2 output patch from a 4x4 input patch with 16 multiplies per (cin, cout) instead of 36:
    Y = A^T [ (G g G^T) * (B^T d B) ] A          (Lavin & Gray 2016; * = element-wise, then summed over cin = 16 independent GEMMs)
so the matrix work drops 2.25x.  Two questions decide whether it pays on this machine:

  accuracy  (CPU, this script's first half)  the transforms mix values of different magnitude before the bf16 hi/lo split;
            emulated here exactly as the MFMA path would do it (operands split hi = bf16(x), lo = bf16(x - hi), the three products
            hi*hi + hi*lo + lo*hi accumulated in fp64 as a stand-in for the fp32 accumulator) against R.mp_conv in fp64.
  rate      (GPU, second half, needs the library)  the 16 GEMMs have K = Cin, not 9 Cin: each is a 1x1 convolution over M/4 "pixels",
            timed as 16 vh_conv launches; the input / output transforms are priced as HBM passes at the rate vh_split sustains
            (they read and write 4x the activation: 16 transformed values per 4 pixels).

  python tools/micro/winograd_price.py accuracy
  python tools/micro/winograd_price.py rate ROWS H W CIN COUT
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def split(x):
    hi = x.to(torch.float32).to(torch.bfloat16)
    lo = (x.to(torch.float32) - hi.to(torch.float32)).to(torch.bfloat16)
    return hi.to(torch.float64), lo.to(torch.float64)


def x3_matmul(a, b):
    """sum_k a[..., k] * b[k, ...] with the bf16x3 product rule."""
    ah, al = split(a)
    bh, bl = split(b)
    return ah @ bh + ah @ bl + al @ bh


def accuracy():
    from oracle import vivid_ref as R
    torch.manual_seed(0)
    for name, cin, cout, hw, act in (("gauss 256->256 16^2", 256, 256, 16, "gauss"), ("silu 256->256 16^2", 256, 256, 16, "silu"),
                                     ("silu 512->512 16^2", 512, 512, 16, "silu"), ("heavy-tail 256->256 16^2", 256, 256, 16, "heavy")):
        x = torch.randn(1, cin, hw, hw, dtype=torch.float64)
        if act == "silu":
            x = torch.nn.functional.silu(x) / 0.596
        if act == "heavy":
            x = x * torch.exp(torch.randn_like(x))
        w = torch.randn(cout, cin, 3, 3, dtype=torch.float64)
        wn = R.mp_weight(w.float()).double()                       # normalised weights as the kernels see them
        ref = torch.nn.functional.conv2d(x, wn, padding=1)
        # direct, bf16x3: im2col GEMM
        cols = torch.nn.functional.unfold(x, 3, padding=1)[0].T    # [HW, cin*9]
        direct = x3_matmul(cols, wn.reshape(cout, -1).T).T.reshape(1, cout, hw, hw)
        # Winograd, bf16x3 on the transformed operands (transforms in fp32, as the loader / weight preparation would)
        U = torch.einsum("ij,ocjk,lk->ocil", G, wn, G).float().double()             # [cout, cin, 4, 4]
        xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
        T = hw // 2
        patches = xp.unfold(2, 4, 2).unfold(3, 4, 2)[0]            # [cin, T, T, 4, 4]
        V = torch.einsum("ij,ctujk,lk->ctuil", BT, patches, BT).float().double()    # [cin, T, T, 4, 4]
        Mm = torch.zeros(cout, T, T, 4, 4, dtype=torch.float64)
        for i in range(4):
            for l in range(4):
                a = V[:, :, :, i, l].reshape(cin, -1).T            # [tiles, cin]
                b = U[:, :, i, l].T                                # [cin, cout]
                Mm[:, :, :, i, l] = x3_matmul(a, b).T.reshape(cout, T, T)
        Y = torch.einsum("ij,otujk,lk->otuil", AT, Mm, AT)         # [cout, T, T, 2, 2]
        wino = Y.permute(0, 1, 3, 2, 4).reshape(1, cout, hw, hw)
        rel = lambda a_, b_: float((a_ - b_).norm() / b_.norm())
        print(f"{name:28s} direct bf16x3 {rel(direct, ref):.2e}   winograd bf16x3 {rel(wino, ref):.2e}   (winograd exact-arithmetic check {rel(torch.einsum('ij,otujk,lk->otuil', AT, torch.einsum('ctuil,ocil->otuil', V, U), AT).permute(0, 1, 3, 2, 4).reshape(1, cout, hw, hw), ref):.1e})")


def rate(rows, h, w, cin, cout):
    from vivid_amd import _lib as L
    ctx = L.Context(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(0)
    M = rows * h * w
    Mt = M // 4
    zeros = torch.zeros(16384, device="cuda")

    def conv_args(m_rows, taps, s8, wt, out):
        return L.ConvArgs(src0=s8.data_ptr(), src1=None, c0=cin, c1=0, scale0=1.0, scale1=1.0, rows=1, h=1, w=m_rows, up=0, taps=taps, pro=0,
                          wt=wt.data_ptr(), cin_pad=cin, k_pad=taps * cin, zeros=zeros.data_ptr(), zeros_bytes=65536, scratch=None, scratch_floats=0,
                          cout=cout, out=out.data_ptr(), out_s8=None, out_s8_c=0, prec=1, kernel=1, epi=0)

    def timed(fn, n=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    # direct 3x3
    x = torch.randn(rows, h, w, cin, generator=g).cuda()
    s8 = torch.empty(M * cin, device="cuda")
    ctx.call("vh_split", L.SplitArgs(src0=x.data_ptr(), src1=None, c0=cin, c1=0, scale0=1.0, scale1=1.0, pro=0, npix=M, c_pad=cin, out=s8.data_ptr(), out_raw=None))
    w9 = torch.randn(cout, cin, 3, 3, generator=g).cuda()
    wt9 = torch.zeros(9 * cin * cout, device="cuda")
    ctx.call("vh_prep_weight", L.PrepWeightArgs(w=w9.data_ptr(), cout=cout, cin=cin, taps=9, cin_pad=cin, k_pad=9 * cin, gain_ptr=None, gain_value=1.0,
                                                wt=wt9.data_ptr(), dst_col0=0, dst_cols=cout, split=2))
    out = torch.empty(M, cout, device="cuda")
    a9 = L.ConvArgs(src0=s8.data_ptr(), src1=None, c0=cin, c1=0, scale0=1.0, scale1=1.0, rows=rows, h=h, w=w, up=0, taps=9, pro=0, wt=wt9.data_ptr(),
                    cin_pad=cin, k_pad=9 * cin, zeros=zeros.data_ptr(), zeros_bytes=65536, scratch=None, scratch_floats=0, cout=cout, out=out.data_ptr(),
                    out_s8=None, out_s8_c=0, prec=1, kernel=1, epi=0)
    t_direct = timed(lambda: ctx.call("vh_conv", a9))
    # the 16 GEMMs: [M/4, cin] x [cin, cout] each, own weights per position, transformed input V (S8) [16][M/4][cin]
    V = torch.empty(16 * Mt * cin, device="cuda")
    V[:Mt * cin * 4] = s8[:Mt * cin * 4]
    V[Mt * cin * 4:] = V[:Mt * cin * 4].repeat(3)
    w1 = torch.randn(cout, cin, 1, 1, generator=g).cuda()
    wts = []
    for _ in range(16):
        wt1 = torch.zeros(cin * cout, device="cuda")
        ctx.call("vh_prep_weight", L.PrepWeightArgs(w=w1.data_ptr(), cout=cout, cin=cin, taps=1, cin_pad=cin, k_pad=cin, gain_ptr=None, gain_value=1.0,
                                                    wt=wt1.data_ptr(), dst_col0=0, dst_cols=cout, split=2))
        wts.append(wt1)
    Mo = torch.empty(16, Mt, cout, device="cuda")
    args = [conv_args(Mt, 1, V[p * Mt * cin:(p + 1) * Mt * cin], wts[p], Mo[p]) for p in range(16)]
    t_gemm = timed(lambda: [ctx.call("vh_conv", a) for a in args])
    # transforms as HBM passes: input reads 4 B/elem and writes 16 B/elem of S8; output reads 16 B/elem of fp32 and writes 4 (+4 S8)
    flops = 2.0 * M * cout * cin * 9
    bw = 5.3e12                                  # what vh_split / vh_pixnorm sustain (bench.py kernels table)
    t_in = (M * cin * 4 * 5) / bw * 1e3
    t_out = (M * cout * 4 * 6) / bw * 1e3
    t_w = t_gemm + t_in + t_out
    print(f"rows={rows} {h}x{w} cin={cin} cout={cout}: direct 3x3 {t_direct:.3f} ms ({flops / t_direct / 1e9:.0f} TF/s) | winograd: 16 GEMMs {t_gemm:.3f} ms "
          f"({flops * 4 / 9 / t_gemm / 1e9:.0f} TF/s executed) + input transform {t_in:.3f} + output transform {t_out:.3f} (HBM passes at 5.3 TB/s) = {t_w:.3f} ms "
          f"-> {t_direct / t_w:.2f}x of direct ({flops / t_w / 1e9:.0f} TF/s algorithmic)")


if __name__ == "__main__":
    if sys.argv[1] == "accuracy":
        accuracy()
    else:
        rate(*[int(v) for v in sys.argv[2:7]])
