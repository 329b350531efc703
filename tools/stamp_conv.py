#!/usr/bin/env python3
"""Diagnostic: per-segment shader-cycle stamps of conv_x3_glds K loop.  Build the stamped library first: make -C vivid_amd/csrc stamp
  python tools/stamp_conv.py rows h w cin cout [workgroups-to-average]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vivid_amd import _lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libvivid_hip_stamp.so")
rows, h, w, cin, cout = [int(x) for x in sys.argv[1:6]]
ctx = L.Context(torch.cuda.current_stream().cuda_stream)
g = torch.Generator().manual_seed(0)
x = torch.randn(rows, h, w, cin, generator=g).cuda()
wgt = torch.randn(cout, cin, 3, 3, generator=g).cuda()
zeros = torch.zeros(16384, device="cuda")
scr = torch.zeros(16 << 20, device="cuda")
k_pad = 9 * cin
M = rows * h * w
s8 = torch.empty(M * cin, device="cuda")
ctx.call("vh_split", L.SplitArgs(src0=x.data_ptr(), src1=None, c0=cin, c1=0, scale0=1.0, scale1=1.0, pro=0, npix=M, c_pad=cin, out=s8.data_ptr(), out_raw=None))
wt = torch.zeros(k_pad // 4 * cout * 4, device="cuda")
ctx.call("vh_prep_weight", L.PrepWeightArgs(w=wgt.data_ptr(), cout=cout, cin=cin, taps=9, cin_pad=cin, k_pad=k_pad, gain_ptr=None, gain_value=1.0, wt=wt.data_ptr(), dst_col0=0, dst_cols=cout, split=2))
out = torch.empty(M, cout, device="cuda")
a = L.ConvArgs(src0=s8.data_ptr(), src1=None, c0=cin, c1=0, scale0=1.0, scale1=1.0, rows=rows, h=h, w=w, up=0, taps=9, pro=0, wt=wt.data_ptr(), cin_pad=cin, k_pad=k_pad,
               zeros=zeros.data_ptr(), zeros_bytes=65536, scratch=scr.data_ptr(), scratch_floats=scr.numel(), cout=cout, out=out.data_ptr(), out_s8=None, out_s8_c=0,
               prec=1, kernel=1, epi=0, cvec=None, cvec_ld=0, res=None, res_up=0, ta=0, tb=0, clip=0)
for _ in range(20):
    ctx.call("vh_conv", a)
torch.cuda.synchronize()
d = scr.view(torch.int64).cpu().numpy()
nb = int(sys.argv[6]) if len(sys.argv) > 6 else 256
t = d[: nb * 8 * 6].reshape(nb * 8, 6).astype(np.float64)
kt = t[:, 4].mean()
print(f"K-tiles per tile {kt:.0f}; per K-tile, cycles (mean over {nb * 8} waves): issue {t[:,0].mean()/kt:.0f}  compute {t[:,1].mean()/kt:.0f}  wait_dma {t[:,2].mean()/kt:.0f}  barrier {t[:,3].mean()/kt:.0f}  | epilogue per tile {t[:,5].mean():.0f}")
print("  by wave (issue/compute/wait/barrier):", [tuple(int(v) for v in t[i::8, :4].mean(axis=0) / kt) for i in range(8)])
