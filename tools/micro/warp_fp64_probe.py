#!/usr/bin/env python3
"""vh_warp_features at C5's geometry (256x256, depth U(1,5), small rotation + translation) against an fp64 evaluation of the oracle's
get_warped_features (training/utils.py:189-216): ulp error of the warped coordinates, and what it and fp32 evaluation of cos(f u + phase)
do to the 2 x 128 feature channels.  (The numbers tests/test_hip_timed_configs.py::test_warp_kernel_against_fp64 asserts.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from vivid_amd import _lib as L
from vivid_amd.geometry import compose_geometry, geometry_stats
from oracle import vivid_ref as R

rows, S = 4, 256
g = torch.Generator().manual_seed(6)
depth = torch.rand(rows, 1, S, S, generator=g) * 4 + 1
th = 0.05 * torch.randn(rows, generator=g)
Rm = torch.zeros(rows, 3, 3)
Rm[:, 0, 0], Rm[:, 0, 2], Rm[:, 1, 1], Rm[:, 2, 0], Rm[:, 2, 2] = th.cos(), th.sin(), 1.0, -th.sin(), th.cos()
K = (torch.tensor([57.7, 57.7, 32.0, 32.0]) * 4).expand(rows, 4)
geo = compose_geometry(torch.cat([Rm, 0.1 * torch.randn(rows, 3, 1, generator=g)], dim=2), K, K, imsize=S)
freqs = 2 * np.pi * torch.randn(128, generator=g)
phases = 2 * np.pi * torch.rand(128, generator=g)
ctx = L.Context(torch.cuda.current_stream().cuda_stream)
src = torch.cat([torch.rand(rows, 3, S, S, generator=g), depth], 1).cuda()
gf, wf = torch.empty(rows, S, S, 128, device="cuda"), torch.empty(rows, S, S, 128, device="cuda")
uv = torch.empty(rows, S, S, 2, device="cuda")
mean, std = geometry_stats(S)
gd, fd, pd = geo.cuda(), freqs.cuda(), phases.cuda()
wa = L.WarpArgs(depth=src.data_ptr(), src_c=4, depth_ch=3, geometry=gd.data_ptr(), freqs=fd.data_ptr(), phases=pd.data_ptr(), rows=rows, s=S,
                grid_feat=gf.data_ptr(), warp_feat=wf.data_ptr(), nonzero_flag=None, uv_out=uv.data_ptr())
for i in range(20):
    wa.mean[i], wa.std[i] = float(mean[i]), float(std[i])
ctx.call("vh_warp_features", wa)
torch.cuda.synchronize()
ar = torch.arange(0, S, dtype=torch.float64)
ii, jj = torch.meshgrid(ar, ar, indexing="ij")
grid = torch.stack([ii, jj], -1)[None].repeat(rows, 1, 1, 1) + 0.5
uv64 = R.warp_grid(depth.double().permute(0, 2, 3, 1), geo.double(), grid)
uv32 = uv.cpu().double()
ulp = 2.0 ** (torch.floor(torch.log2(uv64.abs().clamp_min(1e-30))) - 23)
err_ulp = ((uv32 - uv64).abs() / ulp)
print(f"warped coordinates: |u| up to {uv64.abs().max():.1f}; error in ulp(fp32): max {err_ulp.max():.2f}, p99.9 {err_ulp.flatten().quantile(0.999):.2f}, mean {err_ulp.mean():.3f}")
ae = (uv32 - uv64).abs().flatten()
print(f"  absolute error: max {ae.max():.3e}, p99.9 {ae.quantile(0.999):.3e}, p99 {ae.quantile(0.99):.3e}, median {ae.median():.3e}  (ulp of 256 = {2.0**-15:.3e})")
# oracle's own fp32 evaluation
uvo = R.warp_grid(depth.permute(0, 2, 3, 1), geo, grid.float()).double()
eo = ((uvo - uv64).abs() / ulp)
aeo = (uvo - uv64).abs().flatten()
print(f"  oracle fp32 absolute error: max {aeo.max():.3e}, p99.9 {aeo.quantile(0.999):.3e}, p99 {aeo.quantile(0.99):.3e}, median {aeo.median():.3e}")
print(f"oracle (fp32 torch.inverse / matmul) : max {eo.max():.2f}, p99.9 {eo.flatten().quantile(0.999):.2f}, mean {eo.mean():.3f}")
f64, p64 = freqs[:64].double(), phases[:64].double()
def emb(c):       # [rows,S,S,2] -> [rows,S,S,128], channel = 64*axis + k
    return torch.cat([torch.cos(c[..., 0:1] * f64 + p64), torch.cos(c[..., 1:2] * f64 + p64)], -1) * np.sqrt(2)
exact = emb(uv64)
rl = lambda a, b: float((a - b).norm() / b.norm())
print(f"features, kernel vs fp64                       : rel-L2 {rl(wf.cpu().double(), exact):.3e}")
print(f"features, fp64 embedding of the KERNEL's (u, v): rel-L2 {rl(emb(uv32), exact):.3e}   (coordinate error alone)")
c32 = uv64.float()
e32 = torch.cat([torch.cos(c32[..., 0:1] * freqs[:64] + phases[:64]), torch.cos(c32[..., 1:2] * freqs[:64] + phases[:64])], -1) * np.float32(np.sqrt(2))
print(f"features, fp32 embedding of the EXACT (u, v)   : rel-L2 {rl(e32.double(), exact):.3e}   (fp32 evaluation of cos(f u + phase) alone)")
print(f"features, fp64 embedding of (u, v) + 1 ulp     : rel-L2 {rl(emb(uv64 + ulp), exact):.3e}")
print(f"grid features (pixel centres), kernel vs fp64  : rel-L2 {rl(gf.cpu().double(), emb(grid)):.3e}")
