"""GPU: the configurations bench.py TIMES, at the batch sizes it times them.

The dispatcher picks kernels by problem size (csrc/conv_x3.hip: chunk-major K order above 150 MB inputs, 512x128 / 256x256 tiles
above 512 / 256 workgroups, split-K below 256 tiles; attention: bounded-logit pipelined kernel for long sequences), so a batch-1
evaluation — cheap enough for the CPU oracle — runs DIFFERENT instantiations from the batch the benchmark is quoted on.  Each
BASELINE.json configuration is therefore pinned in two links:

  1. batch 1  vs  the CPU oracle (oracle/vivid_ref.py, itself pinned to the reference by tests/golden/) — directly, both
     arithmetic modes (bf16x3 = the benchmark's; fp32 = the exact-fp32 MFMA kernels);
  2. the timed batch (C2: 16, C4: 4, C5: 16; distinct samples, distinct noise levels) vs the SAME network's batch-1 evaluations of
     samples first / middle / last, <= 2e-5 — sample 0 of the batch IS the oracle-pinned input, and is also compared with the
     oracle itself (<= 1e-4).

Reference path: NVPrecond._forward_dualsource training/models.py:628-689; the guided closure generate_images.py:55-62.
Host cost on the GPU box: three oracle evaluations (C2 net + guidance net ~70 s, C4 ~70 s, C5 ~45 s).
"""
import math

import pytest
import torch

from oracle import vivid_ref as R
from tests.conftest import rel_l2

pytestmark = pytest.mark.gpu

SIGMAS = [5.0, 80.0, 21.0, 9.7, 3.1, 1.2, 0.47, 0.3, 0.11, 0.05, 0.02, 0.002, 40.0, 2.0, 0.8, 0.15]


def _net(cfg, seed, precision):
    import vivid_amd
    net = vivid_amd.NVPrecond.from_config(cfg, precision=precision)
    sd = vivid_amd.synth_state_dict(cfg, seed=seed)
    net.load_state_dict(sd, strict=True)
    return net.cuda(), sd


def _ocfg(cfg):
    d = cfg.to_dict()
    d.pop("use_fp16")
    return R.make_config(**d)


def _batch(R_, B, seed, src_c=3):
    """B distinct samples (dual-source rows interleaved), one noise level per sample."""
    g = torch.Generator().manual_seed(seed)
    src = torch.rand(2 * B, src_c, R_, R_, generator=g) * 2 - 1
    img = torch.rand(B, 3, R_, R_, generator=g) * 2 - 1
    eps = torch.randn(B, 3, R_, R_, generator=g)
    geo = torch.randn(2 * B, 20, generator=g)
    geo[:, [14, 15, 18, 19]] = 0
    sig = torch.tensor(SIGMAS[:B])
    x = (img + sig.view(-1, 1, 1, 1) * eps).repeat_interleave(2, dim=0)
    return src, x, sig.repeat_interleave(2), geo


def _rows(t, i):
    return None if t is None else t[2 * i:2 * i + 2]


def _cuda(*ts):
    return [None if t is None else t.cuda() for t in ts]


def _check_batch_vs_singles(net, src, x, sig, geo, cond, picks, tol=2e-5):
    """One evaluation at the full batch, then `picks` alone at batch 1: same network object, so the only difference is which
    kernels / tile shapes / K orders / split-K partitions the launcher picked for the two grid sizes."""
    full = net(*_cuda(src, x, sig, geo, cond)).cpu()
    assert torch.isfinite(full).all()
    singles = {}
    for i in picks:
        c1 = None if cond is None else cond[i:i + 1]
        one = net(*_cuda(_rows(src, i), _rows(x, i), _rows(sig, i), _rows(geo, i), c1)).cpu()
        singles[i] = one
        err = rel_l2(full[i], one[0])
        assert err < tol, f"sample {i} of the batch differs from its batch-1 evaluation: {err:.2e}"
    return full, singles


# ------------------------------------------------------------------------------------------------ C2 (the headline)
@pytest.fixture(scope="module")
def c2():
    import vivid_amd
    cfg, ucfg = vivid_amd.vivid_base(256), vivid_amd.vivid_uncond(256)
    net, sd = _net(cfg, 0, "bf16x3")
    gnet, usd = _net(ucfg, 1, "bf16x3")
    src, x, sig, geo = _batch(256, 16, 21)
    with torch.no_grad():
        rD = R.nvprecond_forward(sd, _ocfg(cfg), src[:2], x[:2], sig[:2], geo[:2])
        rG = R.nvprecond_forward(usd, _ocfg(ucfg), src[:2], x[:2], sig[:2], None)
    yield dict(net=net, gnet=gnet, inp=(src, x, sig, geo), rD=rD, rG=rG)
    del net, gnet
    torch.cuda.empty_cache()


def test_headline_config_vs_oracle_batch1(c2):
    """BASELINE configs[1] at batch 1 against the CPU oracle directly: the 16384 x 49152 cross-attention, the 256x256-level
    convolutions, the closed-form zero keys of the guidance net, and the guided combination ref.lerp(D, 1.5)."""
    src, x, sig, geo = c2["inp"]
    D = c2["net"](*_cuda(src[:2], x[:2], sig[:2], geo[:2])).cpu()
    Dg = c2["gnet"](*_cuda(src[:2], x[:2], sig[:2])).cpu()
    assert D.shape == c2["rD"].shape == (1, 3, 256, 256)
    assert rel_l2(D, c2["rD"]) < 1e-4
    assert rel_l2(Dg, c2["rG"]) < 1e-4
    assert rel_l2(Dg.lerp(D, 1.5), c2["rG"].lerp(c2["rD"], 1.5)) < 1e-4


def test_headline_config_at_the_timed_batch(c2):
    """Batch 16 — what bench.py times: the chunk-major 512x128 / 256x256 convolution kernels at 0.5-1 GB inputs (FastDiv magic
    numbers, 64-bit offsets, M up to 2.1 M pixels) and the attention grid of 16 x heads x 64 query tiles."""
    src, x, sig, geo = c2["inp"]
    D16, _ = _check_batch_vs_singles(c2["net"], src, x, sig, geo, None, (0, 7, 15))
    assert D16.shape == (16, 3, 256, 256)
    assert rel_l2(D16[:1], c2["rD"]) < 1e-4                       # sample 0 is the oracle's input
    G16, _ = _check_batch_vs_singles(c2["gnet"], src, x, sig, None, None, (0, 7, 15))
    assert rel_l2(G16[:1], c2["rG"]) < 1e-4
    assert rel_l2(G16[:1].lerp(D16[:1], 1.5), c2["rG"].lerp(c2["rD"], 1.5)) < 1e-4


def test_headline_fp32_kernels_vs_oracle_batch1(c2):
    """The exact-fp32 kernel family (bench.py's `other_workloads.c2_fp32`) against the same oracle outputs."""
    import vivid_amd
    src, x, sig, geo = c2["inp"]
    net, _ = _net(vivid_amd.vivid_base(256), 0, "fp32")
    D = net(*_cuda(src[:2], x[:2], sig[:2], geo[:2])).cpu()
    del net
    torch.cuda.empty_cache()
    assert rel_l2(D, c2["rD"]) < 2e-5


# ------------------------------------------------------------------------------------------------ C4 (SR net built at 1024^2)
def test_config4_sr_1024_vs_oracle_and_timed_batch():
    """BASELINE configs[3]: the SR class built with img_resolution=1024.  Batch 1 against the oracle in both arithmetic
    modes; batch 4 (the timed batch: 512x64 'slim' tiles at M = 4.2 M pixels, 32-channel heads on the fused q/k/v epilogue)
    against batch 1."""
    import vivid_amd
    cfg = vivid_amd.vivid_sr(1024, noisy_sr=0.0)
    src, x, sig, geo = _batch(1024, 4, 11)
    g = torch.Generator().manual_seed(4)
    cond = torch.nn.functional.interpolate(torch.rand(4, 3, 256, 256, generator=g) * 2 - 1, size=(1024, 1024), mode="bilinear")
    net, sd = _net(cfg, 0, "bf16x3")
    with torch.no_grad():
        ref = R.nvprecond_forward(sd, _ocfg(cfg), src[:2], x[:2], sig[:2], geo[:2], cond[:1])
    assert ref.shape == (1, 3, 1024, 1024)
    full, singles = _check_batch_vs_singles(net, src, x, sig, geo, cond, (0, 3))
    assert rel_l2(singles[0], ref) < 1e-4
    assert rel_l2(full[:1], ref) < 1e-4
    del net
    torch.cuda.empty_cache()
    net32, _ = _net(cfg, 0, "fp32")
    D32 = net32(*_cuda(src[:2], x[:2], sig[:2], geo[:2], cond[:1])).cpu()
    assert rel_l2(D32, ref) < 2e-5


# ------------------------------------------------------------------------------------------------ C5 (base + depth-warp features)
def test_config5_depth_warp_256_vs_oracle_and_timed_batch():
    """BASELINE configs[4]: base architecture + depth-warp Fourier features at 256^2 (132-channel first convolutions,
    get_warped_features training/utils.py:204-216).  Batch 1 against the oracle; batch 16 against batch 1."""
    import vivid_amd
    from vivid_amd.geometry import compose_geometry
    B = 16
    cfg = vivid_amd.vivid_base(256, warp_depth_coor=True)
    src, x, sig, _ = _batch(256, B, 13, src_c=4)
    src[:, 3] = src[:, 3] * 2 + 3                                   # depth in [1, 5]
    g = torch.Generator().manual_seed(6)
    th = 0.05 * torch.randn(2 * B, generator=g)
    Rm = torch.zeros(2 * B, 3, 3)
    Rm[:, 0, 0], Rm[:, 0, 2], Rm[:, 1, 1], Rm[:, 2, 0], Rm[:, 2, 2] = th.cos(), th.sin(), 1.0, -th.sin(), th.cos()
    K = (torch.tensor([57.7, 57.7, 32.0, 32.0]) * 4).expand(2 * B, 4)
    geo = compose_geometry(torch.cat([Rm, 0.1 * torch.randn(2 * B, 3, 1, generator=g)], dim=2), K, K, imsize=256)
    net, sd = _net(cfg, 0, "bf16x3")
    with torch.no_grad():
        ref = R.nvprecond_forward(sd, _ocfg(cfg), src[:2], x[:2], sig[:2], geo[:2])
    full, singles = _check_batch_vs_singles(net, src, x, sig, geo, None, (0, 7, 15))
    assert full.shape == (B, 3, 256, 256)
    # Tolerance vs the oracle: 2e-4 here, not 1e-4.  DERIVED, not argued: test_warp_kernel_against_fp64 (below) holds the kernel's warped
    # coordinates to the fp64 value within 3 ulp of the image scale - no worse than the oracle's own fp32 evaluation - and measures what that
    # does to this configuration's network INPUT, cos(f u + phase) of a coordinate u <= 256 with f ~ 2 pi N(0,1) (MPFourier on
    # get_warped_features, training/utils.py:204-216): either fp32 evaluation sits 4-5e-5 (rel-L2) from the fp64 features, so two correct fp32
    # evaluations differ by up to ~9e-5 in the input, and D_x by about as much (measured 8e-5).  2e-4 is 2x that, 5x inside north_star's 1e-3.
    # (The exact-fp32 mode of this configuration used to be run here as a cross-check, 1e-5 from bf16x3; the fp64 test replaces it.)
    assert rel_l2(singles[0], ref) < 2e-4
    assert rel_l2(full[:1], ref) < 2e-4


def test_warp_kernel_against_fp64():
    """vh_warp_features at C5's geometry (256^2, depth U(1,5), small rotation + translation) against an fp64 evaluation of the oracle's
    get_warped_features / warp_image (training/utils.py:189-216): the warped coordinates themselves (vh_warp_args.uv_out) and the 2 x 64
    Fourier channels per grid.  Pins the KERNEL, where the C5 test above can only see the network's sensitivity to its input:
      * (u, v) within 3 ulp(fp32) of the image scale (ulp(256) = 3.05e-5; measured max 2.1, p99.9 1.1) and no worse than the oracle's own
        fp32 evaluation at any quantile (torch.inverse + matmul: max 2.4 ulp);
      * feature tensor within 6e-5 rel-L2 of the fp64 features (measured 4.5e-5), made of the coordinate error (4.0e-5) and of evaluating
        cos(f u + phase) in fp32 at |f u| up to ~5000 rad (3.7e-5 even from EXACT coordinates; a 1-ulp change of u moves it by 7e-5).
    (tools/micro/warp_fp64_probe.py prints the same numbers; profiles/r04_warp_kernel_vs_fp64.txt)"""
    from vivid_amd import _lib as L
    from vivid_amd.geometry import compose_geometry, geometry_stats
    rows, S = 4, 256
    g = torch.Generator().manual_seed(6)
    depth = torch.rand(rows, 1, S, S, generator=g) * 4 + 1
    th = 0.05 * torch.randn(rows, generator=g)
    Rm = torch.zeros(rows, 3, 3)
    Rm[:, 0, 0], Rm[:, 0, 2], Rm[:, 1, 1], Rm[:, 2, 0], Rm[:, 2, 2] = th.cos(), th.sin(), 1.0, -th.sin(), th.cos()
    K = (torch.tensor([57.7, 57.7, 32.0, 32.0]) * 4).expand(rows, 4)
    geo = compose_geometry(torch.cat([Rm, 0.1 * torch.randn(rows, 3, 1, generator=g)], dim=2), K, K, imsize=S)
    freqs, phases = 2 * math.pi * torch.randn(128, generator=g), 2 * math.pi * torch.rand(128, generator=g)
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
    uv64 = R.warp_grid(depth.double().permute(0, 2, 3, 1), geo.double(), grid)            # the reference's formula, evaluated in fp64
    uvo = R.warp_grid(depth.permute(0, 2, 3, 1), geo, grid.float()).double()              # ... and as the reference runs it: fp32
    ulp = 2.0 ** -15                                                                        # ulp(fp32) of a coordinate in [256, 512)
    assert float(uv64.abs().max()) < 512
    ae, aeo = (uv.cpu().double() - uv64).abs().flatten(), (uvo - uv64).abs().flatten()
    assert float(ae.max()) < 3 * ulp and float(ae.quantile(0.999)) < 1.5 * ulp, (float(ae.max()) / ulp, float(ae.quantile(0.999)) / ulp)
    for q in (0.5, 0.99, 0.999):
        assert float(ae.quantile(q)) <= 1.25 * float(aeo.quantile(q)) + 0.25 * ulp, q
    f64, p64 = freqs[:64].double(), phases[:64].double()

    def emb(c):       # [rows, S, S, 2] -> [rows, S, S, 128], channel = 64 * axis + k (training/utils.py:214-215, MPFourier models.py:96-101)
        return torch.cat([torch.cos(c[..., 0:1] * f64 + p64), torch.cos(c[..., 1:2] * f64 + p64)], -1) * math.sqrt(2)
    exact = emb(uv64)
    assert rel_l2(wf.cpu(), exact) < 6e-5
    assert rel_l2(emb(uv.cpu().double()), exact) < 5.5e-5       # the coordinate error alone
    assert rel_l2(gf.cpu(), emb(grid)) < 3e-5                   # the un-warped grid: exact coordinates, fp32 cos only
