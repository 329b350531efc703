"""GPU: full-size architectures (the reference's real presets and BASELINE.json's literal 256^2 build).

* vivid-base @64^2 and vivid-sr @256^2 (the reference's own cascade stages, train_nvs.py:28-30) at batch 1:
  HIP vs the CPU oracle directly (the oracle runs these in a few seconds).
* base architecture built @256^2 (BASELINE configs[1]; S = 16384 queries x 49152 keys at the 128^2 level): the direct oracle
  comparisons at batch 1 and the timed-batch checks of BASELINE configs[1], [3] and [4] live in tests/test_hip_timed_configs.py;
  here the size-independent properties —
    - the two independent kernel families (exact-fp32 MFMA 128-tile kernels vs bf16x3 glds / x3-attention kernels)
      agree to 1e-4;
    - samples are independent: a batch of 2 reproduces the two batch-1 results (what multi-GPU sharding relies on,
      generate_images.py:199-200);
    - D(x, sigma) -> c_skip*x as sigma -> 0 (preconditioning, training/models.py:635-636,683).
"""
import pytest
import torch

from oracle import vivid_ref as R
from tests.conftest import rel_l2

pytestmark = pytest.mark.gpu


def _inputs(R_, B, seed, src_c=3):
    g = torch.Generator().manual_seed(seed)
    src = torch.rand(2 * B, src_c, R_, R_, generator=g) * 2 - 1
    img = torch.rand(2 * B, 3, R_, R_, generator=g) * 2 - 1
    eps = torch.randn(B, 3, R_, R_, generator=g).repeat_interleave(2, dim=0)
    geo = torch.randn(2 * B, 20, generator=g)
    geo[:, [14, 15, 18, 19]] = 0
    return src, img, eps, geo


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


@pytest.mark.parametrize("sigma", [5.0, 0.3])
def test_vivid_base_64_vs_oracle(sigma):
    import vivid_amd
    cfg = vivid_amd.vivid_base(64)
    net, sd = _net(cfg, 0, "bf16x3")
    src, img, eps, geo = _inputs(64, 1, 1)
    x = img + sigma * eps
    sig = torch.full((2,), sigma)
    D = net(src.cuda(), x.cuda(), sig.cuda(), geo.cuda())
    with torch.no_grad():
        ref = R.nvprecond_forward(sd, _ocfg(cfg), src, x, sig, geo)
    assert rel_l2(D.cpu(), ref) < 1e-4


def test_vivid_sr_256_vs_oracle():
    import vivid_amd
    cfg = vivid_amd.vivid_sr(256, noisy_sr=0.0)
    net, sd = _net(cfg, 2, "bf16x3")
    src, img, eps, geo = _inputs(256, 1, 3)
    g = torch.Generator().manual_seed(9)
    cond = torch.nn.functional.interpolate(torch.rand(1, 3, 64, 64, generator=g) * 2 - 1, size=(256, 256), mode="bilinear")
    sigma = 2.0
    x = img + sigma * eps
    sig = torch.full((2,), sigma)
    D = net(src.cuda(), x.cuda(), sig.cuda(), geo.cuda(), cond.cuda())
    with torch.no_grad():
        ref = R.nvprecond_forward(sd, _ocfg(cfg), src, x, sig, geo, cond)
    assert rel_l2(D.cpu(), ref) < 1e-4


def test_base_256_kernel_families_agree_and_samples_independent():
    import vivid_amd
    cfg = vivid_amd.vivid_base(256)
    src, img, eps, geo = _inputs(256, 2, 5)
    sigma = 3.0
    x = img + sigma * eps
    sig = torch.full((4,), sigma)
    outs = {}
    for prec in ("fp32", "bf16x3"):
        net, _ = _net(cfg, 0, prec)
        outs[prec] = net(src.cuda(), x.cuda(), sig.cuda(), geo.cuda()).cpu()
        if prec == "bf16x3":
            one = net(src[2:].cuda(), x[2:].cuda(), sig[2:].cuda(), geo[2:].cuda()).cpu()      # second sample alone
            tiny = net(src[:2].cuda(), x[:2].cuda(), torch.full((2,), 1e-3).cuda(), geo[:2].cuda()).cpu()
        del net
        torch.cuda.empty_cache()
    assert outs["fp32"].shape == (2, 3, 256, 256)
    assert rel_l2(outs["bf16x3"], outs["fp32"]) < 1e-4
    # same kernels and per-sample arithmetic, up to the tile / split-K partition the launcher picks per grid size
    assert rel_l2(one[0], outs["bf16x3"][1]) < 2e-5
    # sigma -> 0: c_skip -> 1, c_out -> sigma: D_x = x up to O(sigma)
    assert rel_l2(tiny, x[:2:2]) < 5e-3


def test_seeded_noise_is_placement_independent():
    """StackedRandomGenerator (generate_images.py:120-134): a sample's noise depends on its seed only."""
    import vivid_amd
    a = vivid_amd.StackedRandomGenerator("cuda", [16, 17, 18]).randn([3, 3, 8, 8], device="cuda")
    b = vivid_amd.StackedRandomGenerator("cuda", [18]).randn([1, 3, 8, 8], device="cuda")
    assert torch.equal(a[2], b[0])
