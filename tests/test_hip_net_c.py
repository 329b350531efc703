"""GPU: the C-level whole-network entry points (vh_net_*, include/vivid_hip.h; csrc/net.hip) against vivid_amd.NVPrecond.

NVPrecond.forward (training/models.py:628-749) exists twice in this build: the Python engine (vivid_amd/engine.py, pinned to the
reference by the golden fixtures and the oracle) and the C++ walk a host without Python calls.  Both emit the same kernels with
the same arguments, so their outputs must be EQUAL, not close - any difference is a divergence of the two walks.  One case per
configuration family: dual-source with cross-attention at fused and unfused (4x4) levels, the unconditional guidance net
(closed-form zero keys), super-resolution (32-channel heads, conditioning image), depth-warp features, depth input, the
single-source forward, and the reference's real base@64 preset at two batch sizes; the C net is also held to the golden D_x."""
import os

import numpy as np
import pytest
import torch

from tests.conftest import rel_l2
from tests.golden.cases import CASES, make_inputs, make_randn_like, x_for

pytestmark = pytest.mark.gpu


def _pair(cfg, seed, dual=True, precision="bf16x3"):
    import vivid_amd
    from vivid_amd.cnet import CNet
    if cfg.super_res:               # the SR net's conditioning noise (training/models.py:658) off on both sides; test_c_net_noisy_sr_... has it on
        cfg = vivid_amd.NetConfig(**{**cfg.to_dict(), "noisy_sr": 0.0})
    sd = vivid_amd.synth_state_dict(cfg, seed=seed)
    py = vivid_amd.NVPrecond.from_config(cfg, dual_source=dual, precision=precision)
    py.load_state_dict(sd, strict=True)
    py.noisy_sr = 0.0
    cn = CNet(cfg, dual_source=dual, precision=precision)
    cn.load_state_dict(sd)
    return py.cuda(), cn


@pytest.mark.parametrize("name", ["tiny_dual", "tiny_sr", "tiny_warp", "tiny_warp_zero", "tiny_depth", "tiny_vanilla", "tiny_opts"])
def test_c_net_equals_python_engine_and_golden(name, golden_dir):
    case = CASES[name]
    dual = not case.get("snapshot", False)
    py, cn = _pair(case["cfg"], case["seed"], dual)
    inp = {k: v.cuda() for k, v in make_inputs(case).items()}
    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    for i, sigma in enumerate(case["sigmas"]):
        sig = torch.full((inp["src"].shape[0],), float(sigma), device="cuda")
        x = x_for(inp, sigma)
        a = py(inp["src"], x, sig, inp["geometry"], inp.get("cond"))
        b = cn(inp["src"], x, sig, inp["geometry"], inp.get("cond"))
        torch.cuda.synchronize()
        assert torch.equal(a, b), (name, sigma, rel_l2(b.cpu(), a.cpu()))
        assert rel_l2(b.cpu(), g[f"D_{i}"]) < 1e-4              # the reference's own D_x for this case and noise level


@pytest.mark.parametrize("name", ["tiny_dual", "tiny_sr", "tiny_opts", "tiny_vanilla"])
def test_c_net_fp32_mode_equals_python_engine_and_golden(name, golden_dir):
    """vh_net_config.fp32 = 1: the exact-fp32 walk of the C net (VH_PREC_F32 convolutions, vh_qkv_split / vh_attention, mp_silu / mp_cat in the
    loaders, no S8 tensors) against NVPrecond(precision='fp32') - equal bits - and the reference's golden D_x at the fp32 mode's tolerance."""
    case = CASES[name]
    dual = not case.get("snapshot", False)
    py, cn = _pair(case["cfg"], case["seed"], dual, precision="fp32")
    inp = {k: v.cuda() for k, v in make_inputs(case).items()}
    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    for i, sigma in enumerate(case["sigmas"]):
        sig = torch.full((inp["src"].shape[0],), float(sigma), device="cuda")
        x = x_for(inp, sigma)
        a = py(inp["src"], x, sig, inp["geometry"], inp.get("cond"))
        b = cn(inp["src"], x, sig, inp["geometry"], inp.get("cond"))
        torch.cuda.synchronize()
        assert torch.equal(a, b), (name, sigma, rel_l2(b.cpu(), a.cpu()))
        assert rel_l2(b.cpu(), g[f"D_{i}"]) < 2e-5


def test_c_net_uncond_guidance_net():
    case = CASES["tiny_guided"]
    py, cn = _pair(case["gcfg"], case["seed"] + 1)
    inp = {k: v.cuda() for k, v in make_inputs(case).items()}
    sig = torch.full((inp["src"].shape[0],), 1.3, device="cuda")
    x = x_for(inp, 1.3)
    a = py(inp["src"], x, sig)
    b = cn(None, x, sig, None, None)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


@pytest.mark.parametrize("batch", [1, 3])
def test_c_net_reference_base_preset(batch):
    import vivid_amd
    py, cn = _pair(vivid_amd.vivid_base(64), 0)
    g = torch.Generator().manual_seed(batch)
    src = (torch.rand(2 * batch, 3, 64, 64, generator=g) * 2 - 1).cuda()
    x = (torch.randn(2 * batch, 3, 64, 64, generator=g) * 3).cuda()
    geo = torch.randn(2 * batch, 20, generator=g).cuda()
    sig = torch.tensor([3.0, 3.0, 0.2, 0.2, 40.0, 40.0][:2 * batch]).cuda()
    a = py(src, x, sig, geo)
    for _ in range(2):                                     # replay is repeatable
        b = cn(src, x, sig, geo)
        torch.cuda.synchronize()
        assert torch.equal(a, b)


@pytest.mark.parametrize("name", ["tiny_dual", "tiny_sr", "tiny_warp"])
def test_c_net_split_evaluation_equals_whole(name):
    """vh_net_encode + vh_net_run_bound (the sampler's split evaluation: encoder once per noise level into a feature slot, UNet on that
    slot in place) == vh_net_run, bit for bit; the two slots are independent."""
    case = CASES[name]
    py, cn = _pair(case["cfg"], case["seed"])
    inp = {k: v.cuda() for k, v in make_inputs(case).items()}
    sigmas = case["sigmas"][:2] if len(case["sigmas"]) > 1 else case["sigmas"] * 2
    whole = []
    for slot, sigma in enumerate(sigmas):
        sig = torch.full((inp["src"].shape[0],), float(sigma), device="cuda")
        whole.append(cn(inp["src"], x_for(inp, sigma), sig, inp["geometry"], inp.get("cond")))
        cn.encode(slot, inp["src"], sig, inp["geometry"])
    for slot, sigma in reversed(list(enumerate(sigmas))):            # both slots were filled before either is read
        sig = torch.full((inp["src"].shape[0],), float(sigma), device="cuda")
        b = cn.run_bound(slot, inp["src"], x_for(inp, sigma), sig, inp["geometry"], inp.get("cond"))
        torch.cuda.synchronize()
        assert torch.equal(b, whole[slot]), (name, slot)
        x2 = x_for(inp, sigma) * 0.5                                 # same features, another x: equals a whole evaluation of that x
        b2 = cn.run_bound(slot, inp["src"], x2, sig, inp["geometry"], inp.get("cond"))
        assert torch.equal(b2, cn(inp["src"], x2, sig, inp["geometry"], inp.get("cond")))


def test_c_net_refuses_missing_inputs():
    from vivid_amd import _lib as L
    case = CASES["tiny_dual"]
    _, cn = _pair(case["cfg"], case["seed"])
    inp = {k: v.cuda() for k, v in make_inputs(case).items()}
    sig = torch.full((inp["src"].shape[0],), 1.0, device="cuda")
    with pytest.raises(L.VividHipError, match="geometry is required"):
        cn(inp["src"], x_for(inp, 1.0), sig, None)
    with pytest.raises(L.VividHipError, match="reads src"):
        cn(None, x_for(inp, 1.0), sig, inp["geometry"])


def test_c_net_noisy_sr_conditioning_noise():
    """The reference adds noisy_sr * randn_like(cond) to the conditioning image on EVERY forward, also at inference (training/models.py:658;
    0.25 in --preset=vivid-sr).  vh_net_config.noisy_sr + the `cond_noise` argument: with the draws torch would have made, the C net equals
    vivid_amd.NVPrecond bit for bit; without them it refuses to run."""
    import vivid_amd
    from vivid_amd import _lib as L
    from vivid_amd.cnet import CNet
    case = CASES["tiny_sr"]
    cfg = vivid_amd.NetConfig(**{**case["cfg"].to_dict(), "noisy_sr": 0.25})
    sd = vivid_amd.synth_state_dict(cfg, seed=case["seed"])
    py = vivid_amd.NVPrecond.from_config(cfg, precision="bf16x3")
    py.load_state_dict(sd, strict=True)
    py = py.cuda()
    assert py.noisy_sr == 0.25
    cn = CNet(cfg)
    cn.load_state_dict(sd)
    inp = {k: v.cuda() for k, v in make_inputs(case).items()}
    sig = torch.full((inp["src"].shape[0],), 2.0, device="cuda")
    x = x_for(inp, 2.0)
    torch.manual_seed(11)
    a = py(inp["src"], x, sig, inp["geometry"], inp["cond"])
    torch.manual_seed(11)
    noise = torch.randn_like(inp["cond"])
    b = cn(inp["src"], x, sig, inp["geometry"], inp["cond"], noise)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert not torch.equal(a, cn(inp["src"], x, sig, inp["geometry"], inp["cond"], torch.zeros_like(noise)))
    with pytest.raises(L.VividHipError, match="noisy_sr"):
        cn(inp["src"], x, sig, inp["geometry"], inp["cond"])


@pytest.mark.parametrize("name", ["tiny_dual", "tiny_sr", "tiny_vanilla"])
def test_c_net_feature_lists_and_logvar(name):
    """The rest of NVPrecond.forward's protocol behind the C ABI (training/models.py:664-670, 685-688): `return_features` as caller-visible
    NCHW tensors (vh_net_features), `inject_features` (vh_net_run_inject: the encoder's own list, an edited list, the zero list) and
    the logvar head (vh_net_logvar) - each equal to vivid_amd.NVPrecond, bit for bit."""
    case = CASES[name]
    dual = not case.get("snapshot", False)
    py, cn = _pair(case["cfg"], case["seed"], dual)
    inp = {k: v.cuda() for k, v in make_inputs(case).items()}
    sigma = case["sigmas"][0]
    sig = torch.full((inp["src"].shape[0],), float(sigma), device="cuda")
    x = x_for(inp, sigma)
    fa = py(inp["src"], x, sig, inp["geometry"], inp.get("cond"), return_features=True)
    fb = cn.features(inp["src"], sig, inp["geometry"])
    assert len(fa) == len(fb) == len(cn.feature_shapes()) > 0
    for u, v in zip(fa, fb):
        assert u.shape == v.shape and torch.equal(u.contiguous(), v)
    whole = cn(inp["src"], x, sig, inp["geometry"], inp.get("cond"))
    assert torch.equal(cn.run_inject(inp["src"], x, sig, inp["geometry"], fb, inp.get("cond")), whole)
    for edit in (lambda f: f * 0.5, torch.zeros_like):
        fl = [edit(f) for f in fb]
        a = py(inp["src"], x, sig, inp["geometry"], inp.get("cond"), inject_features=fl)
        b = cn.run_inject(inp["src"], x, sig, inp["geometry"], fl, inp.get("cond"))
        torch.cuda.synchronize()
        assert torch.equal(a, b) and not torch.equal(b, whole)
    _, lv = py(inp["src"], x, sig, inp["geometry"], inp.get("cond"), return_logvar=True)
    assert torch.equal(lv, cn.logvar(sig))


def _torch_schedule(num_steps, sigma_min=0.002, sigma_max=80, rho=7):
    idx = torch.arange(num_steps, dtype=torch.float32)
    t = (sigma_max ** (1 / rho) + idx / (num_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
    return torch.cat([t, torch.zeros_like(t[:1])])


@pytest.mark.parametrize("name,extra,noisy", [("tiny_guided", {}, 0.0), ("tiny_nte", {}, 0.0), ("tiny_sr", {}, 0.0), ("tiny_sr", {"num_steps": 3}, 0.25),
                                              ("tiny_dual", {"num_steps": 4, "S_churn": 2.0}, 0.0), ("tiny_guided", {"num_steps": 5, "pipeline": False}, 0.0)])
def test_c_sampler_equals_python_sampler(name, extra, noisy):
    """vh_edm_sampler (generate_images.py:43-118 behind the C ABI) against vivid_amd.edm_sampler: guided (two nets, the guidance net on a
    second stream), no_time_enc with churn (features once; churn noise through the caller's randn), the SR net without and WITH its
    per-call conditioning noise, churn on a dual-source net (no two calls share a level), and the plain whole-evaluation call pattern.
    Same kernels, same order, same draws: the samples are EQUAL.  The library's own schedule arithmetic (no t_steps given) lands within
    an ulp of torch's levels: the samples then agree to 1e-5."""
    import vivid_amd
    from vivid_amd.cnet import CNet
    case = CASES[name]
    cfg = case["cfg"] if not case["cfg"].super_res else vivid_amd.NetConfig(**{**case["cfg"].to_dict(), "noisy_sr": noisy})
    sd = vivid_amd.synth_state_dict(cfg, seed=case["seed"])
    py = vivid_amd.NVPrecond.from_config(cfg, precision="bf16x3")
    py.load_state_dict(sd, strict=True)
    py = py.cuda()
    cn = CNet(cfg)
    cn.load_state_dict(sd)
    gpy = gcn = None
    if "gcfg" in case:
        gsd = vivid_amd.synth_state_dict(case["gcfg"], seed=case["seed"] + 1)
        gpy = vivid_amd.NVPrecond.from_config(case["gcfg"], precision="bf16x3")
        gpy.load_state_dict(gsd, strict=True)
        gpy = gpy.cuda()
        gcn = CNet(case["gcfg"])
        gcn.load_state_dict(gsd)
    inp = {k: v.cuda() for k, v in make_inputs(case).items()}
    kw = {**case["sampler"], **extra}
    pipeline = kw.pop("pipeline", True)
    if not pipeline:
        os.environ["VIVID_FEATURE_PIPELINE"] = "0"
    try:
        torch.manual_seed(3)
        want = vivid_amd.edm_sampler(py, inp["src"], inp["noise"], labels=inp["geometry"], gnet=gpy if gpy is not None else py,
                                     conditioning_image=inp.get("cond"), randn_like=make_randn_like(case["seed"]), **kw)
    finally:
        os.environ.pop("VIVID_FEATURE_PIPELINE", None)
    churn_rl = make_randn_like(case["seed"])
    rows_elems = inp["noise"].numel()

    def randn(n):      # churn noise (the whole state) from the test's repeatable CPU stream, conditioning noise from torch's CUDA generator - as the Python path
        if n == rows_elems and kw.get("S_churn", 0) > 0:
            return churn_rl(torch.empty(n, device="cuda"))
        return torch.randn(n, device="cuda")
    torch.manual_seed(3)
    got = cn.edm_sampler(inp["src"], inp["noise"], labels=inp["geometry"], gnet=gcn, conditioning_image=inp.get("cond"), randn=randn,
                         t_steps=_torch_schedule(kw["num_steps"]), pipeline=pipeline, **kw)
    assert got.shape == want.shape and torch.equal(got, want), (name, rel_l2(got.cpu(), want.cpu()))
    if not noisy:
        churn_rl = make_randn_like(case["seed"])
        own = cn.edm_sampler(inp["src"], inp["noise"], labels=inp["geometry"], gnet=gcn, conditioning_image=inp.get("cond"), randn=randn,
                             pipeline=pipeline, **kw)
        assert rel_l2(own.cpu(), want.cpu()) < 1e-5
