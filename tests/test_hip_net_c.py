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
from tests.golden.cases import CASES, make_inputs, x_for

pytestmark = pytest.mark.gpu


def _pair(cfg, seed, dual=True):
    import vivid_amd
    from vivid_amd.cnet import CNet
    sd = vivid_amd.synth_state_dict(cfg, seed=seed)
    py = vivid_amd.NVPrecond.from_config(cfg, dual_source=dual, precision="bf16x3")
    py.load_state_dict(sd, strict=True)
    py.noisy_sr = 0.0
    cn = CNet(cfg, dual_source=dual)
    cn.load_state_dict(sd)
    return py.cuda(), cn


@pytest.mark.parametrize("name", ["tiny_dual", "tiny_sr", "tiny_warp", "tiny_warp_zero", "tiny_depth", "tiny_vanilla"])
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
