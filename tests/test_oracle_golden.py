"""CPU: the oracle (oracle/vivid_ref.py) against golden vectors produced by the reference itself
(tests/golden/make_fixtures.py).  This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import vivid_ref as R
from tests.conftest import rel_l2
from tests.golden.cases import CASES, make_inputs, make_randn_like, subsample, x_for
from vivid_amd.weights import synth_state_dict

TOL = 2e-5   # fp32, same torch CPU kernels on both sides; differences are op-order only


def _cfg(c):
    d = c.to_dict()
    d.pop("use_fp16")
    return R.make_config(**d)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, f"{name}.npz"))


@pytest.mark.parametrize("name", list(CASES))
def test_denoiser_matches_reference(name, golden_dir):
    case = CASES[name]
    if not case.get("sigmas"):
        pytest.skip("sampler-only case")
    g = _load(golden_dir, name)
    cfg = _cfg(case["cfg"])
    sd = synth_state_dict(case["cfg"], seed=case["seed"])
    inp = make_inputs(case)
    dual = not case.get("snapshot", False)
    for i, sigma in enumerate(case["sigmas"]):
        sig = torch.full((inp["src"].shape[0],), float(sigma))
        with torch.no_grad():
            D, lv = R.nvprecond_forward(sd, cfg, inp["src"], x_for(inp, sigma), sig, inp["geometry"], inp.get("cond"),
                                        return_logvar=True, dual=dual)
        assert D.shape == g[f"D_{i}"].shape
        assert rel_l2(D, g[f"D_{i}"]) < TOL, (name, sigma)
        assert rel_l2(lv, g[f"logvar_{i}"]) < TOL
        if i == 0:
            with torch.no_grad():
                feats = R.nvprecond_forward(sd, cfg, inp["src"], x_for(inp, sigma), sig, inp["geometry"], inp.get("cond"),
                                            return_features=True, dual=dual)
            assert len(feats) == int(g["n_features"])
            for j, f in enumerate(feats):
                assert tuple(f.shape) == tuple(g[f"feat_shape_{j}"])
                assert rel_l2(subsample(f), g[f"feat_{j}"]) < TOL, (name, "feature", j)


@pytest.mark.parametrize("name", [n for n in CASES if "sampler" in CASES[n]])
def test_sampler_matches_reference(name, golden_dir):
    case = CASES[name]
    g = _load(golden_dir, name)
    dual = not case.get("snapshot", False)
    net = R.OracleNet(_cfg(case["cfg"]), synth_state_dict(case["cfg"], seed=case["seed"]), dual=dual)
    gnet = None
    if "gcfg" in case:
        gnet = R.OracleNet(_cfg(case["gcfg"]), synth_state_dict(case["gcfg"], seed=case["seed"] + 1), dual=dual)
    inp = make_inputs(case)
    out = R.edm_sampler(net, inp["src"], inp["noise"], labels=inp["geometry"], gnet=gnet,
                        conditioning_image=inp.get("cond"), randn_like=make_randn_like(case["seed"]), **case["sampler"])
    assert out.shape == g["sampler_out"].shape
    assert rel_l2(out, g["sampler_out"]) < 1e-4, name


def test_explicit_attention_equals_sdpa():
    case = CASES["tiny_dual"]
    cfg = _cfg(case["cfg"])
    sd = synth_state_dict(case["cfg"], seed=1)
    inp = make_inputs(case)
    sig = torch.full((inp["src"].shape[0],), 1.5)
    with torch.no_grad():
        a = R.nvprecond_forward(sd, cfg, inp["src"], x_for(inp, 1.5), sig, inp["geometry"])
        b = R.nvprecond_forward(sd, cfg, inp["src"], x_for(inp, 1.5), sig, inp["geometry"], explicit_attn=True)
    assert rel_l2(a, b) < 1e-5


def test_codec_and_rank_split():
    x = torch.arange(0, 256, dtype=torch.uint8).reshape(1, 1, 16, 16)
    assert torch.equal(R.decode_latents(R.encode_latents(x)), x)
    # generate_images.py:199-200 with N=128, max_batch=16, world=8 -> one batch of 16 per rank
    parts = [R.rank_batches(128, 16, 8, r) for r in range(8)]
    assert all(len(p) == 1 and len(p[0]) == 16 for p in parts)
    assert sorted(np.concatenate([q for p in parts for q in p]).tolist()) == list(range(128))
