#!/usr/bin/env python3
"""Compare per-block activations of two precision modes (debug aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, vivid_amd
from tests.golden.cases import CASES, make_inputs, x_for
name = sys.argv[1] if len(sys.argv) > 1 else "tiny_vanilla"
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
case = CASES[name]
dual = not case.get("snapshot", False)
taps = {}
for prec in ("fp32", "bf16x3"):
    net = vivid_amd.NVPrecond.from_config(case["cfg"], dual_source=dual, precision=prec)
    net.load_state_dict(vivid_amd.synth_state_dict(case["cfg"], seed=case["seed"]))
    net = net.cuda()
    inp = {k: v.cuda() for k, v in make_inputs(case).items()}
    sig = torch.full((inp["src"].shape[0],), sigma, device="cuda")
    d = {}
    net.trace(inp["src"], x_for(inp, sigma), sig, inp["geometry"], inp.get("cond"), hook=lambda n, t: d.__setitem__(n, t.cpu()))
    taps[prec] = d
for k in taps["fp32"]:
    a, b = taps["fp32"][k], taps["bf16x3"][k]
    e = float((a - b).norm() / a.norm().clamp_min(1e-30))
    print(f"{k:45s} {tuple(a.shape)} rel={e:.2e} finite={bool(torch.isfinite(b).all())}")
