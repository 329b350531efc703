#!/usr/bin/env python3
"""Print rel-L2 of the HIP path against the golden vectors for every fixture case and precision mode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import vivid_amd
from tests.conftest import rel_l2
from tests.golden.cases import CASES, make_inputs, make_randn_like, x_for


def mk(cfg, seed, dual, prec):
    net = vivid_amd.NVPrecond.from_config(cfg, dual_source=dual, precision=prec)
    net.load_state_dict(vivid_amd.synth_state_dict(cfg, seed=seed))
    if cfg.super_res:
        net.noisy_sr = 0.0
    return net.cuda()


for prec in sys.argv[1:] or ["fp32", "bf16x3"]:
    for name, case in CASES.items():
        g = np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz"))
        dual = not case.get("snapshot", False)
        net = mk(case["cfg"], case["seed"], dual, prec)
        inp = {k: v.cuda() for k, v in make_inputs(case).items()}
        errs = []
        for i, sigma in enumerate(case.get("sigmas", [])):
            sig = torch.full((inp["src"].shape[0],), float(sigma), device="cuda")
            D = net(inp["src"], x_for(inp, sigma), sig, inp["geometry"], inp.get("cond"))
            errs.append(rel_l2(D.cpu(), g[f"D_{i}"]))
        line = f"{prec:7s} {name:13s} D_x: " + " ".join(f"{e:.2e}" for e in errs)
        if "sampler" in case:
            gnet = mk(case["gcfg"], case["seed"] + 1, dual, prec) if "gcfg" in case else None
            out = vivid_amd.edm_sampler(net, inp["src"], inp["noise"], labels=inp["geometry"], gnet=gnet,
                                        conditioning_image=inp.get("cond"), randn_like=make_randn_like(case["seed"]), **case["sampler"])
            line += f"  sampler: {rel_l2(out.cpu(), g['sampler_out']):.2e}"
        print(line, flush=True)
