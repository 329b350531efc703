#!/usr/bin/env python3
"""Guided evaluation (net + gnet + sampler update) wall time with the guidance net on a side stream vs serial, per batch size.
  python tools/overlap_probe.py [--res 64] [--batches 1,2,4,8,16,32]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vivid_amd
from vivid_amd.sampler import guided_denoise

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=64)
ap.add_argument("--batches", default="1,2,4,8,16,32")
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
R = a.res
cfg, ucfg = vivid_amd.vivid_base(R), vivid_amd.vivid_uncond(R)
net = vivid_amd.NVPrecond.from_config(cfg); net.load_state_dict(vivid_amd.synth_state_dict(cfg, seed=0)); net = net.cuda()
gnet = vivid_amd.NVPrecond.from_config(ucfg); gnet.load_state_dict(vivid_amd.synth_state_dict(ucfg, seed=1)); gnet = gnet.cuda()
for B in [int(v) for v in a.batches.split(",")]:
    g = torch.Generator().manual_seed(B)
    src = (torch.rand(2 * B, 3, R, R, generator=g) * 2 - 1).cuda()
    x = torch.randn(2 * B, 3, R, R, generator=g).cuda() * 5
    geo = torch.randn(2 * B, 20, generator=g).cuda()
    sig = torch.full((2 * B,), 5.0).cuda()
    res = {}
    for ov in (False, True):
        for _ in range(3):
            guided_denoise(net, gnet, src, x, sig, geo, guidance=1.5, overlap=ov)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            guided_denoise(net, gnet, src, x, sig, geo, guidance=1.5, overlap=ov)
        torch.cuda.synchronize()
        res[ov] = (time.perf_counter() - t0) / a.reps * 1e3
    print(f"res {R} batch {B:3d} ({2 * B * R * R} input pixels): serial {res[False]:.2f} ms, two streams {res[True]:.2f} ms -> {res[False] / res[True]:.2f}x", flush=True)
