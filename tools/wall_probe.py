#!/usr/bin/env python3
"""Wall time of one denoiser evaluation against the sum of its kernel durations, with and without hipGraph replay.
  python tools/wall_probe.py [--res 64] [--batch 1] [--uncond | --sr] [--reps 20]
Launch-/latency-bound shapes (the reference's own base@64 and SR@256 presets at small batch) are judged on this ratio."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vivid_amd

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=64)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--uncond", action="store_true")
ap.add_argument("--sr", action="store_true")
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
cfg = vivid_amd.vivid_sr(a.res) if a.sr else (vivid_amd.vivid_uncond(a.res) if a.uncond else vivid_amd.vivid_base(a.res))
B, R = a.batch, a.res
g = torch.Generator().manual_seed(0)
src = (torch.rand(2 * B, 3, R, R, generator=g) * 2 - 1).cuda()
cond = (torch.rand(B, 3, R, R, generator=g) * 2 - 1).cuda() if a.sr else None
x = torch.randn(2 * B, 3, R, R, generator=g).cuda() * 5
geo = torch.randn(2 * B, 20, generator=g).cuda()
sig = torch.full((2 * B,), 5.0).cuda()
for graph in ("0", "1"):
    os.environ["VIVID_HIPGRAPH"] = graph
    net = vivid_amd.NVPrecond.from_config(cfg)
    net.load_state_dict(vivid_amd.synth_state_dict(cfg, seed=0))
    net = net.cuda()
    for _ in range(3):
        net(src, x, sig, geo, cond)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        net(src, x, sig, geo, cond)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.reps * 1e3
    # plan replay alone (no input copies, no output clone)
    prog = list(net._engine.programs.values())[0]
    t0 = time.perf_counter()
    for _ in range(a.reps):
        prog.plan.run()
    torch.cuda.synchronize()
    replay = (time.perf_counter() - t0) / a.reps * 1e3
    ctx = net._engine.ctx
    ctx.profile_enable(True)
    for _ in range(3):
        prog.plan.run()
    fam = ctx.profile_read()
    ctx.profile_enable(False)
    ksum = sum(v["ms"] for v in fam.values()) / 3
    nl = sum(v["launches"] for v in fam.values()) / 3
    print(f"res {R} batch {B} graph={graph}: net() wall {wall:.2f} ms, plan replay {replay:.2f} ms, sum of kernels {ksum:.2f} ms over {nl:.0f} launches "
          f"-> wall/kernel {wall / ksum:.2f}, replay/kernel {replay / ksum:.2f}, gap per launch {(replay - ksum) / nl * 1e3:.1f} us")
    del net
    torch.cuda.empty_cache()
