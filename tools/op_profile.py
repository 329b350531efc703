#!/usr/bin/env python3
"""Per-launch timing of one denoiser evaluation (HIP events via the C ABI), grouped by op shape.
  python tools/op_profile.py [--res 256] [--batch 16] [--uncond | --sr | --warp]
--sr: the super-resolution net (BASELINE config 4: --res 1024 --batch 4); --warp: base + depth-warp geometry (config 5)."""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vivid_amd

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--uncond", action="store_true")
ap.add_argument("--sr", action="store_true")
ap.add_argument("--warp", action="store_true")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--top", type=int, default=45, help="lines to print (0 = all)")
ap.add_argument("--knobs", default="", help="name=value,... passed to vh_set_knob (A/B runs)")
a = ap.parse_args()
for kv in filter(None, a.knobs.split(",")):
    from vivid_amd import _lib
    _lib.set_knob(kv.split("=")[0], int(kv.split("=")[1]))
if a.sr:
    cfg = vivid_amd.vivid_sr(a.res)
elif a.warp:
    cfg = vivid_amd.NetConfig(**{**vivid_amd.vivid_base(a.res).to_dict(), "warp_depth_coor": True})
else:
    cfg = vivid_amd.vivid_uncond(a.res) if a.uncond else vivid_amd.vivid_base(a.res)
net = vivid_amd.NVPrecond.from_config(cfg)
net.load_state_dict(vivid_amd.synth_state_dict(cfg, seed=0))
net = net.cuda()
B, R = a.batch, a.res
g = torch.Generator().manual_seed(0)
src = (torch.rand(2 * B, 3, R, R, generator=g) * 2 - 1).cuda()
if a.warp:
    src = torch.cat([src, (torch.rand(2 * B, 1, R, R, generator=g) * 4 + 1).cuda()], dim=1)
cond = (torch.rand(B, 3, R, R, generator=g) * 2 - 1).cuda() if a.sr else None
x = torch.randn(2 * B, 3, R, R, generator=g).cuda() * 5
geo = torch.randn(2 * B, 20, generator=g).cuda()
sig = torch.full((2 * B,), 5.0).cuda()
net(src, x, sig, geo, cond)
ctx = net._engine.ctx
ctx.profile_enable(True)
for _ in range(a.reps):
    net(src, x, sig, geo, cond)
recs = ctx.profile_read_list()
ctx.profile_enable(False)
prog = list(net._engine.programs.values())[0]
n = len(prog.oplog)
assert len(recs) == n * a.reps, (len(recs), n)
agg = collections.OrderedDict()
for i, d in enumerate(prog.oplog):
    ms = sum(recs[i + r * n][1] for r in range(a.reps)) / a.reps
    fam, _, fl, by = recs[i]
    e = agg.setdefault(d, [0, 0.0, 0.0, 0.0])
    e[0] += 1; e[1] += ms; e[2] += fl; e[3] += by
tot = sum(e[1] for e in agg.values())
print(f"total {tot:.1f} ms over {n} launches")
for d, (c, ms, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:(a.top or None)]:
    print(f"{ms:9.2f} ms {100*ms/tot:5.1f}%  x{c:3d}  {fl/ms/1e9 if ms else 0:7.1f} TF/s {by/ms/1e6 if ms else 0:8.1f} GB/s  {d}")
