#!/usr/bin/env python3
"""Generate golden fixtures by running the REFERENCE itself on CPU.

Runs only in the build container (needs /root/reference, which never travels
to the GPU box).  Writes small .npz files next to this script; the tests read
only those files.  Nothing of the reference's source is stored: the fixtures
hold seeded inputs and the reference's outputs (sub-sampled where large).

  python tests/golden/make_fixtures.py            # all cases
  python tests/golden/make_fixtures.py tiny_dual  # one case

Import notes: `training/utils.py` imports kornia and
`training/custom_litdata_loader.py` imports litdata/torchvision at module top;
none of them is touched by the denoiser path, and none is installed here, so
empty placeholder modules are registered for the import to succeed.
Weights come from `vivid_amd.weights.synth_state_dict` and are loaded into the
reference module with strict=True, which also pins the state_dict key names.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("VIVID_REFERENCE", "/root/reference")

from vivid_amd.arch import NetConfig          # noqa: E402
from vivid_amd.weights import synth_state_dict  # noqa: E402
from tests.golden.cases import CASES, make_inputs, subsample, x_for, make_randn_like  # noqa: E402


def import_reference(snapshot=False):
    sys.dont_write_bytecode = True
    for name in ["kornia", "litdata", "torchvision", "torchvision.transforms", "torchvision.transforms.functional"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    for m in [k for k in sys.modules if k == "training" or k.startswith("training.") or k == "generate_images"]:
        del sys.modules[m]
    paths = [os.path.join(REF, "experiments", "code"), REF] if snapshot else [REF]
    for p in reversed(paths):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    import training.models as models
    import generate_images as gen
    return models, gen


def build_ref_net(models, cfg: NetConfig, seed, snapshot=False):
    kw = dict(img_resolution=cfg.img_resolution, img_channels=cfg.img_channels,
              model_channels=cfg.model_channels, channel_mult=list(cfg.channel_mult),
              num_blocks=cfg.num_blocks, attn_resolutions=list(cfg.attn_resolutions),
              extra_attn=cfg.extra_attn, use_fp16=False, super_res=cfg.super_res,
              no_time_enc=cfg.no_time_enc, depth_input=cfg.depth_input,
              warp_depth_coor=cfg.warp_depth_coor, uncond=cfg.uncond, noisy_sr=cfg.noisy_sr)
    if cfg.channel_mult_noise is not None:
        kw["channel_mult_noise"] = cfg.channel_mult_noise
    if cfg.channel_mult_emb is not None:
        kw["channel_mult_emb"] = cfg.channel_mult_emb
    if tuple(cfg.resample_filter) != (1.0, 1.0):
        kw["resample_filter"] = list(cfg.resample_filter)
    if snapshot:
        kw["label_dim"] = cfg.source_label_dim
    else:
        kw.update(source_label_dim=cfg.source_label_dim, target_label_dim=cfg.target_label_dim)
    net = models.NVPrecond(**kw).eval().requires_grad_(False)
    sd = synth_state_dict(cfg, seed=seed)
    missing = net.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return net, sd


def run_case(name):
    case = CASES[name]
    snapshot = case.get("snapshot", False)
    models, gen = import_reference(snapshot)
    cfg = case["cfg"]
    net, _ = build_ref_net(models, cfg, case["seed"], snapshot)
    if cfg.super_res:
        net.noisy_sr = 0.0                         # forward draws randn internally (:658); pinned off
    inp = make_inputs(case)
    out = {}
    torch.manual_seed(0)
    with torch.no_grad():
        for i, sigma in enumerate(case.get("sigmas", [])):
            sig = torch.full((inp["src"].shape[0],), float(sigma))
            x = x_for(inp, sigma)
            D, lv = net(inp["src"], x, sig, inp["geometry"], inp.get("cond"), return_logvar=True)
            out[f"D_{i}"] = D.numpy()
            out[f"logvar_{i}"] = lv.numpy()
            if i == 0:
                feats = net(inp["src"], x, sig, inp["geometry"], inp.get("cond"), return_features=True)
                out["n_features"] = np.array(len(feats))
                for j, f in enumerate(feats):
                    out[f"feat_shape_{j}"] = np.array(f.shape)
                    out[f"feat_{j}"] = subsample(f).numpy()
        if "sampler" in case:
            sk = dict(case["sampler"])
            guidance = sk.get("guidance", 1)
            gnet = None
            if guidance != 1:
                ucfg = case["gcfg"]
                unet_, _ = build_ref_net(models, ucfg, case["seed"] + 1, snapshot)
                if snapshot:
                    gnet = unet_
                else:
                    # HEAD's gnet(src, x, t) raises (geometry=None, no zero-feature branch in the
                    # dual-source forward).  The adapter supplies what the uncond
                    # branch means: zero geometry and zero features (models.py:631, 727-736).
                    fshapes = [f.shape for f in net(inp["src"], inp["img"], torch.ones(inp["src"].shape[0]), inp["geometry"], inp.get("cond"), return_features=True)]
                    def gnet(src, x, t, _n=unet_, _fs=fshapes):
                        z = [torch.zeros(s) for s in _fs]
                        return _n(src, x, t, torch.zeros(src.shape[0], 20), None, inject_features=z)
            lat = gen.edm_sampler(net, inp["src"], inp["noise"], labels=inp["geometry"], gnet=gnet,
                                  conditioning_image=inp.get("cond"), randn_like=make_randn_like(case["seed"]), **sk)
            out["sampler_out"] = lat.numpy()
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB), keys={len(out)}")


if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    for n in names:
        run_case(n)
