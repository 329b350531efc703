#!/usr/bin/env python3
"""Write a network snapshot with the REFERENCE's own persistence machinery (build container only).

  python tests/golden/make_snapshot.py OUT.pkl [case]

Mirrors what `training/training_loop.py:485-496` dumps: EasyDict(encoder, dataset_kwargs, loss_fn, ema=<fp16 net>).
The network is the reference's NVPrecond for the given golden case, carrying `synth_state_dict(cfg, seed)`.
The file embeds the reference's module source as text (that is the format), so it is written to a temporary path
by the test that needs it and never committed.
"""
import copy
import os
import pickle
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests.golden.cases import CASES                                # noqa: E402
from tests.golden.make_fixtures import build_ref_net, import_reference  # noqa: E402


def main(out_path, case_name="tiny_dual"):
    import torch
    case = CASES[case_name]
    models, _ = import_reference(False)
    import dnnlib
    import training.encoders as encoders
    net, _ = build_ref_net(models, case["cfg"], case["seed"], False)
    data = dnnlib.EasyDict(encoder=encoders.StandardRGBEncoder(), dataset_kwargs=dict(path="nowhere", split="train"), loss_fn=None)
    data.ema = copy.deepcopy(net).cpu().eval().requires_grad_(False).to(torch.float16)
    with open(out_path, "wb") as f:
        pickle.dump(data, f)
    print(f"wrote {out_path} ({os.path.getsize(out_path) >> 10} KiB)")


if __name__ == "__main__":
    main(*sys.argv[1:3])
