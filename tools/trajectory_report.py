#!/usr/bin/env python3
"""Error of the HIP sampler's state against the CPU oracle's after every denoiser call of the reference-true cascade
(vivid-base@64 + guidance, 32 steps = 63 calls; vivid-sr@256, 16 steps = 31 calls) - the table DESIGN.md 4 quotes.
  python tools/trajectory_report.py            (GPU box; ~3 min of host time for the oracle)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_hip_trajectory import base_stage, sr_stage  # noqa: E402

print("base stage ...", flush=True)
base = base_stage()
for prec in ("bf16x3", "fp32"):
    r = base[prec]
    pc = r["per_call"]
    print(f"== {prec}: base stage final rel-L2 {r['final']:.3e} (uint8 max diff {r['u8']}); state entering call k (sigma): "
          + "  ".join(f"{k}({r['sigmas'][k]:.3g}):{pc[k]:.1e}" for k in range(0, len(pc), 4)), flush=True)
print("SR stage ...", flush=True)
sr = sr_stage(base)
for prec, r in sr.items():
    ps = r["per_call"]
    print(f"== {prec}: SR stage final rel-L2 {r['final']:.3e} (uint8 max diff {r['u8']}, differing pixels {100 * r['u8_frac']:.3f} %); state entering call k: "
          + "  ".join(f"{k}:{ps[k]:.1e}" for k in range(0, len(ps), 3)), flush=True)
