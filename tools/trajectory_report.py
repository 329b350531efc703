#!/usr/bin/env python3
"""Error of the HIP sampler's state against the CPU oracle's after every denoiser call of the reference-true cascade
(vivid-base@64 + guidance, 32 steps = 63 calls; vivid-sr@256, 16 steps = 31 calls) - the table DESIGN.md 4 quotes.
  python tools/trajectory_report.py            (GPU box; ~3 min of host time for the oracle)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_hip_trajectory import trajectory_errors  # noqa: E402

res = trajectory_errors()
for prec, r in res.items():
    print(f"== {prec}: base stage final rel-L2 {r['base_final']:.3e} (uint8 max diff {r['u8_base']}), "
          f"SR stage final {r['sr_final']:.3e} (uint8 max diff {r['u8_sr']}, differing pixels {100 * r['u8_sr_frac']:.3f} %)")
    pc = r["base_per_call"]
    print("   base stage, state entering call k (sigma): " + "  ".join(f"{k}({r['sigmas'][k]:.3g}):{pc[k]:.1e}" for k in range(0, len(pc), 4)))
    ps = r["sr_per_call"]
    print("   SR stage,   state entering call k:         " + "  ".join(f"{k}:{ps[k]:.1e}" for k in range(0, len(ps), 3)))
