#!/usr/bin/env python3
"""Board power and shader clock under each kernel family (is the chip at its power cap under the MFMA-bound kernels?).

  python tools/power_probe.py [--seconds 8]
Runs, back to back on one device: idle, a C2 3x3 convolution in a loop, the C2 cross-view attention in a loop, a vh_split (HBM-bound) in a loop,
whole C2 evaluations in a loop - while a thread samples the amdgpu hwmon files (power1_average / power1_input, power1_cap, freq1_input = sclk,
freq2_input = mclk, temp1_input) at ~10 Hz; falls back to `rocm-smi --showpower --showclocks --json` when sysfs is not readable.
Prints one line per phase (median and max power, median sclk) and the achieved rate of the looped op."""
import argparse
import glob
import json
import math
import os
import statistics
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vivid_amd  # noqa: E402
from vivid_amd import _lib as L  # noqa: E402


def hwmon_dirs():
    """hwmon directory of THE card this process computes on (by PCI address; a shared host shows every card's files)."""
    pr = torch.cuda.get_device_properties(0)
    addr = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    own = sorted(glob.glob(f"/sys/bus/pci/devices/{addr}/hwmon/hwmon*"))
    if own:
        return own, addr
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        if os.path.exists(os.path.join(d, "power1_average")) or os.path.exists(os.path.join(d, "power1_input")):
            out.append(d)
    return out, None


def read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except Exception:
        return None


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.dirs, self.addr = hwmon_dirs()
        self.rows, self.phase, self.stop = [], "init", False
        self.mode = "sysfs" if self.dirs else "rocm-smi"

    def sample(self):
        if self.mode == "sysfs":
            best = None
            for d in self.dirs:                     # the busiest card is ours (one GPU visible on the box; several hwmon dirs on a shared host)
                p = read_int(os.path.join(d, "power1_average")) or read_int(os.path.join(d, "power1_input"))
                if p is None:
                    continue
                r = {"power_w": p / 1e6, "cap_w": (read_int(os.path.join(d, "power1_cap")) or 0) / 1e6,
                     "sclk_mhz": (read_int(os.path.join(d, "freq1_input")) or 0) / 1e6, "mclk_mhz": (read_int(os.path.join(d, "freq2_input")) or 0) / 1e6,
                     "temp_c": (read_int(os.path.join(d, "temp1_input")) or 0) / 1e3}
                if best is None or r["power_w"] > best["power_w"]:
                    best = r
            return best
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            j = json.loads(o)
            best = None
            for card, v in j.items():
                p = next((float(x) for k, x in v.items() if "ower" in k and "(W)" in k), None)
                s = next((x for k, x in v.items() if k.startswith("sclk clock speed")), "")
                if p is None:
                    continue
                r = {"power_w": p, "cap_w": 0.0, "sclk_mhz": float(s.strip("()Mhz")) if s else 0.0, "mclk_mhz": 0.0, "temp_c": 0.0}
                if best is None or p > best["power_w"]:
                    best = r
            return best
        except Exception:
            return None

    def run(self):
        while not self.stop:
            r = self.sample()
            if r:
                r["phase"], r["t"] = self.phase, time.time()
                self.rows.append(r)
            time.sleep(0.1 if self.mode == "sysfs" else 0.5)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=8.0)
    a = ap.parse_args()
    smp = Sampler()
    print(f"sampling through {smp.mode}: " + (f"card at PCI {smp.addr}" if smp.addr else f"{len(smp.dirs)} hwmon dir(s), busiest card taken"))
    smp.start()
    ctx = L.Context(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(0)
    results = {}

    def loop(name, fn, work, unit):
        fn(); torch.cuda.synchronize()
        smp.phase = name
        t0, n = time.time(), 0
        while time.time() - t0 < a.seconds:
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            n += 10
        dt = time.time() - t0
        results[name] = (work * n / dt, unit)
        smp.phase = "gap"
        time.sleep(1.0)

    smp.phase = "idle"; time.sleep(2.0); smp.phase = "gap"
    # 3x3 convolution of the C2 128x128 level (wide tile, chunk-major K)
    rows, h, w, cin, cout = 32, 128, 128, 256, 256
    M = rows * h * w
    x = torch.randn(M, cin, generator=g).cuda()
    wgt = torch.randn(cout, cin, 3, 3, generator=g).cuda()
    zeros = torch.zeros(16384, device="cuda")
    s8 = torch.empty(M * cin, device="cuda")
    ctx.call("vh_split", L.SplitArgs(src0=x.data_ptr(), src1=None, c0=cin, c1=0, scale0=1.0, scale1=1.0, pro=0, npix=M, c_pad=cin, out=s8.data_ptr(), out_raw=None))
    wt = torch.zeros(9 * cin // 4 * cout * 4, device="cuda")
    ctx.call("vh_prep_weight", L.PrepWeightArgs(w=wgt.data_ptr(), cout=cout, cin=cin, taps=9, cin_pad=cin, k_pad=9 * cin, gain_ptr=None, gain_value=1.0,
                                                wt=wt.data_ptr(), dst_col0=0, dst_cols=cout, split=2))
    out = torch.empty(M, cout, device="cuda")
    ca = L.ConvArgs(src0=s8.data_ptr(), src1=None, c0=cin, c1=0, scale0=1.0, scale1=1.0, rows=rows, h=h, w=w, up=0, taps=9, pro=0, wt=wt.data_ptr(),
                    cin_pad=cin, k_pad=9 * cin, zeros=zeros.data_ptr(), zeros_bytes=65536, scratch=None, scratch_floats=0, cout=cout, out=out.data_ptr(),
                    out_s8=None, out_s8_c=0, prec=1, kernel=1, epi=0, cvec=None, cvec_ld=0, res=None, res_up=0, ta=0, tb=0, clip=0)
    loop("conv3x3 32x128x128 256->256", lambda: ctx.call("vh_conv", ca), 2.0 * M * cout * cin * 9 / 1e12, "TF/s")
    # HBM-bound: the split itself
    sa = L.SplitArgs(src0=x.data_ptr(), src1=None, c0=cin, c1=0, scale0=1.0, scale1=1.0, pro=1, npix=M, c_pad=cin, out=s8.data_ptr(), out_raw=None)
    loop("vh_split 32x128x128x256", lambda: ctx.call("vh_split", sa), 8.0 * M * cin / 1e9, "GB/s")
    del x, out, s8
    # cross-view attention of the C2 128x128 level
    b, heads, S, KL, D = 8, 4, 16384, 49152, 64
    Cc = heads * D
    qkv = torch.randn(b, S, 3 * Cc, generator=g).cuda()
    kv = torch.randn(b, KL - S, 2 * Cc, generator=g).cuda()
    Q = torch.zeros(b * heads * S * D, device="cuda"); K = torch.zeros(b * heads * KL * D, device="cuda"); V = torch.zeros(b * heads * KL * D, device="cuda")
    ao = torch.empty(b, S, Cc, device="cuda")
    LOG2E = 1.4426950408889634
    ctx.call("vh_qkv_split_x3", L.QkvSplitArgs(inp=qkv.data_ptr(), rows=b, s=S, heads=heads, d=D, nj=3, rows_per_b=1, koff=0, kl=KL, qscale=LOG2E / math.sqrt(D),
                                                q=Q.data_ptr(), k=K.data_ptr(), v=V.data_ptr()))
    ctx.call("vh_qkv_split_x3", L.QkvSplitArgs(inp=kv.data_ptr(), rows=b, s=KL - S, heads=heads, d=D, nj=2, rows_per_b=1, koff=S, kl=KL, qscale=1.0, q=None,
                                                k=K.data_ptr(), v=V.data_ptr()))
    aa = L.AttentionArgs(q=Q.data_ptr(), k=K.data_ptr(), v=V.data_ptr(), b=b, heads=heads, s=S, kl=KL, d=D, n_zero_keys=0.0, out=ao.data_ptr(), out_s8=0,
                         logit_bound=LOG2E * math.sqrt(D) * 1.001)
    loop("attention b8 h4 S16384 KL49152", lambda: ctx.call("vh_attention_x3", aa), 4.0 * b * heads * S * KL * D / 1e12, "TF/s")
    del qkv, kv, Q, K, V, ao
    # whole C2 evaluations
    cfg = vivid_amd.vivid_base(256)
    net = vivid_amd.NVPrecond.from_config(cfg)
    net.load_state_dict(vivid_amd.synth_state_dict(cfg, seed=0))
    net = net.cuda()
    B, R = 16, 256
    src = (torch.rand(2 * B, 3, R, R, generator=g) * 2 - 1).cuda()
    xx = torch.randn(2 * B, 3, R, R, generator=g).cuda() * 5
    geo = torch.randn(2 * B, 20, generator=g).cuda()
    sig = torch.full((2 * B,), 5.0).cuda()
    smp.phase = "gap"
    net(src, xx, sig, geo); torch.cuda.synchronize()
    smp.phase = "C2 net evaluation (B=16)"
    t0, n = time.time(), 0
    while time.time() - t0 < a.seconds:
        net(src, xx, sig, geo); torch.cuda.synchronize(); n += 1
    results["C2 net evaluation (B=16)"] = (n / (time.time() - t0), "evals/s")
    smp.phase = "gap"
    smp.stop = True
    time.sleep(0.3)
    phases = []
    for r in smp.rows:
        if r["phase"] not in phases and r["phase"] not in ("gap", "init"):
            phases.append(r["phase"])
    cap = max((r["cap_w"] for r in smp.rows), default=0.0)
    print(f"power cap reported: {cap:.0f} W;  {len(smp.rows)} samples")
    print(f"{'phase':38s} {'n':>4s} {'P median W':>11s} {'P max W':>8s} {'sclk median MHz':>16s} {'sclk min':>9s} {'mclk':>6s} {'temp C':>7s}   rate")
    for ph in phases:
        rs = [r for r in smp.rows if r["phase"] == ph]
        rs = rs[len(rs) // 5:] if len(rs) > 10 else rs       # drop the ramp
        rate = results.get(ph)
        print(f"{ph:38s} {len(rs):4d} {statistics.median(r['power_w'] for r in rs):11.0f} {max(r['power_w'] for r in rs):8.0f} "
              f"{statistics.median(r['sclk_mhz'] for r in rs):16.0f} {min(r['sclk_mhz'] for r in rs):9.0f} {statistics.median(r['mclk_mhz'] for r in rs):6.0f} "
              f"{statistics.median(r['temp_c'] for r in rs):7.0f}   " + (f"{rate[0]:.1f} {rate[1]}" if rate else ""))


if __name__ == "__main__":
    main()
