#!/usr/bin/env python3
"""In-kernel shader clock of the two MFMA-bound kernels under sustained load (MI355X_MICROARCH.md, DVFS give-back item 6).

Needs the diagnostic builds:   make -C vivid_amd/csrc variant NAME=clkc SRC=conv_x3 DEFS=-DVH_CLOCK=1
                               make -C vivid_amd/csrc variant NAME=clka SRC=attention_x3 DEFS=-DVH_CLOCK=1
  python tools/clock_probe.py conv ROWS H W CIN COUT      |     python tools/clock_probe.py attn B HEADS S KL
Launches the op back to back for SECONDS (default 3) on random data, then reads, per workgroup of the last launch, shader cycles
(s_memtime) and 100 MHz ticks (s_memrealtime) spent in the main loop: clock = cycles / ticks * 100 MHz; prints the median."""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from vivid_amd import _lib as L  # noqa: E402

kind = sys.argv[1]
args = [int(x) for x in sys.argv[2:]]
L.LIB_PATH = os.path.join(ROOT, "vivid_amd", os.environ.get("CLOCK_LIB", "libvivid_hip_clkc.so" if kind == "conv" else "libvivid_hip_clka.so"))
ctx = L.Context(torch.cuda.current_stream().cuda_stream)
dbg = torch.zeros(12 << 14, dtype=torch.int64, device="cuda")
L.set_knob("dbg_lo", dbg.data_ptr() & 0xFFFFFFFF if (dbg.data_ptr() & 0xFFFFFFFF) < 2 ** 31 else (dbg.data_ptr() & 0xFFFFFFFF) - 2 ** 32)
L.set_knob("dbg_hi", dbg.data_ptr() >> 32)
g = torch.Generator().manual_seed(0)
if kind == "conv":
    rows, h, w, cin, cout = args[:5]
    x = torch.randn(rows, h, w, cin, generator=g).cuda()
    wgt = torch.randn(cout, cin, 3, 3, generator=g).cuda()
    zeros = torch.zeros(16384, device="cuda")
    M, k_pad = rows * h * w, 9 * cin
    s8 = torch.empty(M * cin, device="cuda")
    ctx.call("vh_split", L.SplitArgs(src0=x.data_ptr(), src1=None, c0=cin, c1=0, scale0=1.0, scale1=1.0, pro=0, npix=M, c_pad=cin, out=s8.data_ptr(), out_raw=None))
    wt = torch.zeros(k_pad // 4 * cout * 4, device="cuda")
    ctx.call("vh_prep_weight", L.PrepWeightArgs(w=wgt.data_ptr(), cout=cout, cin=cin, taps=9, cin_pad=cin, k_pad=k_pad, gain_ptr=None, gain_value=1.0,
                                                wt=wt.data_ptr(), dst_col0=0, dst_cols=cout, split=2))
    out = torch.empty(M, cout, device="cuda")
    srcf = int(os.environ.get("SRCF32", "0"))              # 1: vh_conv_args.src_f32 - the patch staged from the fp32 tensor through registers (mp_silu + split in the kernel)
    op, a = "vh_conv", L.ConvArgs(src0=x.data_ptr() if srcf else s8.data_ptr(), src_f32=srcf, src1=None, c0=cin, c1=0, scale0=1.0, scale1=1.0, rows=rows, h=h, w=w, up=0, taps=9, pro=srcf,
                                  wt=wt.data_ptr(), cin_pad=cin, k_pad=k_pad, zeros=zeros.data_ptr(), zeros_bytes=65536, scratch=None, scratch_floats=0,
                                  cout=cout, out=out.data_ptr(), out_s8=None, out_s8_c=0, prec=1, kernel=1, epi=0, tile=int(os.environ.get('TILE', '0')))
    flops = 2.0 * M * cout * cin * 9
else:
    b, heads, S, KL = args[:4]
    D, C = 64, heads * 64
    qkv = torch.randn(b, S, 3 * C, generator=g).cuda()
    kv = torch.randn(b, max(KL - S, 1), 2 * C, generator=g).cuda()
    klp = (KL + 63) // 64 * 64
    Q = torch.zeros(b * heads * S * D, device="cuda"); K = torch.zeros(b * heads * klp * D, device="cuda"); V = torch.zeros(b * heads * klp * D, device="cuda")
    out = torch.empty(b, S, C, device="cuda")
    ctx.call("vh_qkv_split_x3", L.QkvSplitArgs(inp=qkv.data_ptr(), rows=b, s=S, heads=heads, d=D, nj=3, rows_per_b=1, koff=0, kl=KL,
                                                qscale=1.4426950408889634 / math.sqrt(D), q=Q.data_ptr(), k=K.data_ptr(), v=V.data_ptr()))
    if KL > S:
        ctx.call("vh_qkv_split_x3", L.QkvSplitArgs(inp=kv.data_ptr(), rows=b, s=KL - S, heads=heads, d=D, nj=2, rows_per_b=1, koff=S, kl=KL,
                                                    qscale=1.0, q=None, k=K.data_ptr(), v=V.data_ptr()))
    op, a = "vh_attention_x3", L.AttentionArgs(q=Q.data_ptr(), k=K.data_ptr(), v=V.data_ptr(), b=b, heads=heads, s=S, kl=KL, d=D, n_zero_keys=0.0,
                                               out=out.data_ptr(), out_s8=0, logit_bound=1.4426950408889634 * math.sqrt(D) * 1.001)
    flops = 4.0 * b * heads * S * KL * D
seconds = float(os.environ.get("SECONDS_LOAD", "3"))
ctx.call(op, a)
torch.cuda.synchronize()
t0, n = time.perf_counter(), 0
while time.perf_counter() - t0 < seconds:
    for _ in range(20):
        ctx.call(op, a)
    n += 20
    torch.cuda.synchronize()
dt = time.perf_counter() - t0
W = 12 if kind == "conv" else 2
d = dbg.cpu().numpy().reshape(-1, W)
d = d[d[:, 1] > 0].astype(np.float64)
ghz = d[:, 0] / d[:, 1] * 0.1
print(f"{kind} {args}: {n} launches in {dt:.2f} s = {dt / n * 1e3:.4f} ms each, {flops * n / dt / 1e12:.1f} TF/s algorithmic; "
      f"in-kernel clock over {len(d)} workgroups: median {np.median(ghz):.3f} GHz (p10 {np.percentile(ghz, 10):.3f}, p90 {np.percentile(ghz, 90):.3f}); "
      f"loop cycles per workgroup median {np.median(d[:, 0]):.0f}")
if kind == "conv":
    clk = np.median(ghz)
    wg_per_cu = max(1.0, len(d) / 256.0)
    per_tile_us = dt / n / wg_per_cu * 1e6
    print(f"  per workgroup: prologue {np.median(d[:, 2]) / clk / 1e3:.2f} us, K loop {np.median(d[:, 0]) / clk / 1e3:.2f} us, epilogue issue {np.median(d[:, 3]) / clk / 1e3:.2f} us; "
          f"launch time / workgroups per CU = {per_tile_us:.2f} us -> unaccounted (store drain, dispatch) {per_tile_us - (np.median(d[:, 0]) + np.median(d[:, 2]) + np.median(d[:, 3])) / clk / 1e3:.2f} us")
    print("  epilogue, wave 0: cumulative us after each of its 32x32 blocks:", " ".join(f"{np.median(d[:, 4 + b]) / clk / 1e3:.2f}" for b in range(6)))
    print(f"  prologue, wave 0: index math {np.median(d[:, 10]) / clk / 1e3:.2f} us, first DMA issue + landing {np.median(d[:, 11]) / clk / 1e3:.2f} us")
    if os.environ.get("TILE") == "8" and os.environ.get("PLACEMENT"):
        # which workgroup ids share a CU, and how far apart their K loops start (100 MHz ticks): HW_ID / XCC_ID logged by the -DVH_CLOCK build
        raw = dbg.cpu().numpy().reshape(-1, W)
        place = {}
        for wg in range(min(len(raw), 2048)):
            hw, xcc = int(raw[wg, 6]), int(raw[wg, 7]) & 0xF
            if raw[wg, 1] <= 0:
                continue
            key = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)
            place.setdefault(key, []).append((int(raw[wg, 8]), int(raw[wg, 9]), wg))
        print(f"  placement: {len(place)} distinct (xcc, se, sh, cu) among the first 2048 workgroups")
        for key in sorted(place)[:6]:
            v = sorted(place[key])
            t0 = v[0][0]
            print("   ", key, " ".join(f"wg{wg}:[{a - t0},{b - t0}]" for a, b, wg in v[:8]))
    if os.environ.get("TILE") == "8":      # conv_x3_patch (CLOCK_LIB = a -DVH_CLOCK variant of conv_patch.hip): slots 4 / 5 hold sums over the K loop
        print(f"  patch kernel, wave 0: chunk-boundary patch reloads {np.median(d[:, 4]) / clk / 1e3:.2f} us, per-K-tile wait + barrier {np.median(d[:, 5]) / clk / 1e3:.2f} us "
              f"(both inside the K loop's {np.median(d[:, 0]) / clk / 1e3:.2f} us)")
