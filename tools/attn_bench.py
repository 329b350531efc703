#!/usr/bin/env python3
"""Time vh_attention_x3 (and fp32) on one shape: python tools/attn_bench.py b heads S KL [D]"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vivid_amd import _lib as L
b, heads, S, KL = [int(x) for x in sys.argv[1:5]]
D = int(sys.argv[5]) if len(sys.argv) > 5 else 64
ctx = L.Context(torch.cuda.current_stream().cuda_stream)
g = torch.Generator().manual_seed(0)
C = heads * D
qkv = torch.randn(b, S, 3 * C, generator=g).cuda()
kv = torch.randn(b, max(KL - S, 1), 2 * C, generator=g).cuda()
klp = (KL + 63) // 64 * 64
flops = 4.0 * b * heads * S * KL * D
res = {}
for x3 in (0, 1):
    Q = torch.zeros(b * heads * S * D, device="cuda"); K = torch.zeros(b * heads * klp * D, device="cuda"); V = torch.zeros(b * heads * klp * D, device="cuda")
    out = torch.empty(b, S, C, device="cuda")
    sp, at = ("vh_qkv_split_x3", "vh_attention_x3") if x3 else ("vh_qkv_split", "vh_attention")
    ctx.call(sp, L.QkvSplitArgs(inp=qkv.data_ptr(), rows=b, s=S, heads=heads, d=D, nj=3, rows_per_b=1, koff=0, kl=KL, qscale=1.4426950408889634 / math.sqrt(D), q=Q.data_ptr(), k=K.data_ptr(), v=V.data_ptr()))
    if KL > S:
        ctx.call(sp, L.QkvSplitArgs(inp=kv.data_ptr(), rows=b, s=KL - S, heads=heads, d=D, nj=2, rows_per_b=1, koff=S, kl=KL, qscale=1.0, q=None, k=K.data_ptr(), v=V.data_ptr()))
    a = L.AttentionArgs(q=Q.data_ptr(), k=K.data_ptr(), v=V.data_ptr(), b=b, heads=heads, s=S, kl=KL, d=D, n_zero_keys=0.0, out=out.data_ptr(), out_s8=0, logit_bound=(1.4426950408889634 * math.sqrt(D) * 1.001 if (x3 and os.environ.get('BOUND', '1') != '0') else 0.0))
    for _ in range(2): ctx.call(at, a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 3
    e0.record()
    for _ in range(n): ctx.call(at, a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    res[x3] = out
    err = float((out - res[0]).norm() / res[0].norm())
    print(f"x3={x3} {ms:8.3f} ms {flops/ms/1e9:7.1f} TF/s rel-vs-fp32 {err:.1e}")
