"""Accuracy of a 2304-deep GEMM evaluated as bf16x3 (hi*hi + hi*lo + lo*hi) against two int8 slices per operand (per-row 15-bit fixed point,\n3 or 4 slice products, exact integer accumulation), relative L2 error against fp64.  numpy only; see DESIGN.md section 9."""
import numpy as np
rng = np.random.default_rng(0)
def bf16(x):
    x = x.astype(np.float32); u = x.view(np.uint32)
    r = ((u >> 16) & 1) + 0x7fff
    return ((u + r) & 0xffff0000).view(np.float32)
def split_bf16(x):
    h = bf16(x); l = bf16(x - h); return h.astype(np.float64), l.astype(np.float64)
def split_i8(x, axis):
    # per-row (axis) scale to 15-bit signed fixed point; two balanced int8 slices: q = 256*q1 + q0, q0 in [-128,127]
    s = np.max(np.abs(x), axis=axis, keepdims=True) / 32512.0
    q = np.rint(x / s).astype(np.int64)
    q0 = ((q + 128) & 255) - 128
    q1 = (q - q0) >> 8
    return q1.astype(np.float64), q0.astype(np.float64), s
M, K, N = 2048, 2304, 256
for name, act in (("silu(normalised gaussian)", lambda z: z / (1 + np.exp(-z)) / 0.596), ("gaussian", lambda z: z), ("heavy tail (t3)", None)):
    if act is None:
        x = rng.standard_t(3, size=(M, K))
    else:
        x = act(rng.standard_normal((M, K)))
    w = rng.standard_normal((N, K)); w /= np.sqrt((w * w).sum(1, keepdims=True)) / 1.0
    ref = x @ w.T
    xh, xl = split_bf16(x); wh, wl = split_bf16(w)
    y3 = xh @ wh.T + xh @ wl.T + xl @ wh.T
    x1, x0, sx = split_i8(x, 1); w1, w0, sw = split_i8(w, 1)
    yi = (65536.0 * (x1 @ w1.T) + 256.0 * (x1 @ w0.T + x0 @ w1.T)) * sx * sw.T
    yi4 = yi + (x0 @ w0.T) * sx * sw.T
    e = lambda y: np.linalg.norm(y - ref) / np.linalg.norm(ref)
    print(f"{name:28s} bf16x3 {e(y3):.2e}   int8 2-slice, 3 products {e(yi):.2e}   (4 products {e(yi4):.2e})   fp32 accumulate of exact products ~{np.finfo(np.float32).eps * np.sqrt(K) / 4:.1e}")
