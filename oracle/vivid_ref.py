"""CPU ORACLE — test infrastructure, NOT product code.

A plain-PyTorch (fp32, CPU) restatement of the reference's denoiser hot path
(`training/models.py`, `generate_images.py:43-134`, `training/utils.py:84-94,
142-148,189-216`, `training/encoders.py:58-62`).  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module; `vivid_amd/` never does.

Pinning: the restatement is checked against outputs of the reference itself,
imported on CPU in the build container (`tests/golden/make_fixtures.py` →
`tests/golden/*.npz`; `tests/test_oracle_golden.py`).  The reference has no
tests or golden vectors of its own, and the contractions it
calls live in torch (conv2d / matmul / scaled_dot_product_attention), so the
fixtures were made with the torch 2.10.0 CPU kernels of this image.

Design: functional.  A network is (config dict, state_dict); nothing here
subclasses `torch.nn.Module`.  Every function cites the reference lines it
follows.  Activations are NCHW like the reference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# --------------------------------------------------------------------------
# Magnitude-preserving primitives


def normalize(x: Tensor, dim=None, eps: float = 1e-4) -> Tensor:
    """training/models.py:37-42 — x / (eps + ||x||_dim * sqrt(norm.numel()/x.numel()))."""
    if dim is None:
        dim = list(range(1, x.ndim))
    n = torch.linalg.vector_norm(x, dim=dim, keepdim=True, dtype=torch.float32)
    n = eps + n * math.sqrt(n.numel() / x.numel())
    return x / n


def resample(x: Tensor, mode: str, f=(1, 1)) -> Tensor:
    """training/models.py:48-61.  With the default f=[1,1]: 'down' is a depthwise stride-2 conv with a constant
    0.25 filter (= 2x2 mean), 'up' a depthwise transposed conv with a ones filter (= 2x nearest replicate).
    Any other even-length filter: g = outer(f, f) / sum(f)^2; 'down' = depthwise stride-2 correlation with g,
    padding (len-1)//2; 'up' = depthwise stride-2 transposed convolution with 4 g, same padding."""
    if mode == "keep":
        return x
    if tuple(float(v) for v in f) == (1.0, 1.0):
        if mode == "down":
            return F.avg_pool2d(x, 2)
        assert mode == "up"
        return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    f1 = torch.tensor([float(v) for v in f], dtype=torch.float32)
    assert f1.ndim == 1 and len(f1) % 2 == 0
    pad = (len(f1) - 1) // 2
    f1 = f1 / f1.sum()
    c = x.shape[1]
    g = torch.outer(f1, f1)[None, None].to(x.dtype).tile([c, 1, 1, 1])
    if mode == "down":
        return F.conv2d(x, g, groups=c, stride=2, padding=pad)
    assert mode == "up"
    return F.conv_transpose2d(x, g * 4, groups=c, stride=2, padding=pad)


def mp_silu(x: Tensor) -> Tensor:
    """training/models.py:66-67."""
    return F.silu(x) / 0.596


def mp_sum(a: Tensor, b: Tensor, t: float = 0.5) -> Tensor:
    """training/models.py:72-73."""
    return a.lerp(b, t) / math.sqrt((1 - t) ** 2 + t ** 2)


def mp_cat(a: Tensor, b: Tensor, t: float = 0.5) -> Tensor:
    """training/models.py:78-84 (dim=1)."""
    Na, Nb = a.shape[1], b.shape[1]
    C = math.sqrt((Na + Nb) / ((1 - t) ** 2 + t ** 2))
    return torch.cat([C / math.sqrt(Na) * (1 - t) * a, C / math.sqrt(Nb) * t * b], dim=1)


def mp_fourier(x: Tensor, freqs: Tensor, phases: Tensor) -> Tensor:
    """training/models.py:96-101."""
    y = x.to(torch.float32).ger(freqs.to(torch.float32)) + phases.to(torch.float32)
    return (y.cos() * math.sqrt(2)).to(x.dtype)


def mp_weight(w: Tensor, gain=1.0) -> Tensor:
    """Weight part of MPConv.forward, training/models.py:115-120: per-output-channel
    normalisation then gain/sqrt(fan_in)."""
    w = normalize(w.to(torch.float32))
    return w * (gain / math.sqrt(w[0].numel()))


def mp_conv(x: Tensor, w: Tensor, gain=1.0) -> Tensor:
    """training/models.py:114-126."""
    w = mp_weight(w, gain).to(x.dtype)
    if w.ndim == 2:
        return x @ w.t()
    return F.conv2d(x, w, padding=w.shape[-1] // 2)


# --------------------------------------------------------------------------
# Architecture walk (restates UNet.__init__ :339-384, XAttnUNet.__init__ :430-480,
# UNetEncoder.__init__ :526-534, SRXAttnUNet.__init__ :577-582)

DEFAULTS = dict(
    img_channels=3, source_label_dim=20, target_label_dim=40, model_channels=128,
    channel_mult=(1, 2, 3, 4), num_blocks=3, attn_resolutions=(16, 8), extra_attn=None,
    label_balance=0.5, concat_balance=0.5, res_balance=0.3, attn_balance=0.3, clip_act=256.0,
    sigma_data=0.5, logvar_channels=128, super_res=False, no_time_enc=None, depth_input=False,
    warp_depth_coor=False, uncond=None, noisy_sr=0.25,
    channel_mult_noise=None, channel_mult_emb=None, resample_filter=(1, 1),
)


def make_config(**kw) -> dict:
    cfg = dict(DEFAULTS)
    for k, v in kw.items():
        if k == "use_fp16":
            continue
        if k not in cfg and k != "img_resolution":
            raise TypeError(f"unknown config key {k}")
        cfg[k] = v
    assert "img_resolution" in cfg
    return cfg


def _walk(cfg: dict, role: str):
    """Yields (group, name, info) for every entry of enc/dec in construction order."""
    R = cfg["img_resolution"]
    warp = cfg["logvar_channels"] * int(bool(cfg["warp_depth_coor"]))
    if role == "encoder":
        cph = 64
        cin0 = cfg["img_channels"] + int(bool(cfg["depth_input"])) + warp + 1
    else:
        cph = 32 if cfg["super_res"] else 64
        cin0 = cfg["img_channels"] + warp + 1
        if cfg["super_res"]:
            cin0 = 2 * (cin0 - 1) + 1
    cblock = [cfg["model_channels"] * m for m in cfg["channel_mult"]]
    nb, ea, ar = cfg["num_blocks"], cfg["extra_attn"], cfg["attn_resolutions"]
    entries = []
    cout = cin0
    for level, ch in enumerate(cblock):
        res = R >> level
        if level == 0:
            entries.append(("enc", f"{res}x{res}_conv", dict(kind="conv", cin=cout, cout=ch)))
            cout = ch
        else:
            entries.append(("enc", f"{res}x{res}_down", dict(kind="block", cin=cout, cout=cout, flavor="enc", resample="down", attn=False)))
        for idx in range(nb):
            cin, cout = cout, ch
            attn = res in ar or (ea is not None and ea == idx and level != 0)
            entries.append(("enc", f"{res}x{res}_block{idx}", dict(kind="block", cin=cin, cout=cout, flavor="enc", resample="keep", attn=attn)))
    skips = [e[2]["cout"] for e in entries]
    for level, ch in reversed(list(enumerate(cblock))):
        res = R >> level
        if level == len(cblock) - 1:
            entries.append(("dec", f"{res}x{res}_in0", dict(kind="block", cin=cout, cout=cout, flavor="dec", resample="keep", attn=True)))
            entries.append(("dec", f"{res}x{res}_in1", dict(kind="block", cin=cout, cout=cout, flavor="dec", resample="keep", attn=False)))
        else:
            entries.append(("dec", f"{res}x{res}_up", dict(kind="block", cin=cout, cout=cout, flavor="dec", resample="up", attn=False)))
        for idx in range(nb + 1):
            cin, cout = cout + skips.pop(), ch
            attn = res in ar or (ea is not None and ea == nb - idx and level != 0)
            entries.append(("dec", f"{res}x{res}_block{idx}", dict(kind="block", cin=cin, cout=cout, flavor="dec", resample="keep", attn=attn, cat=True)))
    for _, _, info in entries:
        if info["kind"] == "block":
            info["heads"] = info["cout"] // cph if info["attn"] else 0
            info["xattn"] = role == "unet" and info["attn"]
    if role == "encoder":
        for g, n, info in reversed(entries):
            if g != "dec":
                break
            if info["heads"] == 0:
                info["dead"] = True
            else:
                break
    return entries, cblock


# --------------------------------------------------------------------------
# Blocks


def _attention(q: Tensor, k: Tensor, v: Tensor, explicit: bool) -> Tensor:
    """q [B,h,S,D], k/v [B,h,K,D] -> [B,h,S,D]; softmax(q k^T / sqrt(D)) v
    (training/models.py:198,305 SDPA defaults; snapshot experiments/code/training/models.py:190-191)."""
    if not explicit:
        return F.scaled_dot_product_attention(q, k, v)
    w = (q @ k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    return w.softmax(dim=-1) @ v


def block_forward(sd: Dict[str, Tensor], p: str, info: dict, cfg: dict, x: Tensor, emb: Tensor,
                  f1: Optional[Tensor] = None, f2: Optional[Tensor] = None, explicit_attn: bool = False,
                  taps: Optional[dict] = None) -> Tensor:
    """Block.forward training/models.py:165-206 / XAttnBlock.forward :251-315."""
    x = resample(x, info["resample"], cfg.get("resample_filter", (1, 1)))
    has_skip = info["cin"] != info["cout"]
    if info["flavor"] == "enc":
        if has_skip:
            x = mp_conv(x, sd[p + "conv_skip.weight"])
        x = normalize(x, dim=1)
    y = mp_conv(mp_silu(x), sd[p + "conv_res0.weight"])
    c = mp_conv(emb, sd[p + "emb_linear.weight"], gain=sd[p + "emb_gain"]) + 1
    y = mp_silu(y * c.unsqueeze(2).unsqueeze(3).to(y.dtype))
    y = mp_conv(y, sd[p + "conv_res1.weight"])
    if info["flavor"] == "dec" and has_skip:
        x = mp_conv(x, sd[p + "conv_skip.weight"])
    x = mp_sum(x, y, t=cfg["res_balance"])
    if taps is not None:
        taps[p + "res"] = x
    h = info["heads"]
    if h:
        B, C, H, W = x.shape
        S = H * W
        D = C // h
        qkv = normalize(mp_conv(x, sd[p + "attn_qkv.weight"]).view(B, h, D, 3, S), dim=2)
        q, k, v = qkv.unbind(3)
        if info["xattn"]:
            ks, vs = [k], [v]
            for f in (f1, f2):
                if f is None:
                    continue
                Sc = f.shape[2] * f.shape[3]
                kv = normalize(mp_conv(f, sd[p + "x_attn_kv.weight"]).view(B, h, D, 2, Sc), dim=2)
                kc, vc = kv.unbind(3)
                ks.append(kc)
                vs.append(vc)
            k, v = torch.cat(ks, dim=3), torch.cat(vs, dim=3)
        y = _attention(q.transpose(-1, -2), k.transpose(-1, -2), v.transpose(-1, -2), explicit_attn)
        y = y.transpose(-1, -2).reshape(B, C, H, W)
        y = mp_conv(y, sd[p + "attn_proj.weight"])
        x = mp_sum(x, y, t=cfg["attn_balance"])
    if cfg["clip_act"] is not None:
        x = x.clip(-cfg["clip_act"], cfg["clip_act"])
    return x


def _embedding(sd, p, cfg, noise_labels: Tensor, geometry: Optional[Tensor]) -> Tensor:
    """UNet.forward :388-391 / XAttnUNet.forward :485-488."""
    emb = mp_conv(mp_fourier(noise_labels, sd[p + "emb_fourier.freqs"], sd[p + "emb_fourier.phases"]), sd[p + "emb_noise.weight"])
    if (p + "emb_label.weight") in sd and geometry is not None:
        emb = mp_sum(emb, mp_conv(geometry, sd[p + "emb_label.weight"]), t=cfg["label_balance"])
    return mp_silu(emb)


def encoder_forward(sd, cfg, x: Tensor, noise_labels: Tensor, geometry: Tensor, *, prefix="encoder.",
                    explicit_attn=False, taps=None) -> List[Tensor]:
    """UNetEncoder.forward training/models.py:536-570 — list of the outputs of every attention block."""
    entries, _ = _walk(cfg, "encoder")
    emb = _embedding(sd, prefix, cfg, noise_labels, geometry)
    x = torch.cat([x, torch.ones_like(x[:, :1])], dim=1)
    skips, feats = [], []
    for g, name, info in entries:
        if info.get("dead"):
            break
        p = f"{prefix}{g}.{name}."
        if info["kind"] == "conv":
            x = mp_conv(x, sd[p + "weight"])
        else:
            if info.get("cat"):
                x = mp_cat(x, skips.pop(), t=cfg["concat_balance"])
            x = block_forward(sd, p, info, cfg, x, emb, explicit_attn=explicit_attn, taps=taps)
            if info["heads"] > 0:
                feats.append(x)
        if g == "enc":
            skips.append(x)
        if taps is not None:
            taps[p + "out"] = x
    return feats


def xunet_forward(sd, cfg, x: Tensor, features: Sequence[Tensor], noise_labels: Tensor, geometry: Optional[Tensor], *,
                  prefix="unet.", dual: bool = True, explicit_attn=False, taps=None) -> Tensor:
    """XAttnUNet.forward training/models.py:483-518 (dual=True: HEAD, features de-interleaved
    into two sources :491-492; dual=False: snapshot single-source,
    experiments/code/training/models.py:455-483)."""
    entries, _ = _walk(cfg, "unet")
    emb = _embedding(sd, prefix, cfg, noise_labels, geometry)
    if dual:
        fa = [f[0::2] for f in features]
        fb = [f[1::2] for f in features]
    else:
        fa = list(features)
        fb = [None] * len(fa)
    x = torch.cat([x, torch.ones_like(x[:, :1])], dim=1)
    skips = []
    for g, name, info in entries:
        p = f"{prefix}{g}.{name}."
        if info["kind"] == "conv":
            x = mp_conv(x, sd[p + "weight"])
        else:
            if info.get("cat"):
                x = mp_cat(x, skips.pop(), t=cfg["concat_balance"])
            if info["xattn"]:
                x = block_forward(sd, p, info, cfg, x, emb, fa.pop(0), fb.pop(0), explicit_attn=explicit_attn, taps=taps)
            else:
                x = block_forward(sd, p, info, cfg, x, emb, explicit_attn=explicit_attn, taps=taps)
        if g == "enc":
            skips.append(x)
        if taps is not None:
            taps[p + "out"] = x
    return mp_conv(x, sd[prefix + "out_conv.weight"], gain=sd[prefix + "out_gain"])


def zero_features(cfg, n_rows: int, dtype=torch.float32) -> List[Tensor]:
    """The zero feature list of the unconditional branch, training/models.py:727-736."""
    entries, _ = _walk(cfg, "unet")
    R = cfg["img_resolution"]
    out = []
    for g, name, info in entries:
        if info["kind"] == "block" and info["xattn"]:
            res = int(name.split("x")[0])
            out.append(torch.zeros(n_rows, info["cout"], res, res, dtype=dtype))
    return out


# --------------------------------------------------------------------------
# Geometry / depth-warp features (config 5)

_MEAN = torch.tensor([9.6681e-01, -1.6038e-04, -3.7034e-05, -1.6904e-03, -8.7718e-05,
                      9.9869e-01, 3.1288e-03, -1.0794e-03, 1.0653e-05, 3.0997e-03,
                      9.6691e-01, 1.2561e-02, 5.7708e+01, 5.7704e+01, 3.2000e+01,
                      3.2000e+01, 5.7708e+01, 5.7704e+01, 3.2000e+01, 3.2000e+01])
_STD = torch.tensor([0.1104, 0.0346, 0.2279, 0.4930, 0.0347, 0.0091, 0.0367, 0.2208, 0.2279,
                     0.0368, 0.1088, 1.0751, 6.6464, 6.6511, 0.0000, 0.0000, 6.6464, 6.6511,
                     0.0000, 0.0000])


def _stats(imsize, like: Tensor):
    """training/utils.py:69,77-78 / :89-91."""
    mean, std = _MEAN.clone().to(like), _STD.clone().to(like)
    mean[12:] *= imsize / 64
    std[12:] *= (imsize / 64) ** 2
    return mean, std


def compose_geometry(tgt2src: Tensor, src_K4: Tensor, tgt_K4: Tensor, imsize=64) -> Tensor:
    """training/utils.py:64-81 (intrinsics already as (fx,fy,cx,cy) 4-vectors)."""
    mean, std = _stats(imsize, tgt2src)
    g = torch.cat((tgt2src.reshape(*tgt2src.shape[:-2], 12), src_K4, tgt_K4), -1)
    return torch.where(std > 0, (g - mean) / std, torch.zeros_like(g))


def _K3(t: Tensor) -> Tensor:
    """training/utils.py:54-61."""
    K = torch.zeros(t.shape[:-1] + (3, 3), dtype=t.dtype)
    K[..., 0, 0], K[..., 1, 1], K[..., 0, 2], K[..., 1, 2] = t.unbind(-1)
    K[..., 2, 2] = 1
    return K


def decompose_geometry(t: Tensor, imsize=64):
    """training/utils.py:84-94."""
    mean, std = _stats(imsize, t)
    t = t * std + mean
    return t[..., :12].reshape(*t.shape[:-1], 3, 4), _K3(t[..., 12:16]), _K3(t[..., 16:])


def warp_grid(depth_bhwc: Tensor, geometry: Tensor, grid: Tensor) -> Tensor:
    """training/utils.py:189-201: unproject source pixel centres with depth, move to
    the target frame with the inverse of tgt2src, project with K_tgt, divide, NaN->0."""
    tgt2src, Ks, Kt = decompose_geometry(geometry[:, None], imsize=grid.shape[-2])
    p = torch.cat([grid, torch.ones_like(grid[..., :1])], -1)
    w = p @ torch.inverse(Ks).transpose(-1, -2)
    w = torch.cat([w * depth_bhwc, torch.ones_like(depth_bhwc)], dim=-1)
    bottom = torch.tensor([0, 0, 0, 1], dtype=tgt2src.dtype).reshape(1, 1, 1, 4).repeat(tgt2src.shape[:-2] + (1, 1))
    E = torch.cat([tgt2src, bottom], -2)                        # :142-148
    w = w @ torch.inverse(E).transpose(-1, -2)
    w = w[..., :3] @ Kt.transpose(-1, -2)
    g = (w / w[..., 2:])[..., :2]
    return torch.where(torch.isnan(g), torch.zeros_like(g), g)


def warped_features(depth: Tensor, geometry: Tensor, freqs: Tensor, phases: Tensor):
    """training/utils.py:204-216 — (features of the pixel grid, features of the warped grid),
    each [N,128,H,W]; channel = 64*axis + fourier index, axis 0 = row coordinate (meshgrid 'ij')."""
    N, _, _, S = depth.shape
    ar = torch.arange(0, S, dtype=depth.dtype)
    ii, jj = torch.meshgrid(ar, ar, indexing="ij")
    grid = torch.stack([ii, jj], dim=-1)[None].repeat(N, 1, 1, 1) + 0.5       # [N,H,W,2]
    wg = warp_grid(depth.permute(0, 2, 3, 1), geometry, grid)
    def emb(g):
        e = mp_fourier(g.reshape(-1), freqs, phases)[..., :64].reshape(N, S, S, 128)
        return e.permute(0, 3, 1, 2)
    return emb(grid), emb(wg)


# --------------------------------------------------------------------------
# Denoiser


def nvprecond_forward(sd, cfg, src: Tensor, dst: Tensor, sigma: Tensor, geometry: Optional[Tensor] = None,
                      conditioning_image: Optional[Tensor] = None, *, return_logvar=False, return_features=False,
                      inject_features=None, sr_noise: Optional[Tensor] = None, dual: bool = True,
                      explicit_attn: bool = False, taps: Optional[dict] = None):
    """NVPrecond._forward_dualsource training/models.py:628-689 (dual=True) and the
    single-source forward :691-749 with the snapshot's label handling
    (experiments/code/training/models.py:584) (dual=False).

    `sr_noise` replaces `randn_like(conditioning_image)` of :658/:721 so runs are repeatable
    (None = zeros, i.e. noisy_sr disabled).  For an `uncond` net, geometry is zeroed (:631) and
    features default to zeros (:727-736; at HEAD that branch is only reachable through
    `inject_features`)."""
    x = dst.to(torch.float32)
    sig = sigma.to(torch.float32).reshape(-1, 1, 1, 1)
    n_rows = x.shape[0]
    label_dim = cfg["source_label_dim"]
    if geometry is None:
        geometry = torch.zeros(n_rows, label_dim)
    geometry = geometry.to(torch.float32) * int(not cfg["uncond"])
    sd_ = cfg["sigma_data"]
    c_skip = sd_ ** 2 / (sig ** 2 + sd_ ** 2)
    c_out = sig * sd_ / (sig ** 2 + sd_ ** 2).sqrt()
    c_in = 1 / (sd_ ** 2 + sig ** 2).sqrt()
    c_noise = sig.flatten().log() / 4
    x_in = c_in * x
    src = src.to(torch.float32)
    if cfg["warp_depth_coor"]:
        assert src.shape[1] == 4
        depth = src[:, 3:]
        if torch.all(src[:, :3] == 0):
            sg = dg = torch.zeros(src.shape[:1] + (128,) + src.shape[-2:])
        else:
            sg, dg = warped_features(depth, geometry, sd["logvar_fourier.freqs"], sd["logvar_fourier.phases"])
        src = torch.cat([src[:, :3], sg], dim=1)
        x_in = torch.cat([x_in, dg], dim=1)
    if cfg["super_res"]:
        assert conditioning_image is not None
        cond = conditioning_image.to(torch.float32)
        if sr_noise is not None:
            cond = cond + cfg["noisy_sr"] * sr_noise
        x_in = torch.cat([x_in, cond.repeat_interleave(2, dim=0) if dual else cond], dim=1)
    if inject_features is not None:
        features = list(inject_features)
    elif cfg["uncond"]:
        features = zero_features(cfg, n_rows)
    else:
        features = encoder_forward(sd, cfg, src, c_noise * int(not cfg["no_time_enc"]), geometry,
                                   explicit_attn=explicit_attn, taps=taps)
    if return_features:
        return features
    if dual:
        B = n_rows // 2
        F_x = xunet_forward(sd, cfg, x_in[::2], features, c_noise[::2], geometry.reshape(B, -1),
                            dual=True, explicit_attn=explicit_attn, taps=taps)
        D_x = c_skip[::2] * x[::2] + c_out[::2] * F_x
        cn = c_noise[::2]
    else:
        F_x = xunet_forward(sd, cfg, x_in, features, c_noise, geometry, dual=False, explicit_attn=explicit_attn, taps=taps)
        D_x = c_skip * x + c_out * F_x
        cn = c_noise
    if return_logvar:
        lv = mp_conv(mp_fourier(cn, sd["logvar_fourier.freqs"], sd["logvar_fourier.phases"]), sd["logvar_linear.weight"])
        return D_x, lv.reshape(-1, 1, 1, 1)
    return D_x


class OracleNet:
    """Callable with the reference's `net(src, dst, sigma, geometry, cond, ...)` protocol
     around :func:`nvprecond_forward`, so the oracle sampler can drive it."""

    def __init__(self, cfg: dict, sd: Dict[str, Tensor], dual: bool = True):
        self.cfg, self.sd, self.dual = cfg, {k: v.detach().to(torch.float32).cpu() for k, v in sd.items()}, dual
        self.img_resolution = cfg["img_resolution"]
        self.img_channels = cfg["img_channels"]
        self.no_time_enc = cfg["no_time_enc"]
        self.super_res = cfg["super_res"]
        self.depth_input = cfg["depth_input"]
        self.sr_noise = None

    def __call__(self, src, dst, sigma, geometry=None, conditioning_image=None, force_fp32=False,
                 return_logvar=False, return_features=False, inject_features=None):
        with torch.no_grad():
            return nvprecond_forward(self.sd, self.cfg, src, dst, sigma, geometry, conditioning_image,
                                     return_logvar=return_logvar, return_features=return_features,
                                     inject_features=inject_features, sr_noise=self.sr_noise, dual=self.dual)


# --------------------------------------------------------------------------
# Sampler, RNG stacking, pixel codec


def edm_sampler(net, src, noise, labels=None, gnet=None, conditioning_image=None,
                num_steps=32, sigma_min=0.002, sigma_max=80, rho=7, guidance=1,
                S_churn=0, S_min=0, S_max=float("inf"), S_noise=1,
                dtype=torch.float32, randn_like=torch.randn_like, trace: Optional[list] = None):
    """generate_images.py:43-118.  `trace`, if a list, receives (t, x_in, D) per denoiser call."""
    features = None
    if getattr(net, "no_time_enc", None):
        features = net(src, torch.zeros_like(src), torch.ones(src.shape[0], dtype=dtype), labels,
                       conditioning_image, return_features=True)                               # :52-53

    def denoise(x, t):
        t = t.expand(x.shape[0])
        Dx = net(src, x, t, labels, conditioning_image, inject_features=features).to(dtype)   # :57
        if guidance != 1:
            Dx = gnet(src, x, t).to(dtype).lerp(Dx, guidance)                                  # :61-62
        if trace is not None:
            trace.append((float(t[0]), x.clone(), Dx.clone()))
        return Dx

    idx = torch.arange(num_steps, dtype=dtype)
    t_steps = (sigma_max ** (1 / rho) + idx / (num_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
    t_steps = torch.cat([t_steps, torch.zeros_like(t_steps[:1])])                              # :68-70
    x_next = noise.to(dtype) * t_steps[0]
    dual = False
    for i, (t_cur, t_next) in enumerate(zip(t_steps[:-1], t_steps[1:])):
        x_cur = x_next
        if S_churn > 0 and S_min <= t_cur <= S_max:                                            # :78-84
            gamma = min(S_churn / num_steps, np.sqrt(2) - 1)
            t_hat = t_cur + gamma * t_cur
            x_hat = x_cur + (t_hat ** 2 - t_cur ** 2).sqrt() * S_noise * randn_like(x_cur)
        else:
            t_hat, x_hat = t_cur, x_cur
        D = denoise(x_hat, t_hat)
        dual = D.shape[0] != x_hat.shape[0]                                                    # :90
        xh = x_hat[::2] if dual else x_hat
        d_cur = (xh - D) / t_hat
        half = xh + (t_next - t_hat) * d_cur
        if i < num_steps - 1:                                                                  # :104-114
            x_probe = half.repeat_interleave(2, dim=0) if dual else half
            Dp = denoise(x_probe, t_next)
            d_prime = (half - Dp) / t_next
            half = xh + (t_next - t_hat) * (0.5 * d_cur + 0.5 * d_prime)
        x_next = half.repeat_interleave(2, dim=0) if dual else half                            # :96-98,110-111
    return x_next[::2] if dual else x_next


class StackedRandomGenerator:
    """generate_images.py:120-134 (CPU generators)."""

    def __init__(self, device, seeds):
        self.generators = [torch.Generator(device).manual_seed(int(s) % (1 << 32)) for s in seeds]

    def randn(self, size, **kw):
        assert size[0] == len(self.generators)
        return torch.stack([torch.randn(size[1:], generator=g, **kw) for g in self.generators])

    def randn_like(self, inp):
        return self.randn(inp.shape, dtype=inp.dtype, layout=inp.layout, device=inp.device)


def encode_latents(x_u8: Tensor) -> Tensor:
    """training/encoders.py:58-59."""
    return x_u8.to(torch.float32) / 127.5 - 1


def decode_latents(x: Tensor) -> Tensor:
    """training/encoders.py:61-62."""
    return (x.to(torch.float32) * 127.5 + 128).clip(0, 255).to(torch.uint8)


def rank_batches(n_seeds: int, max_batch: int, world: int, rank: int):
    """generate_images.py:199-200 — which seed indices each rank processes."""
    num_batches = max((n_seeds - 1) // (max_batch * world) + 1, 1) * world
    return np.array_split(np.arange(n_seeds), num_batches)[rank::world]
