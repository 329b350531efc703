"""Architecture tables for the VIVID denoiser (pure data, no tensors).

The reference builds its networks imperatively inside ``torch.nn.Module``
constructors (``training/models.py:322-384`` UNet, ``:413-480`` XAttnUNet,
``:524-534`` UNetEncoder trimming, ``:576-582`` SRXAttnUNet, ``:591-624``
NVPrecond).  Here the same generator is restated as a function that returns a
flat table of block descriptors; the HIP executor (``vivid_amd/engine.py``)
walks the table, and ``vivid_amd/weights.py`` derives the reference's
state_dict key names and shapes from it.
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import List, Optional, Tuple


@dataclass(frozen=True)
class NetConfig:
    """Constructor arguments of ``NVPrecond`` (training/models.py:591-606) plus
    the UNet/Block keyword arguments it forwards (``:322-337``, ``:133-146``)."""
    img_resolution: int
    img_channels: int = 3
    source_label_dim: int = 20
    target_label_dim: int = 40
    model_channels: int = 128
    channel_mult: Tuple[int, ...] = (1, 2, 3, 4)
    num_blocks: int = 3
    attn_resolutions: Tuple[int, ...] = (16, 8)
    extra_attn: Optional[int] = None
    channel_mult_noise: Optional[int] = None   # cnoise = model_channels * this (None: first level's channels, :340)
    channel_mult_emb: Optional[int] = None     # cemb = model_channels * this (None: widest level, :341)
    resample_filter: Tuple[float, ...] = (1.0, 1.0)   # Block's up/down filter (:139, :48-61); even length
    label_balance: float = 0.5
    concat_balance: float = 0.5
    res_balance: float = 0.3
    attn_balance: float = 0.3
    clip_act: Optional[float] = 256.0
    sigma_data: float = 0.5
    logvar_channels: int = 128
    super_res: bool = False
    no_time_enc: Optional[bool] = None
    depth_input: bool = False
    warp_depth_coor: bool = False
    uncond: Optional[bool] = None
    noisy_sr: float = 0.25
    use_fp16: bool = True          # accepted for API parity; this build computes in fp32

    def to_dict(self):
        return asdict(self)


# Presets named by train_nvs.py:27-31 / experiments/training_options.json.
def vivid_base(img_resolution=64, **kw) -> NetConfig:
    return NetConfig(img_resolution=img_resolution, model_channels=128, extra_attn=1, **kw)


def vivid_uncond(img_resolution=64, **kw) -> NetConfig:
    return NetConfig(img_resolution=img_resolution, model_channels=128, extra_attn=1, uncond=True, **kw)


def vivid_sr(img_resolution=256, **kw) -> NetConfig:
    return NetConfig(img_resolution=img_resolution, model_channels=64, super_res=True, **kw)


@dataclass
class BlockSpec:
    name: str                 # e.g. "32x32_block1"
    kind: str                 # "conv" (bare MPConv 3x3) | "block"
    cin: int
    cout: int
    res: int                  # OUTPUT spatial resolution of the entry
    flavor: str = "enc"       # "enc" | "dec"
    resample: str = "keep"    # "keep" | "up" | "down"
    heads: int = 0            # 0 = no attention
    xattn: bool = False       # XAttnBlock (consumes a feature pair)
    takes_skip: bool = False  # decoder "block*" entries: mp_cat with a skip first
    skip_ch: int = 0          # channels of that skip
    live: bool = True         # False = trimmed from UNetEncoder (training/models.py:530-534)

    @property
    def has_skip_conv(self):
        return self.kind == "block" and self.cin != self.cout


@dataclass
class UNetSpec:
    img_resolution: int
    in_channels: int          # channels of the first conv's input INCLUDING the ones channel
    label_dim: int
    cnoise: int
    cemb: int
    channels_per_head: int
    enc: List[BlockSpec] = field(default_factory=list)
    dec: List[BlockSpec] = field(default_factory=list)
    out_channels: int = 0     # 0 = no out_conv (encoder)
    last_ch: int = 0          # channels entering out_conv

    def live_blocks(self):
        return [b for b in self.enc + self.dec if b.live]

    def feature_blocks(self):
        """Blocks whose output is a cross-attention feature (encoder) or that
        consume one (xattn UNet), in consumption order (training/models.py:547-570, 498-515)."""
        return [b for b in self.enc + self.dec if b.live and b.heads > 0]


def unet_spec(cfg: NetConfig, *, role: str) -> UNetSpec:
    """role = "encoder" (UNetEncoder, training/models.py:524-534) or "unet"
    (XAttnUNet / SRXAttnUNet, :413-480, :576-582)."""
    assert role in ("encoder", "unet")
    R = cfg.img_resolution
    warp = cfg.logvar_channels * int(cfg.warp_depth_coor)
    if role == "encoder":
        img_ch = cfg.img_channels + int(cfg.depth_input) + warp     # :620
        label_dim = cfg.source_label_dim
        cph = 64
        in_ch = img_ch + 1
    else:
        img_ch = cfg.img_channels + warp                             # :622
        label_dim = cfg.target_label_dim
        cph = 32 if cfg.super_res else 64                            # :578
        in_ch = img_ch + 1
        if cfg.super_res:
            in_ch = 2 * (in_ch - 1) + 1                              # :581
    cblock = [cfg.model_channels * m for m in cfg.channel_mult]
    cnoise = cfg.model_channels * cfg.channel_mult_noise if cfg.channel_mult_noise is not None else cblock[0]   # :340
    cemb = cfg.model_channels * cfg.channel_mult_emb if cfg.channel_mult_emb is not None else max(cblock)       # :341
    spec = UNetSpec(img_resolution=R, in_channels=in_ch, label_dim=label_dim,
                    cnoise=cnoise, cemb=cemb, channels_per_head=cph)
    xattn = role == "unet"

    def heads_of(attn, cout):
        return cout // cph if attn else 0

    cout = in_ch
    L = len(cblock)
    for level, ch in enumerate(cblock):
        res = R >> level
        if level == 0:
            spec.enc.append(BlockSpec(f"{res}x{res}_conv", "conv", cout, ch, res))
            cout = ch
        else:
            spec.enc.append(BlockSpec(f"{res}x{res}_down", "block", cout, cout, res, "enc", "down"))
        for idx in range(cfg.num_blocks):
            cin, cout = cout, ch
            attn = res in cfg.attn_resolutions or (cfg.extra_attn is not None and cfg.extra_attn == idx and level != 0)
            h = heads_of(attn, cout)
            if attn and xattn and h == 0:
                raise ValueError("attention requested at a level with fewer channels than one head; "
                                 "the reference pops features the encoder never produced here")
            spec.enc.append(BlockSpec(f"{res}x{res}_block{idx}", "block", cin, cout, res, "enc",
                                      heads=h, xattn=xattn and attn))
    skips = [b.cout for b in spec.enc]
    for level, ch in reversed(list(enumerate(cblock))):
        res = R >> level
        if level == L - 1:
            h = heads_of(True, cout)
            spec.dec.append(BlockSpec(f"{res}x{res}_in0", "block", cout, cout, res, "dec", heads=h, xattn=xattn))
            spec.dec.append(BlockSpec(f"{res}x{res}_in1", "block", cout, cout, res, "dec"))
        else:
            spec.dec.append(BlockSpec(f"{res}x{res}_up", "block", cout, cout, res, "dec", "up"))
        for idx in range(cfg.num_blocks + 1):
            sk = skips.pop()
            cin, cout = cout + sk, ch
            attn = res in cfg.attn_resolutions or (cfg.extra_attn is not None and cfg.extra_attn == cfg.num_blocks - idx and level != 0)
            h = heads_of(attn, cout)
            spec.dec.append(BlockSpec(f"{res}x{res}_block{idx}", "block", cin, cout, res, "dec",
                                      heads=h, xattn=xattn and attn, takes_skip=True, skip_ch=sk))
    spec.last_ch = cout
    if role == "unet":
        spec.out_channels = 3                                        # :480 (hard-coded 3)
    else:
        for b in reversed(spec.dec):                                 # :530-534
            if b.heads == 0:
                b.live = False
            else:
                break
    return spec


def feature_shapes(cfg: NetConfig):
    """[(channels, res)] of the encoder feature list, in order."""
    return [(b.cout, b.res) for b in unet_spec(cfg, role="encoder").feature_blocks()]
