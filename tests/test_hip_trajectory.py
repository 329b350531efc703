"""GPU: a FULL-LENGTH guided trajectory of the reference's real presets against the CPU oracle.

The golden sampler fixtures are 2-4 steps on 64-channel toy nets; the reference's default run is 32 steps = 63 guided denoiser
evaluations (generate_images.py:45, 74, 104), and in bf16x3 mode every product carries ~2^-16 of rounding through ~100 layers per
call.  This file runs the reference-true cascade at its real widths:

  stage 1  vivid-base @64 + vivid-uncond @64 (train_nvs.py:28-29), batch 1, edm_sampler(num_steps=32, guidance=1.5): 63 calls of
           each net, HIP (bf16x3 and fp32) vs oracle.edm_sampler on the same noise;
  stage 2  vivid-sr @256 (train_nvs.py:30; noisy_sr = 0 so that both sides see the same conditioning), 16 steps = 31 calls, on the
           stage-1 output resized as generate_images.py:322 does (net = gnet = sr_model, guidance 1, :324-326).

Gates: rel-L2 of the final latents <= 1e-3 (north_star's tolerance), decoded uint8 images within +-1 LSB.  `trajectory_errors`
also returns the error after every denoiser call (the HIP sampler's x against the oracle's x at the same call), which
tools/trajectory_report.py prints for DESIGN.md 4.
"""
import pytest
import torch

from oracle import vivid_ref as R
from tests.conftest import rel_l2

pytestmark = pytest.mark.gpu


class _Recorder:
    """Wraps a HIP net with the reference's call protocol and keeps every x it was called with."""

    def __init__(self, net):
        self.net, self.xs = net, []
        self.no_time_enc = net.no_time_enc
        self.img_resolution, self.img_channels = net.img_resolution, net.img_channels

    def __call__(self, src, x, t, *a, **kw):
        self.xs.append(x[::2].clone())
        return self.net(src, x, t, *a, **kw)


def _ocfg(cfg):
    d = cfg.to_dict()
    d.pop("use_fp16")
    return R.make_config(**d)


def _hip(cfg, sd, precision):
    import vivid_amd
    net = vivid_amd.NVPrecond.from_config(cfg, precision=precision)
    net.load_state_dict(sd, strict=True)
    return net.cuda()


def base_stage_inputs(seed=31):
    g = torch.Generator().manual_seed(seed)
    src = (torch.rand(2, 3, 64, 64, generator=g) * 2 - 1)
    geo = torch.randn(2, 20, generator=g)
    geo[:, [14, 15, 18, 19]] = 0
    noise = R.StackedRandomGenerator("cpu", [16]).randn([1, 3, 64, 64]).repeat_interleave(2, dim=0)
    return src, geo, noise


def trajectory_errors(precisions=("bf16x3", "fp32"), num_steps=32, sr_steps=16):
    """Returns {precision: dict(base_final, base_per_call, sr_final, sr_per_call, u8_base, u8_sr)} (errors are rel-L2 vs the oracle)."""
    import vivid_amd
    from vivid_amd.generate import resize
    bcfg, ucfg = vivid_amd.vivid_base(64), vivid_amd.vivid_uncond(64)
    scfg = vivid_amd.vivid_sr(256, noisy_sr=0.0)
    bsd, usd, ssd = (vivid_amd.synth_state_dict(c, seed=s) for c, s in ((bcfg, 0), (ucfg, 1), (scfg, 2)))
    src, geo, noise = base_stage_inputs()
    g = torch.Generator().manual_seed(32)
    sr_src = torch.rand(2, 3, 256, 256, generator=g) * 2 - 1
    sr_noise = R.StackedRandomGenerator("cpu", [16]).randn([1, 3, 256, 256]).repeat_interleave(2, dim=0)

    # oracle cascade
    obase, ounc, osr = R.OracleNet(_ocfg(bcfg), bsd), R.OracleNet(_ocfg(ucfg), usd), R.OracleNet(_ocfg(scfg), ssd)
    tr_b, tr_s = [], []
    lat = R.edm_sampler(obase, src, noise, labels=geo, gnet=ounc, num_steps=num_steps, guidance=1.5, trace=tr_b)
    low = torch.nn.functional.interpolate(lat, size=(256, 256), mode="bilinear", align_corners=False, antialias=True)
    sr_lat = R.edm_sampler(osr, sr_src, sr_noise, labels=geo, gnet=osr, conditioning_image=low, num_steps=sr_steps, trace=tr_s)
    assert len(tr_b) == 2 * num_steps - 1 and len(tr_s) == 2 * sr_steps - 1

    out = {}
    for prec in precisions:
        net, gnet, sr = _hip(bcfg, bsd, prec), _hip(ucfg, usd, prec), _hip(scfg, ssd, prec)
        rec = _Recorder(net)
        hlat = vivid_amd.edm_sampler(rec, src.cuda(), noise.cuda(), labels=geo.cuda(), gnet=gnet, num_steps=num_steps, guidance=1.5)
        hlow = resize(hlat, 256)
        rec_s = _Recorder(sr)
        hsr = vivid_amd.edm_sampler(rec_s, sr_src.cuda(), sr_noise.cuda(), labels=geo.cuda(), gnet=sr, conditioning_image=hlow,
                                    num_steps=sr_steps)
        assert len(rec.xs) == len(tr_b) and len(rec_s.xs) == len(tr_s)
        out[prec] = dict(
            base_final=rel_l2(hlat.cpu(), lat), sr_final=rel_l2(hsr.cpu(), sr_lat),
            base_per_call=[rel_l2(h.cpu(), o[1][::2]) for h, o in zip(rec.xs, tr_b)],
            sr_per_call=[rel_l2(h.cpu(), o[1][::2]) for h, o in zip(rec_s.xs, tr_s)],
            sigmas=[o[0] for o in tr_b],
            u8_base=int((R.decode_latents(hlat.cpu()).int() - R.decode_latents(lat).int()).abs().max()),
            u8_sr=int((R.decode_latents(hsr.cpu()).int() - R.decode_latents(sr_lat).int()).abs().max()),
            u8_sr_frac=float(((R.decode_latents(hsr.cpu()).int() - R.decode_latents(sr_lat).int()).abs() > 0).float().mean()))
        del net, gnet, sr, rec, rec_s
        torch.cuda.empty_cache()
    return out


def test_full_length_guided_cascade_vs_oracle():
    res = trajectory_errors()
    for prec, r in res.items():
        msg = f"{prec}: base {r['base_final']:.2e} (max over calls {max(r['base_per_call']):.2e}), sr {r['sr_final']:.2e}"
        print(msg)
        assert r["base_final"] < 1e-3, msg
        assert r["sr_final"] < 1e-3, msg
        assert max(r["base_per_call"]) < 1e-3 and max(r["sr_per_call"]) < 1e-3, msg
        assert r["u8_base"] <= 1 and r["u8_sr"] <= 1, (prec, r["u8_base"], r["u8_sr"])
