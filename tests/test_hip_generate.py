"""GPU: the sampling driver (vivid_amd.generate, mirror of generate_images.py:139-343) and the bilinear resize it uses."""
import os

import numpy as np
import pytest
import torch

from oracle import vivid_ref as R
from tests.conftest import rel_l2
from tests.golden.cases import CASES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hin,hout,aa", [(8, 32, False), (8, 32, True), (32, 8, True), (32, 8, False), (24, 9, True), (64, 256, True)])
def test_resize_matches_torch_interpolate(hin, hout, aa):
    from vivid_amd.generate import resize
    g = torch.Generator().manual_seed(hin * 100 + hout)
    x = torch.randn(2, 3, hin, hin, generator=g)
    ref = torch.nn.functional.interpolate(x, size=(hout, hout), mode="bilinear", align_corners=False, antialias=aa)
    got = resize(x.cuda(), hout, antialias=aa)
    assert rel_l2(got.cpu(), ref) < 1e-6


def _mk(cfg, seed):
    import vivid_amd
    net = vivid_amd.NVPrecond.from_config(cfg)
    net.load_state_dict(vivid_amd.synth_state_dict(cfg, seed=seed))
    return net.cuda()


def _data(n_rows, res, seed, const=False):
    """Collated batches in DualSourceCollate's format: rows interleaved [s1, s2, s1, s2, ...]."""
    g = torch.Generator().manual_seed(seed)

    def gen():
        while True:
            if const:
                one = torch.randint(0, 256, (1, 3, res, res), generator=torch.Generator().manual_seed(seed)).float()
                src = one.repeat(n_rows, 1, 1, 1)
                geo = torch.zeros(n_rows, 20)
            else:
                src = torch.randint(0, 256, (n_rows, 3, res, res), generator=g).float()
                geo = torch.randn(n_rows, 20, generator=g)
                geo[:, [14, 15, 18, 19]] = 0
            yield dict(src_image=src, tgt_image=src.flip(-1), geometry=geo,
                       sr_src_image=torch.nn.functional.interpolate(src, scale_factor=2), sr_tgt_image=torch.nn.functional.interpolate(src.flip(-1), scale_factor=2),
                       sr_geometry=geo)
    return gen()


def test_driver_matches_oracle_pipeline_and_is_seed_deterministic(tmp_path):
    import vivid_amd
    from vivid_amd.generate import generate_images_nvs
    case = CASES["tiny_guided"]
    cfg, gcfg = case["cfg"], case["gcfg"]
    net, gnet = _mk(cfg, 5), _mk(gcfg, 6)
    seeds = [16, 17, 18]
    kw = dict(num_steps=3, guidance=1.5)
    # (a) one batch of 3, CPU generators so that the oracle can draw the same noise
    out = list(generate_images_nvs(net, gnet, seeds=seeds, max_batch_size=4, data=_data(8, 16, 1), rng_device="cpu",
                                   outdir=str(tmp_path), **kw))
    assert len(out) == 1 and out[0].images.shape == (3, 3, 16, 16) and out[0].images.dtype == torch.uint8
    for s in seeds:
        for stem in ("src", "tgt", "sample"):
            assert os.path.exists(tmp_path / f"{stem}_{s:06d}.png")
    # oracle pipeline on the same batch
    batch = next(_data(8, 16, 1))
    src_u8, geo = batch["src_image"][::2][:3], batch["geometry"][::2][:3]
    src = R.encode_latents(src_u8).repeat_interleave(2, dim=0)
    labels = geo.repeat_interleave(2, dim=0)
    noise = R.StackedRandomGenerator("cpu", seeds).randn([3, 3, 16, 16]).repeat_interleave(2, dim=0)
    d = cfg.to_dict(); d.pop("use_fp16"); gd = gcfg.to_dict(); gd.pop("use_fp16")
    onet = R.OracleNet(R.make_config(**d), vivid_amd.synth_state_dict(cfg, seed=5))
    ognet = R.OracleNet(R.make_config(**gd), vivid_amd.synth_state_dict(gcfg, seed=6))
    ref = R.decode_latents(R.edm_sampler(onet, src, noise, labels=labels, gnet=ognet, **kw))
    diff = (out[0].images.cpu().int() - ref.int()).abs()
    assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 0.01
    # (b) a sample depends on its seed, not on how seeds are batched (constant data, device generators)
    a = list(generate_images_nvs(net, gnet, seeds=seeds, max_batch_size=4, data=_data(8, 16, 2, const=True), **kw))
    b = list(generate_images_nvs(net, gnet, seeds=seeds, max_batch_size=1, data=_data(8, 16, 2, const=True), **kw))
    assert len(b) == 3
    ia = a[0].images.cpu().int()
    ib = torch.cat([r.images for r in b]).cpu().int()
    assert int((ia - ib).abs().max()) <= 1


def _mk_sr(cfg, seed):
    """SR net with the conditioning noise switched off (the reference draws randn inside forward, training/models.py:658)."""
    import vivid_amd
    cfg0 = cfg.__class__(**{**cfg.to_dict(), "noisy_sr": 0.0})
    net = vivid_amd.NVPrecond.from_config(cfg0)
    net.load_state_dict(vivid_amd.synth_state_dict(cfg, seed=seed))
    return net.cuda()


def test_sr_cascade_matches_oracle_pipeline():
    """generate_images.py:310-327: base sampler -> bilinear (anti-aliased) resize of the LATENTS to the SR resolution -> second
    edm_sampler with gnet = sr_model and the low-res latents as conditioning image -> decode.  Oracle pipeline on the same batch
    and seeds: oracle samplers + F.interpolate(antialias=True), which is what torchvision's resize calls for tensors."""
    import vivid_amd
    from vivid_amd.generate import generate_images_nvs
    bcfg, scfg = CASES["tiny_dual"]["cfg"], CASES["tiny_sr"]["cfg"]
    base, sr = _mk(bcfg, 3), _mk_sr(scfg, 9)
    seeds = [16, 17]
    # 4 steps per stage: the last Heun correction divides by t_next = sigma_min = 0.002, so a 2-step schedule (80 -> 0.002 in one
    # step) amplifies any rounding difference in D by 0.5*80/0.002 = 2e4 - in the reference as much as here
    steps = 4
    out = list(generate_images_nvs(base, seeds=seeds, max_batch_size=2, data=_data(4, 16, 3), sr_model=sr, num_steps=steps, rng_device="cpu"))
    assert out[0].images.shape == (2, 3, 32, 32) and out[0].images.dtype == torch.uint8
    assert out[0].noise.shape == (4, 3, 32, 32) and out[0].src.shape == (2, 3, 32, 32)
    batch = next(_data(4, 16, 3))
    rep = lambda t: t.repeat_interleave(2, dim=0)      # noqa: E731
    pick = lambda k: batch[k][::2][:2]                 # noqa: E731
    d = bcfg.to_dict(); d.pop("use_fp16")
    sd_ = scfg.to_dict(); sd_.pop("use_fp16"); sd_["noisy_sr"] = 0.0
    obase = R.OracleNet(R.make_config(**d), vivid_amd.synth_state_dict(bcfg, seed=3))
    osr = R.OracleNet(R.make_config(**sd_), vivid_amd.synth_state_dict(scfg, seed=9))
    noise = rep(R.StackedRandomGenerator("cpu", seeds).randn([2, 3, 16, 16]))
    lat = R.edm_sampler(obase, rep(R.encode_latents(pick("src_image"))), noise, labels=rep(pick("geometry")), gnet=obase, num_steps=steps)
    low = torch.nn.functional.interpolate(lat, size=(32, 32), mode="bilinear", align_corners=False, antialias=True)
    sr_noise = rep(R.StackedRandomGenerator("cpu", seeds).randn([2, 3, 32, 32]))
    sr_lat = R.edm_sampler(osr, rep(R.encode_latents(pick("sr_src_image"))), sr_noise, labels=rep(pick("sr_geometry")), gnet=osr,
                           conditioning_image=low, num_steps=steps)
    ref = R.decode_latents(sr_lat)
    diff = (out[0].images.cpu().int() - ref.int()).abs()
    assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 0.01


def test_sr_handoff_downscale_matches_antialiased_interpolate():
    """The other resize of the cascade (:299-302): the SR net's conditioning image is the target shrunk 4x with the triangle
    filter widened to the scale (torchvision resize(antialias) on tensors = aten upsample_bilinear2d_aa) and blown up again."""
    from vivid_amd.generate import resize
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    f = torch.nn.functional.interpolate
    ref = f(f(x, size=(16, 16), mode="bilinear", antialias=True), size=(64, 64), mode="bilinear", antialias=True)
    got = resize(resize(x.cuda(), 16), 64)
    assert rel_l2(got.cpu(), ref) < 1e-6


def test_urls_are_refused():
    from vivid_amd.generate import generate_images_nvs
    with pytest.raises(NotImplementedError, match="needs the network"):
        generate_images_nvs("https://example.invalid/vivid-base.pkl", data=[])


def _write_snapshot_like(path, cfg, sd):
    """A file with the structure of the reference's network-snapshot-*.pkl (training_loop.py:485-496), written WITHOUT the
    reference: stand-in modules named like the reference's supply the globals the pickle stream names."""
    import pickle
    import sys
    import types

    class EasyDict(dict):
        pass

    def _reconstruct_persistent_obj(meta):       # never called here: only its qualified name goes into the stream
        raise AssertionError

    class Rec:
        def __init__(self, meta):
            self.meta = meta

        def __reduce__(self):
            return (_reconstruct_persistent_obj, (self.meta,))

    def module_tree(prefix_items):
        """nested {_parameters, _buffers, _modules} dicts from dotted names"""
        root = dict(_parameters={}, _buffers={}, _modules={}, _non_persistent_buffers_set=set())
        for name, t in prefix_items:
            node = root
            parts = name.split(".")
            for p in parts[:-1]:
                if p not in node["_modules"]:
                    node["_modules"][p] = Rec(dict(type="class", version=6, module_src="raise SystemExit('executed')", class_name="Sub",
                                                   state=dict(_parameters={}, _buffers={}, _modules={}, _non_persistent_buffers_set=set())))
                node = node["_modules"][p].meta["state"]
            node["_parameters"][parts[-1]] = torch.nn.Parameter(t.to(torch.float16), requires_grad=False)
        return root

    kw = cfg.to_dict()
    for k in ("precision",):
        kw.pop(k, None)
    state = module_tree(sd.items())
    state.update(_init_args=[], _init_kwargs=kw)
    top = Rec(dict(type="class", version=6, module_src="raise SystemExit('executed')", class_name="NVPrecond", state=state))
    fake = {"torch_utils": types.ModuleType("torch_utils"), "torch_utils.persistence": types.ModuleType("torch_utils.persistence"),
            "dnnlib": types.ModuleType("dnnlib"), "dnnlib.util": types.ModuleType("dnnlib.util")}
    fake["torch_utils.persistence"]._reconstruct_persistent_obj = _reconstruct_persistent_obj
    _reconstruct_persistent_obj.__module__, _reconstruct_persistent_obj.__qualname__ = "torch_utils.persistence", "_reconstruct_persistent_obj"
    fake["dnnlib.util"].EasyDict = EasyDict
    EasyDict.__module__, EasyDict.__qualname__ = "dnnlib.util", "EasyDict"
    saved = {k: sys.modules.get(k) for k in fake}
    sys.modules.update(fake)
    try:
        with open(path, "wb") as f:
            pickle.dump(EasyDict(encoder=None, dataset_kwargs=dict(path="x"), loss_fn=None, ema=top), f)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_snapshot_codec_is_taken_from_the_file(tmp_path):
    """generate_images.py:170-173: encoder=None means the snapshot's own codec; one this build does not have must raise instead of
    being replaced by the RGB codec."""
    import pickle
    import vivid_amd
    from vivid_amd.generate import generate_images_nvs
    from vivid_amd import snapshot
    cfg = CASES["tiny_dual"]["cfg"]
    path = str(tmp_path / "network-snapshot-0000002.pkl")
    _write_snapshot_like(path, cfg, vivid_amd.synth_state_dict(cfg, seed=3))
    data = snapshot.read_snapshot(path)
    assert isinstance(snapshot.snapshot_encoder(data), vivid_amd.StandardRGBEncoder)
    data["encoder"] = snapshot.SnapshotNet("StabilityVAEEncoder", {})
    with pytest.raises(TypeError, match="unsupported encoder"):
        snapshot.snapshot_encoder(data)


def test_driver_accepts_snapshot_paths(tmp_path):
    """generate_images_nvs(net="network-snapshot-*.pkl") (generate_images.py:164-169) through vivid_amd.snapshot: the images equal
    those of the module built directly from the same (fp16-rounded) weights."""
    import vivid_amd
    from vivid_amd.generate import generate_images_nvs
    cfg = CASES["tiny_dual"]["cfg"]
    sd = vivid_amd.synth_state_dict(cfg, seed=3)
    path = str(tmp_path / "network-snapshot-0000001.pkl")
    _write_snapshot_like(path, cfg, sd)
    direct = vivid_amd.NVPrecond.from_config(cfg)
    direct.load_state_dict({k: v.to(torch.float16).to(torch.float32) for k, v in sd.items()})
    direct = direct.cuda()
    a = list(generate_images_nvs(path, seeds=[16, 17], max_batch_size=2, data=_data(4, 16, 5), num_steps=2))
    b = list(generate_images_nvs(direct, seeds=[16, 17], max_batch_size=2, data=_data(4, 16, 5), num_steps=2))
    assert a[0].images.shape == (2, 3, 16, 16)
    assert torch.equal(a[0].images, b[0].images)


@pytest.mark.parametrize("mode,hin,hout", [("bicubic", 64, 518), ("bicubic", 37, 20), ("bilinear", 518, 64), ("bilinear", 9, 33)])
def test_resize_align_corners_modes_match_torch(mode, hin, hout):
    from vivid_amd.encoders import _resize
    g = torch.Generator().manual_seed(hin + hout)
    x = torch.randn(2, 3, hin, hin + 3, generator=g)
    ref = torch.nn.functional.interpolate(x, size=(hout, hout + 1), mode=mode, align_corners=True)
    got = _resize(x.cuda(), (hout, hout + 1), mode, True)
    assert rel_l2(got.cpu(), ref) < 2e-6


def test_depth_front_end_matches_reference_formulas():
    """depth_prepare / get_depth / add_depth of training/utils.py:107-139 with a stand-in depth network (the real one is external).
    kornia's resize is F.interpolate with the same arguments (kornia itself is not installed: SURVEY 8(c))."""
    import vivid_amd
    g = torch.Generator().manual_seed(11)
    img = torch.randint(0, 256, (2, 3, 64, 64), generator=g).float()
    src = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    want_in = ((torch.nn.functional.interpolate(img / 255, size=(518, 518), mode="bicubic", align_corners=True) - mean) / std).to(torch.float16)
    got_in = vivid_amd.depth_prepare(img.cuda())
    assert got_in.dtype == torch.float16 and got_in.shape == (2, 3, 518, 518)
    assert rel_l2(got_in.float().cpu(), want_in.float()) < 1e-3          # fp16 output: one rounding apart at most

    def fake_model(x):                                                   # [N,3,518,518] fp16 -> positive "depth" [N,518,518]
        return x.float().abs().mean(dim=1) + 0.5

    want_d = torch.nn.functional.interpolate(fake_model(want_in)[:, None], (64, 64), mode="bilinear", align_corners=True)
    got_d = vivid_amd.get_depth(fake_model, img.cuda())
    assert got_d.shape == (2, 1, 64, 64) and rel_l2(got_d.cpu(), want_d) < 1e-3
    for inv in (False, True):
        d = want_d
        if inv:
            d = 1 / d
            d = d / d.amax((1, 2, 3), keepdim=True)
            d = (d - 0.4947) / 0.2294
        want = torch.cat([src, d], dim=1)
        got = vivid_amd.add_depth_from_model(fake_model, img.cuda(), src.cuda(), inv)
        assert got.shape == (2, 4, 64, 64) and rel_l2(got.cpu(), want) < 2e-3
