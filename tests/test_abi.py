"""CPU: the C-ABI library loads and exports every symbol include/vivid_hip.h declares,
and the ctypes mirrors have the sizes the header implies (no compute calls: no GPU here)."""
import ctypes as C
import os
import re

import pytest

from vivid_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header():
    return open(os.path.join(ROOT, "include", "vivid_hip.h")).read()


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    declared = set(re.findall(r"^(?:int|size_t|const char\*)\s+(vh_\w+)\s*\(", _header(), flags=re.M))
    assert declared, "no declarations found in the header"
    assert declared == set(_lib.OPS) | set(_lib.CONTROL) | set(_lib.NET)
    for name in declared:
        assert hasattr(L, name), name
    assert L.vh_abi_version() == _lib.ABI_VERSION


def test_struct_mirrors_match_header_field_counts():
    h = _header()
    pairs = {"vh_prep_weight_args": _lib.PrepWeightArgs, "vh_conv_args": _lib.ConvArgs, "vh_pixnorm_args": _lib.PixnormArgs,
             "vh_qkv_split_args": _lib.QkvSplitArgs, "vh_split_args": _lib.SplitArgs, "vh_attention_args": _lib.AttentionArgs, "vh_embed_args": _lib.EmbedArgs,
             "vh_linear_args": _lib.LinearArgs, "vh_segment": _lib.Segment, "vh_assemble_args": _lib.AssembleArgs,
             "vh_precond_out_args": _lib.PrecondOutArgs, "vh_warp_args": _lib.WarpArgs,
             "vh_sampler_step_args": _lib.SamplerStepArgs, "vh_qkv_epilogue": _lib.QkvEpilogue, "vh_codec_args": _lib.CodecArgs, "vh_add_depth_args": _lib.AddDepthArgs, "vh_resize_args": _lib.ResizeArgs,
             "vh_nonzero_args": _lib.NonzeroArgs, "vh_resample_args": _lib.ResampleArgs, "vh_moments_args": _lib.MomentsArgs,
             "vh_psnr_args": _lib.PsnrArgs, "vh_net_config": _lib.NetConfigC}
    for cname, st in pairs.items():
        m = re.search(r"typedef struct \{([^{}]*)\}\s*" + cname + r"\s*;", h, flags=re.S)
        assert m, cname
        body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
        n = 0
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            n += decl.count(",") + 1
        assert n == len(st._fields_), (cname, n, len(st._fields_))


def test_argument_validation_without_gpu():
    """Host-side validation rejects bad shapes before any launch (safe without a GPU)."""
    L = _lib.lib()
    ctx = _lib.Context(0)
    a = _lib.ConvArgs(taps=5)
    with pytest.raises(_lib.VividHipError, match="taps"):
        ctx.call("vh_conv", a)
    a = _lib.AttentionArgs(q=16, k=16, v=16, out=16, b=1, heads=1, s=4, kl=4, d=96)
    with pytest.raises(_lib.VividHipError, match="head dim"):
        ctx.call("vh_attention", a)
    ctx.plan_begin()
    with pytest.raises(_lib.VividHipError):
        ctx.plan_begin()
    plan = ctx.plan_end()
    assert plan.num_ops == 0
    # an op refused while recording: vh_plan_abort drops the partial plan and the context records again afterwards
    ctx.plan_begin()
    with pytest.raises(_lib.VividHipError, match="taps"):
        ctx.call("vh_conv", _lib.ConvArgs(taps=5))
    ctx.plan_abort()
    ctx.plan_abort()                      # no-op when not recording
    ctx.plan_begin()
    assert ctx.plan_end().num_ops == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No silent fallback: without libvivid_hip.so the loader raises and so does the product path that needs it."""
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libvivid_hip.so"))
    with pytest.raises(_lib.VividHipError, match="not built"):
        _lib.lib()


def test_cpu_tensors_are_refused_by_the_network():
    """The denoiser has no CPU implementation in the product: CPU inputs raise instead of taking another path."""
    import torch
    import vivid_amd
    from tests.golden.cases import CASES
    cfg = CASES["tiny_dual"]["cfg"]
    net = vivid_amd.NVPrecond.from_config(cfg)
    x = torch.zeros(2, 3, cfg.img_resolution, cfg.img_resolution)
    with pytest.raises(RuntimeError):
        net(x, x, torch.ones(2), torch.zeros(2, 20))
