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


STRUCT_MIRRORS = {"vh_prep_weight_args": "PrepWeightArgs", "vh_conv_args": "ConvArgs", "vh_pixnorm_args": "PixnormArgs",
                  "vh_qkv_split_args": "QkvSplitArgs", "vh_split_args": "SplitArgs", "vh_attention_args": "AttentionArgs",
                  "vh_embed_args": "EmbedArgs", "vh_linear_args": "LinearArgs", "vh_segment": "Segment", "vh_assemble_args": "AssembleArgs",
                  "vh_precond_out_args": "PrecondOutArgs", "vh_warp_args": "WarpArgs", "vh_sampler_step_args": "SamplerStepArgs",
                  "vh_qkv_epilogue": "QkvEpilogue", "vh_codec_args": "CodecArgs", "vh_add_depth_args": "AddDepthArgs",
                  "vh_resize_args": "ResizeArgs", "vh_nonzero_args": "NonzeroArgs", "vh_resample_args": "ResampleArgs",
                  "vh_moments_args": "MomentsArgs", "vh_psnr_args": "PsnrArgs", "vh_net_config": "NetConfigC", "vh_s8_sink": "S8Sink", "vh_layout_args": "LayoutArgs", "vh_axpy_args": "AxpyArgs", "vh_sampler_config": "SamplerConfigC"}


def _header_structs():
    """{struct name: [field names in declaration order]} parsed from the header's `typedef struct { ... } name;` blocks."""
    out = {}
    for m in re.finditer(r"typedef struct \{([^{}]*)\}\s*(\w+)\s*;", _header(), flags=re.S):
        body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for d in decl.split(","):
                d = re.sub(r"\[[^\]]*\]", "", d).strip()          # drop array extents
                names.append(re.findall(r"(\w+)\s*$", d)[0])       # the declarator's identifier is its last word
        out[m.group(2)] = names
    return out


def test_struct_mirrors_match_header_layout(tmp_path):
    """Layout, not field counts: a C program compiled against include/vivid_hip.h prints sizeof / offsetof of every field of every
    argument struct; the ctypes mirrors in _lib.py must agree field by field (a size_t <-> int swap or two reordered fields of the same
    count would pass a count check and corrupt every call)."""
    import shutil
    import subprocess
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    structs = _header_structs()
    assert set(structs) == set(STRUCT_MIRRORS), (sorted(set(structs) ^ set(STRUCT_MIRRORS)))
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "vivid_hip.h"', "int main(void) {"]
    for sname, fields in structs.items():
        lines.append(f'  printf("S {sname} %zu\\n", sizeof({sname}));')
        for f in fields:
            lines.append(f'  printf("F {sname} {f} %zu %zu\\n", offsetof({sname}, {f}), sizeof((({sname}*)0)->{f}));')
    lines += ["  return 0;", "}"]
    src, exe = tmp_path / "layout.c", tmp_path / "layout"
    src.write_text("\n".join(lines))
    subprocess.check_call([cc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    rows = subprocess.check_output([str(exe)], text=True).split("\n")
    sizes, fields = {}, {}
    for r in rows:
        t = r.split()
        if t and t[0] == "S":
            sizes[t[1]] = int(t[2])
        elif t and t[0] == "F":
            fields.setdefault(t[1], []).append((t[2], int(t[3]), int(t[4])))
    for sname, pyname in STRUCT_MIRRORS.items():
        st = getattr(_lib, pyname)
        assert C.sizeof(st) == sizes[sname], (sname, C.sizeof(st), sizes[sname])
        assert len(st._fields_) == len(fields[sname]), (sname, len(st._fields_), len(fields[sname]))
        for (pyfield, ctype), (cfield, off, size) in zip(st._fields_, fields[sname]):
            d = getattr(st, pyfield)
            assert (d.offset, d.size) == (off, size), f"{sname}.{cfield} (ctypes {pyname}.{pyfield}): ctypes offset/size {(d.offset, d.size)}, C {(off, size)}"


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
