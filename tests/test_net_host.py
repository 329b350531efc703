"""CPU: host-side behaviour of the drop-in NVPrecond / sampler objects that needs no GPU.

* weight-staleness detection (training/models.py:115-120 re-normalises weights on every forward, so the reference can never
  run with stale weights; here they are prepared once per weight version and the version must notice every way of changing them);
* `noisy_sr` is read per call like the reference's attribute (:658);
* StackedRandomGenerator's contract (generate_images.py:120-134): per-seed streams, independent of batch composition;
* the binding refuses diagnostic builds, the product build reports none.
"""
import copy

import pytest
import torch

import vivid_amd
from oracle import vivid_ref as R
from tests.golden.cases import CASES


def _tiny():
    return vivid_amd.NVPrecond.from_config(CASES["tiny_dual"]["cfg"])


def test_fingerprint_notices_every_kind_of_weight_change():
    net = _tiny()
    fp = [net._fingerprint()]

    def changed():
        fp.append(net._fingerprint())
        return fp[-1] != fp[-2]

    assert not changed()
    p = next(net.parameters())
    p.data.add_(1.0)                                             # in-place edit through .data does not bump _version ...
    p.add_(0.0)                                                  # ... an in-place op on the tensor does
    assert changed()
    node = net.unet.enc
    name = next(iter(node._modules))
    blk = node._modules[name]
    pname = next(iter(blk._parameters))
    setattr(blk, pname, torch.nn.Parameter(torch.zeros_like(blk._parameters[pname]), requires_grad=False))   # EMA swap by assignment
    assert changed()
    blk.register_buffer("extra_buf", torch.zeros(1))
    assert changed()
    del blk.extra_buf
    assert changed()
    net.load_state_dict(net.state_dict())
    assert changed()
    net.float()                                                  # _apply
    assert changed()
    assert not changed()
    twin = copy.deepcopy(net)                                    # a copy gets its own epoch cell, shared by all of ITS nodes
    t0 = twin._fingerprint()
    blk.register_buffer("extra_buf2", torch.zeros(1))
    assert twin._fingerprint() == t0 and changed()


def test_noisy_sr_is_a_live_attribute():
    cfg = CASES["tiny_sr"]["cfg"]
    net = vivid_amd.NVPrecond.from_config(cfg)
    assert net.noisy_sr == cfg.noisy_sr
    net.noisy_sr = 0.0
    assert net.noisy_sr == 0.0 and net.cfg.noisy_sr == cfg.noisy_sr


def test_stacked_generator_streams_depend_on_the_seed_only():
    a = vivid_amd.StackedRandomGenerator("cpu", [16, 17, 2 ** 32 + 18])
    x = a.randn([3, 3, 8, 8])
    y = a.randn_like(x)
    b = vivid_amd.StackedRandomGenerator("cpu", [18])
    assert torch.equal(b.randn([1, 3, 8, 8])[0], x[2])           # seed taken mod 2^32, row independent of its batch
    assert torch.equal(b.randn_like(x[:1])[0], y[2])             # second draw continues the same stream
    ref = R.StackedRandomGenerator("cpu", [16, 17, 18]).randn([3, 3, 8, 8])
    assert torch.equal(ref, x)
    with pytest.raises(ValueError):
        a.randn([2, 3, 8, 8])
    # randint (generate_images.py:132-134): the same per-seed streams, the reference's keyword-only `size`
    c = vivid_amd.StackedRandomGenerator("cpu", [5, 6]).randint(10, size=[2, 4])
    want = torch.stack([torch.randint(10, size=[4], generator=torch.Generator().manual_seed(s)) for s in (5, 6)])
    assert torch.equal(c, want)
    assert torch.equal(vivid_amd.StackedRandomGenerator("cpu", [6]).randint(10, size=[1, 4])[0], c[1])


def test_library_is_a_product_build():
    from vivid_amd import _lib
    L = _lib.lib()
    assert L.vh_diag_flags() == 0
    assert L.vh_abi_version() == _lib.ABI_VERSION == 5
    for knob, default in (("conv_korder", -1), ("conv_stagger", -1), ("attn_pipe", 1), ("attn_nomax", 1), ("attn_xcd", 1), ("attn_m16", 1)):
        _lib.set_knob(knob, default)
    with pytest.raises(_lib.VividHipError):
        _lib.set_knob("no_such_knob", 1)
