"""CPU: the C-level network object (vh_net_*, include/vivid_hip.h) without a GPU - what it derives from the constructor arguments.

The architecture generator and the parameter table exist twice: in Python (vivid_amd/arch.py + weights.py, pinned to the
reference by `load_state_dict(strict=True)` when the golden fixtures are made) and in C++ (csrc/net.hip) for hosts without Python.
Here the two are held against each other for every configuration family the reference has: same state_dict keys and shapes
(training/models.py:322-384, 413-480, 524-534, 576-582, 591-624) and the same workspace size, which only comes out equal if the
two walks allocate and release the same buffers in the same order."""
import pytest

import vivid_amd
from tests.golden.cases import CASES
from vivid_amd.cnet import CNet
from vivid_amd.engine import Engine
from vivid_amd.weights import state_dict_shapes

CONFIGS = {
    "vivid_base_64": (vivid_amd.vivid_base(64), True), "vivid_uncond_64": (vivid_amd.vivid_uncond(64), True),
    "vivid_sr_256": (vivid_amd.vivid_sr(256), True), "base_built_at_256": (vivid_amd.vivid_base(256), True),
    "warp_256": (vivid_amd.vivid_base(256, warp_depth_coor=True), True), "sr_built_at_1024": (vivid_amd.vivid_sr(1024), True),
    "tiny_dual": (CASES["tiny_dual"]["cfg"], True), "tiny_vanilla_single_source": (CASES["tiny_vanilla"]["cfg"], False),
    "tiny_depth": (CASES["tiny_depth"]["cfg"], True), "tiny_opts_fir_filter": (CASES["tiny_opts"]["cfg"], True),
}


@pytest.mark.parametrize("name", list(CONFIGS))
def test_parameter_table_matches_the_state_dict_layout(name):
    cfg, dual = CONFIGS[name]
    net = CNet(cfg, dual_source=dual)
    got = dict(net.params())
    want = {k: tuple(v) for k, v in state_dict_shapes(cfg).items()}
    assert got == want
    assert len(net.params()) == len(want)


@pytest.mark.parametrize("name,batch", [("vivid_base_64", 1), ("vivid_base_64", 5), ("vivid_uncond_64", 2), ("vivid_sr_256", 1),
                                        ("base_built_at_256", 2), ("warp_256", 1), ("tiny_vanilla_single_source", 3), ("tiny_opts_fir_filter", 2)])
def test_workspace_size_equals_the_python_engines(name, batch):
    cfg, dual = CONFIGS[name]
    net = CNet(cfg, dual_source=dual)
    eng = Engine(cfg, dual_source=dual, precision="bf16x3")
    assert net.workspace_bytes(batch) == eng.measure_workspace("uncond" if cfg.uncond else "full", batch, has_cond=bool(cfg.super_res))


@pytest.mark.parametrize("name,batch", [("vivid_base_64", 2), ("vivid_uncond_64", 1), ("vivid_sr_256", 1), ("tiny_vanilla_single_source", 3), ("tiny_opts_fir_filter", 1),
                                        ("channels_not_32", 2)])
def test_fp32_walk_workspace_and_parameters_equal_the_python_engines(name, batch):
    """vh_net_config.fp32: the exact-fp32 walk (no S8 tensors, no fused weights) against Engine(precision='fp32'), which also takes channel
    counts the bf16x3 path refuses."""
    cfg, dual = CONFIGS.get(name, (vivid_amd.NetConfig(img_resolution=16, model_channels=16, channel_mult=(1, 4), attn_resolutions=(8,)), True))
    net = CNet(cfg, dual_source=dual, precision="fp32")
    eng = Engine(cfg, dual_source=dual, precision="fp32")
    assert net.workspace_bytes(batch) == eng.measure_workspace("uncond" if cfg.uncond else "full", batch, has_cond=bool(cfg.super_res))
    assert dict(net.params()) == {k: tuple(v) for k, v in state_dict_shapes(cfg).items()}
    if not cfg.uncond:
        assert net._L.vh_net_workspace_bytes_mode(net.handle, CNet.FEATURES, batch) == eng.measure_workspace("features", batch)


def test_encoder_only_workspace_equals_the_python_engines():
    cfg, dual = CONFIGS["vivid_base_64"]
    net = CNet(cfg, dual_source=dual)
    eng = Engine(cfg, dual_source=dual, precision="bf16x3")
    assert net._L.vh_net_workspace_bytes_mode(net.handle, CNet.FEATURES, 3) == eng.measure_workspace("features", 3)
    assert net._L.vh_net_workspace_bytes_mode(net.handle, CNet.BOUND, 3) > 0
    unc = CNet(vivid_amd.vivid_uncond(64))
    assert unc._L.vh_net_workspace_bytes_mode(unc.handle, CNet.FEATURES, 1) == 0      # an uncond net has no encoder


def test_bad_configurations_are_refused():
    from vivid_amd import _lib as L
    with pytest.raises(L.VividHipError, match="multiples of 32"):
        CNet(vivid_amd.NetConfig(img_resolution=16, model_channels=48))
    with pytest.raises(ValueError, match="resample_filter"):                  # (the reference asserts an even length, training/models.py:52)
        CNet(vivid_amd.NetConfig(**{**CASES["tiny_opts"]["cfg"].to_dict(), "resample_filter": (1.0, 2.0, 1.0)}))
    net = CNet(vivid_amd.vivid_base(64))
    with pytest.raises(L.VividHipError, match="no parameter named"):
        L.check(net._L.vh_net_bind_param(net.handle, b"encoder.nonsense", 16), "vh_net_bind_param")
