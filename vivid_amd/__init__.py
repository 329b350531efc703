"""vivid_amd — MI355X-native (gfx950) denoiser for VIVID novel-view synthesis.

Drop-in for the reference's hot path: ``NVPrecond`` (training/models.py) and
``edm_sampler`` (generate_images.py), computed by hand-written HIP kernels in
``libvivid_hip.so``.  See DESIGN.md.
"""
from .arch import NetConfig, vivid_base, vivid_sr, vivid_uncond  # noqa: F401
from .encoders import StandardRGBEncoder, add_depth, add_depth_from_model, depth_prepare, get_depth  # noqa: F401
from .generate import generate_images_nvs  # noqa: F401
from .snapshot import load_network_pkl, read_snapshot  # noqa: F401
from .net import NVPrecond  # noqa: F401
from .sampler import StackedRandomGenerator, edm_sampler  # noqa: F401
from .weights import synth_state_dict  # noqa: F401

__all__ = ["NVPrecond", "edm_sampler", "StackedRandomGenerator", "StandardRGBEncoder", "add_depth", "add_depth_from_model", "depth_prepare", "get_depth", "generate_images_nvs", "load_network_pkl", "read_snapshot", "NetConfig", "vivid_base", "vivid_sr",
           "vivid_uncond", "synth_state_dict"]
