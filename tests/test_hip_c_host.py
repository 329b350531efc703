"""GPU: a compiled C host of libvivid_hip.so (tests/c_host/net_host.c: C99, no Python, no C++) runs a whole NVPrecond evaluation through
vh_net_create ... vh_net_run (and the sampler's split evaluation vh_net_encode + vh_net_run_bound) and must reproduce the reference's
golden D_x (training/models.py:628-689, fixtures generated from the imported reference) and vivid_amd.NVPrecond bit for bit.

What this pins that the ctypes tests cannot: the header compiles as C, a C compiler's layout of vh_net_config is the one the library
reads (the blob carries the struct bytes ctypes wrote: a size or offset mismatch changes the architecture the library builds), the
parameter table can be walked and bound by name from C, and nothing in the call sequence leans on Python-side state."""
import ctypes as C
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest
import torch

from tests.conftest import rel_l2
from tests.golden.cases import CASES, make_inputs, x_for

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_host(tmp):
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    exe = os.path.join(tmp, "net_host")
    lib = os.path.join(ROOT, "vivid_amd")
    subprocess.check_call([cc, "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                           os.path.join(ROOT, "tests", "c_host", "net_host.c"), "-L", lib, "-lvivid_hip", "-L", "/opt/rocm/lib", "-lamdhip64",
                           f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def write_blob(path, cfg_c, batch, mode, sd, inputs):
    def tensor(f, name, t):
        nb = name.encode()
        f.write(struct.pack("<i", len(nb)) + nb)
        if t is None:
            f.write(struct.pack("<q", 0))
        else:
            a = np.ascontiguousarray(t.detach().cpu().to(torch.float32).numpy())
            f.write(struct.pack("<q", a.size) + a.tobytes())
    with open(path, "wb") as f:
        raw = bytes(cfg_c)
        f.write(struct.pack("<ii", 0x56484E54, len(raw)) + raw + struct.pack("<iii", batch, mode, len(sd)))
        for k, v in sd.items():
            tensor(f, k, v)
        for name in ("src", "x", "sigma", "geometry", "cond"):
            tensor(f, name, inputs.get(name))


def test_c_host_builds_without_a_gpu(tmp_path):
    """(CPU) the host program compiles and links against the header and the library as they are."""
    assert os.path.exists(build_host(str(tmp_path)))


@pytest.mark.gpu
@pytest.mark.parametrize("name,mode", [("tiny_dual", 0), ("tiny_dual", 1), ("tiny_sr", 0)])
def test_c_host_reproduces_golden_and_python(name, mode, tmp_path, golden_dir):
    import vivid_amd
    from vivid_amd.cnet import c_config
    case = CASES[name]
    cfg = case["cfg"]
    if cfg.super_res:               # (the host passes no conditioning noise: noisy_sr = 0 on both sides)
        cfg = vivid_amd.NetConfig(**{**cfg.to_dict(), "noisy_sr": 0.0})
    sd = vivid_amd.synth_state_dict(cfg, seed=case["seed"])
    inp = make_inputs(case)
    sigma = case["sigmas"][0]
    rows = inp["src"].shape[0]
    ins = dict(src=inp["src"], x=x_for(inp, sigma), sigma=torch.full((rows,), float(sigma)), geometry=inp["geometry"], cond=inp.get("cond"))
    exe = build_host(str(tmp_path))
    blob, out = str(tmp_path / "blob.bin"), str(tmp_path / "out.bin")
    write_blob(blob, c_config(cfg, True), rows // 2, mode, sd, ins)
    r = subprocess.run([exe, blob, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    R = cfg.img_resolution
    got = torch.from_numpy(np.fromfile(out, dtype=np.float32).reshape(rows // 2, 3, R, R))
    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    assert rel_l2(got, g["D_0"]) < 1e-4                       # the reference's own D_x
    py = vivid_amd.NVPrecond.from_config(cfg, precision="bf16x3")
    py.load_state_dict(sd, strict=True)
    py.noisy_sr = 0.0
    py = py.cuda()
    want = py(*(None if v is None else v.cuda() for v in (ins["src"], ins["x"], ins["sigma"], ins["geometry"], ins["cond"])))
    assert torch.equal(got, want.cpu())
