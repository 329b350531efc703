"""GPU: the FID / PSNR statistics path (vivid_amd.metrics on vh_moments / vh_psnr_sum) against numpy fp64 restatements of
calculate_metrics.py:147,158-182, and its end-of-run collective on backend "nccl" (RCCL) at world size 1."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests._nccl_metrics_worker import Detector, fake_batches

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _numpy_stats(batches, det):
    """calculate_metrics.py:158-182 with numpy in fp64 (features cast from the detector's fp32 output)."""
    fg = np.concatenate([det(b["images"].cuda()).cpu().double().numpy() for b in batches])
    ft = np.concatenate([det(b["tgt"].cuda()).cpu().double().numpy() for b in batches])
    fs = np.concatenate([det(b["src"].cuda()).cpu().double().numpy() for b in batches])
    n = fg.shape[0]

    def gauss(f):
        mu = f.sum(0) / n
        return mu, (f.T @ f - np.outer(mu, mu) * n) / (n - 1)
    x = torch.cat([b["images"] for b in batches]).float()
    y = torch.cat([b["tgt"] for b in batches]).float()
    psnr = float((10 * torch.log10(255 ** 2 / ((x - y) ** 2).mean((1, 2, 3)))).double().mean())
    return gauss(fg), gauss(ft), gauss(np.concatenate([fg, fs], 1)), psnr, n


def _check(z, batches, det):
    (mu, sg), (rmu, rsg), (jmu, jsg), psnr, n = _numpy_stats(batches, det)
    assert int(z["n"]) == n
    for got, want in ((z["mu"], mu), (z["sigma"], sg), (z["rmu"], rmu), (z["rsigma"], rsg), (z["jmu"], jmu), (z["jsigma"], jsg)):
        np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-12 * np.abs(want).max())
    assert abs(float(np.asarray(z["psnr"]).reshape(-1)[0]) - psnr) < 1e-6 * abs(psnr)


def test_device_statistics_match_numpy_fp64():
    from vivid_amd import metrics as vm
    det = Detector(torch.device("cuda"))
    batches = fake_batches(23, 5)
    it = vm.calculate_stats_for_iterable_nvs(batches, {"fid": det}, metrics=["fid", "joint_fid", "psnr"], device="cuda")
    assert len(it) == len(batches)
    for r, ref in it:
        pass
    z = dict(mu=r.stats["fid"]["mu"], sigma=r.stats["fid"]["sigma"], jmu=r.stats["joint_fid"]["mu"], jsigma=r.stats["joint_fid"]["sigma"],
             rmu=ref.stats["fid"]["mu"], rsigma=ref.stats["fid"]["sigma"], psnr=r.stats["psnr"]["val"], n=r.stats["num_images"])
    _check(z, batches, det)
    res = vm.calculate_metrics_from_stats_nvs(r.stats, ref.stats)
    assert res["fid"] > 0 and res["joint_fid"] > 0 and 0 < res["psnr"] < 20
    assert abs(vm.calculate_metrics_from_stats_nvs(r.stats, r.stats, metrics=["fid"])["fid"]) < 1e-6


def test_statistics_with_rccl_process_group(tmp_path):
    """The reference's one data-carrying collective (calculate_metrics.py:176-182,236) on RCCL: a child process initialises
    torch.distributed with backend "nccl" before touching the GPU, runs the statistics on cuda:0 and all_reduces the bank."""
    out = str(tmp_path / "stats.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_nccl_metrics_worker.py"), out], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    z = np.load(out)
    _check(z, fake_batches(23, 5), Detector(torch.device("cuda")))
    assert float(z["fid"]) > 0
