"""CPU, world_size 2 over gloo: the seed-sharding rule and the one-bucket fp64 moment all_reduce."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from vivid_amd import distributed as vd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_seeds, max_batch, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    vd.init("gloo")
    try:
        batches = vd.rank_batches(n_seeds, max_batch)
        g = torch.Generator().manual_seed(5)
        feats_all = torch.randn(n_seeds, 8, generator=g)          # "feature of seed i" — same on every rank
        from vivid_amd.metrics import MomentBank
        bank = MomentBank({"f": 8}, (), "cpu", allow_host=True)     # CPU rehearsal bank: the reduce is what is under test
        for b in batches:
            if len(b):
                f = feats_all[torch.as_tensor(b)]
                bank.add_features("f", f, f)
                bank.add_counts(len(b), len(b))
            torch.distributed.barrier()                           # one barrier per batch (generate_images.py:340)
        gen, _ = bank.all_reduce().finalize(False)               # ONE all_reduce of the whole bank
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)   # bench.py's max-over-ranks timing
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), idx=np.concatenate(batches) if batches else np.zeros(0),
                 mu=gen["f"]["mu"], cov=gen["f"]["sigma"], n=gen["num_images"], tmax=float(t))
    finally:
        torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_seeds,max_batch", [(128, 16), (10, 4), (3, 32)])
def test_two_rank_sharding_and_moment_allreduce(tmp_path, n_seeds, max_batch):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, n_seeds, max_batch, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"r{k}.npz") for k in range(world)]
    idx = np.concatenate([x["idx"] for x in r]).astype(int)
    assert sorted(idx.tolist()) == list(range(n_seeds)), "every seed exactly once across ranks"
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(n_seeds, 8, generator=g).double()
    for x in r:
        assert int(x["n"]) == n_seeds and x["tmax"] == 2.0
        np.testing.assert_allclose(x["mu"], feats.mean(0).numpy(), rtol=1e-12, atol=1e-12)
        if n_seeds > 1:
            np.testing.assert_allclose(x["cov"], torch.cov(feats.T).numpy(), rtol=1e-10, atol=1e-12)


def test_rank_batches_matches_reference_rule():
    # C3 of BASELINE.json: 128 seeds, batch 16, 8 ranks -> one batch of 16 per rank
    for rank in range(8):
        b = vd.rank_batches(128, 16, world_size=8, rank=rank)
        assert len(b) == 1 and len(b[0]) == 16 and b[0][0] == 16 * rank
    # uneven: 10 seeds, max 4, 2 ranks -> 4 batches [3,3,2,2], ranks take every other one
    a = [vd.rank_batches(10, 4, world_size=2, rank=r) for r in range(2)]
    assert [len(x) for x in a[0]] == [3, 2] and [len(x) for x in a[1]] == [3, 2]


class _Det:
    """A stand-in feature detector (fixed random projection of the mean-pooled image)."""
    feature_dim = 6

    def __init__(self):
        g = torch.Generator().manual_seed(3)
        self.w = torch.randn(3 * 4 * 4, 6, generator=g)

    def __call__(self, img):
        x = torch.nn.functional.adaptive_avg_pool2d(img.float() / 255, 4).flatten(1)
        return x @ self.w


def _fake_batches(n_seeds, max_batch, world, rank):
    """What generate_images_nvs yields, without a GPU: image/tgt/src per seed are functions of the seed."""
    out = []
    for idx in vd.rank_batches(n_seeds, max_batch, world, rank):
        def mk(off):
            return torch.stack([torch.randint(0, 256, (3, 8, 8), generator=torch.Generator().manual_seed(int(i) * 7 + off)) for i in idx]) \
                if len(idx) else torch.zeros(0, 3, 8, 8, dtype=torch.int64)
        out.append(dict(images=mk(0), tgt=mk(1), src=mk(2)))
    return out


def _metrics_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    vd.init("gloo")
    try:
        from vivid_amd import metrics as vm
        it = vm.calculate_stats_for_iterable_nvs(_fake_batches(12, 3, world, rank), {"fid": _Det()}, metrics=["fid", "joint_fid", "psnr"], device="cpu")
        for r, ref in it:
            pass
        res = vm.calculate_metrics_from_stats_nvs(r.stats, ref.stats)
        np.savez(os.path.join(out_dir, f"m{rank}.npz"), **{k: np.array(v) for k, v in res.items()}, n=r.num_images)
    finally:
        torch.distributed.destroy_process_group()


def test_fid_stats_two_ranks_equal_one_rank(tmp_path):
    from vivid_amd import metrics as vm
    world, port = 2, _free_port()
    mp.spawn(_metrics_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    it = vm.calculate_stats_for_iterable_nvs(_fake_batches(12, 3, 1, 0), {"fid": _Det()}, metrics=["fid", "joint_fid", "psnr"], device="cpu")
    for r, ref in it:
        pass
    single = vm.calculate_metrics_from_stats_nvs(r.stats, ref.stats)
    assert set(single) == {"fid", "joint_fid", "psnr"} and single["fid"] > 0
    for k in range(world):
        m = np.load(tmp_path / f"m{k}.npz")
        assert int(m["n"]) == 12
        for key, v in single.items():
            np.testing.assert_allclose(float(m[key]), v, rtol=1e-8, atol=1e-10)
    # identical statistics give distance 0
    z = vm.calculate_metrics_from_stats_nvs(r.stats, r.stats, metrics=["fid"])
    assert abs(z["fid"]) < 1e-8


def test_bank_is_strict_about_its_device():
    from vivid_amd import metrics as vm
    with pytest.raises(RuntimeError, match="allow_host"):
        vm.MomentBank({"fid": 4}, (), "cpu")


def test_joint_blocks_equal_the_reference_formulation():
    """The five-block bank against calculate_metrics.py:158-182 restated with numpy: moments of cat([f, fs]) computed whole."""
    from vivid_amd import metrics as vm
    g = torch.Generator().manual_seed(9)
    F, n = 5, 11
    fg, ft, fs = (torch.randn(n, F, generator=g) for _ in range(3))
    bank = vm.MomentBank({"fid": F}, ("fid",), "cpu", allow_host=True)
    for lo, hi in ((0, 4), (4, 11)):
        bank.add_features("fid", fg[lo:hi], ft[lo:hi], fs[lo:hi])
        bank.add_counts(hi - lo, hi - lo)
    gen, ref = bank.finalize(False)
    for side, f in ((gen, fg), (ref, ft)):
        j = torch.cat([f, fs], -1).double().numpy()
        mu = j.sum(0) / n
        sigma = (j.T @ j - np.outer(mu, mu) * n) / (n - 1)
        np.testing.assert_allclose(side["joint_fid"]["mu"], mu, rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(side["joint_fid"]["sigma"], sigma, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(side["fid"]["sigma"], sigma[:F, :F], rtol=1e-12, atol=1e-13)
    # closed form: two Gaussians with commuting covariances, d = |dmu|^2 + sum (sqrt(a) - sqrt(b))^2
    a, b = np.array([1.0, 4.0, 9.0]), np.array([4.0, 1.0, 16.0])
    d = vm.frechet_distance(np.zeros(3), np.diag(a), np.array([1.0, 2.0, 2.0]), np.diag(b))
    assert abs(d - (9.0 + ((np.sqrt(a) - np.sqrt(b)) ** 2).sum())) < 1e-9
