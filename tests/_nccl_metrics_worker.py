"""Child process of tests/test_hip_metrics.py: the metrics run of calculate_metrics.py `gen` on ONE GPU with the process group on
backend "nccl" (= RCCL on ROCm), world size 1.  The group is initialised before anything else touches the GPU.  Writes the final
statistics to argv[1] (.npz)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402


def fake_batches(n_seeds, max_batch):
    from vivid_amd import distributed as vd
    out = []
    for idx in vd.rank_batches(n_seeds, max_batch, 1, 0):
        def mk(off):
            return torch.stack([torch.randint(0, 256, (3, 16, 16), generator=torch.Generator().manual_seed(int(i) * 7 + off), dtype=torch.uint8) for i in idx])
        out.append(dict(images=mk(0), tgt=mk(1), src=mk(2)))
    return out


class Detector:
    """Fixed random projection of the 4x4 mean-pooled image: deterministic stand-in for Inception / DINOv2 (both need the network)."""
    feature_dim = 70

    def __init__(self, device):
        g = torch.Generator().manual_seed(3)
        self.w = torch.randn(3 * 4 * 4, self.feature_dim, generator=g).to(device)

    def __call__(self, img):
        x = torch.nn.functional.adaptive_avg_pool2d(img.float() / 255, 4).flatten(1)
        return x @ self.w


def main():
    out_path = sys.argv[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    from vivid_amd import distributed as vd
    vd.init("nccl")                                            # RCCL process group first
    assert torch.distributed.get_backend() == "nccl"
    from vivid_amd import metrics as vm
    dev = torch.device("cuda", 0)
    it = vm.calculate_stats_for_iterable_nvs(fake_batches(23, 5), {"fid": Detector(dev)}, metrics=["fid", "joint_fid", "psnr"], device=dev)
    for r, ref in it:
        pass
    # a second, explicit all_reduce of a bank on RCCL: at world size 1 SUM must leave it unchanged
    bank = vm.MomentBank({"fid": 8}, ("fid",), dev)
    f = torch.randn(6, 8, device=dev)
    bank.add_features("fid", f, f, f)
    before = bank.flat.clone()
    bank.all_reduce()
    torch.cuda.synchronize()
    assert torch.equal(before, bank.flat)
    res = vm.calculate_metrics_from_stats_nvs(r.stats, ref.stats)
    np.savez(out_path, mu=r.stats["fid"]["mu"], sigma=r.stats["fid"]["sigma"], jmu=r.stats["joint_fid"]["mu"],
             jsigma=r.stats["joint_fid"]["sigma"], rmu=ref.stats["fid"]["mu"], rsigma=ref.stats["fid"]["sigma"],
             psnr=r.stats["psnr"]["val"], n=r.stats["num_images"], fid=res["fid"], joint_fid=res["joint_fid"])
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
