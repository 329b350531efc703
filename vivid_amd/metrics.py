"""FID / joint-FID / PSNR statistics of a stream of generated batches on MI355X.

What the reference computes (``calculate_metrics.py:134-248``): per detector, the fp64 sum and second-moment matrix of the
features of generated and of ground-truth images, the same for the joint features ``[image | source]``, PSNR against the
target, an ``all_reduce`` of every accumulator on the last batch, then (``:295-322``) the Fréchet distance between the two
Gaussians.  The call surface is kept (:func:`calculate_stats_for_iterable_nvs` yields ``(r, ref)`` records whose ``.stats`` is
filled on the last batch; :func:`calculate_metrics_from_stats_nvs` turns two stats dicts into numbers); the data path is
designed for this machine:

* **one bank, five blocks.**  Every accumulator of a run lives in ONE flat fp64 device buffer (:class:`MomentBank`).  The
  joint second moment of ``[g | s]`` is ``[[g'g, g's], [s'g, s's]]``, so instead of the reference's two extra (2F)² products
  per batch the bank keeps the blocks ``g'g, t't, s's, g's, t's`` once (5 F² doubles instead of 10 F²) and assembles the joint
  matrices when the run ends;
* **fp64 on the matrix cores.**  Blocks are accumulated by ``vh_moments`` (``v_mfma_f64_16x16x4_f64``, products of the fp32
  features in fp64 = the reference's ``features.to(float64)``), PSNR by ``vh_psnr_sum``; there is no per-batch host
  synchronisation;
* **one collective.**  The bank is all_reduced as a single bucket at the end (RCCL over xGMI when the process group's backend
  is "nccl"): ≤ 170 MB once per run for Inception-sized features, instead of 8 + 2 per-batch collectives.  The running image
  count therefore is the LOCAL count until the last batch (the reference all_reduces two scalars and reads them back on the
  host every batch, for its progress display only).

Detectors (Inception-v3 from an NGC pickle, DINOv2 from torch.hub; ``calculate_metrics.py:45-47,63``) need the network and
stay external: callables ``images_uint8[N,3,H,W] -> features[N,F]`` with a ``feature_dim`` attribute.  The matrix square root
of the Fréchet distance is taken on the host with scipy, as in the reference.
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import os

import numpy as np
import torch

from . import _lib as L
from .generate import EasyDict

def _rank() -> int:
    return torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0


DETECTOR_METRICS = ("fid", "fd_dinov2")
STAT_METRICS = DETECTOR_METRICS + tuple("joint_" + m for m in DETECTOR_METRICS)


class MomentBank:
    """All fp64 accumulators of one metrics run in one flat buffer.

    ``dims``: {detector metric: feature dim}; ``joint``: the detector metrics whose joint [image | source] statistics are
    wanted.  Tensors handed to :meth:`add_features` / :meth:`add_psnr` must live on the bank's device.  On a GPU the
    accumulation runs in libvivid_hip.so; a bank on the CPU exists only to rehearse the multi-rank reduce over gloo and has to
    be asked for explicitly (``allow_host=True``) - it is never chosen silently."""

    def __init__(self, dims: Dict[str, int], joint: Sequence[str] = (), device="cuda", allow_host: bool = False):
        self.device = torch.device(device)
        if self.device.type != "cuda" and not allow_host:
            raise RuntimeError("MomentBank accumulates on the GPU (vh_moments); pass allow_host=True for a CPU rehearsal bank")
        self.dims, self.joint = dict(dims), tuple(j for j in joint if j in dims)
        self._slots: Dict[Tuple[str, str], Tuple[int, Tuple[int, ...]]] = {}
        off = 0

        def slot(metric, name, *shape):
            nonlocal off
            n = int(np.prod(shape))
            self._slots[(metric, name)] = (off, tuple(shape))
            off += (n + 1) // 2 * 2                          # keep every block 16-byte aligned

        slot("", "counts", 2)                                # images, targets seen
        slot("", "psnr", 1)
        for m, F in self.dims.items():
            for who in "gt":
                slot(m, "sum_" + who, F)
                slot(m, who + who, F, F)
            if m in self.joint:
                slot(m, "sum_s", F)
                for blk in ("ss", "gs", "ts"):
                    slot(m, blk, F, F)
        self.flat = torch.zeros(off, dtype=torch.float64, device=self.device)
        self._ctx = None

    def view(self, metric: str, name: str) -> torch.Tensor:
        off, shape = self._slots[(metric, name)]
        return self.flat[off:off + int(np.prod(shape))].view(shape)

    # -- accumulation ----------------------------------------------------------------------------
    def _context(self):
        from .sampler import _context
        return _context(self.device)

    def _outer(self, metric: str, block: str, a: torch.Tensor, b: torch.Tensor, sum_name: Optional[str]):
        """block += a^T b (fp64 products of the fp32 features); sum_name += column sums of a."""
        out = self.view(metric, block)
        ssum = self.view(metric, sum_name) if sum_name else None
        if a.shape[0] == 0:
            return
        if self.device.type == "cuda":
            a32, b32 = a.to(torch.float32).contiguous(), b.to(torch.float32).contiguous()
            with torch.cuda.device(self.device):
                self._context().call("vh_moments", L.MomentsArgs(a=a32.data_ptr(), b=b32.data_ptr(), n=a32.shape[0], fa=a32.shape[1],
                                                                 fb=b32.shape[1], outer=out.data_ptr(),
                                                                 sum_a=ssum.data_ptr() if ssum is not None else None))
        else:
            a64, b64 = a.to(torch.float64), b.to(torch.float64)
            out += a64.T @ b64
            if ssum is not None:
                ssum += a64.sum(0)

    def add_features(self, metric: str, gen: torch.Tensor, tgt: torch.Tensor, src: Optional[torch.Tensor] = None):
        """Detector features of one batch: generated images, ground truth, and (joint statistics) source views."""
        F = self.dims[metric]
        for f in (gen, tgt) + ((src,) if src is not None else ()):
            if f.ndim != 2 or f.shape[1] != F or f.device != self.flat.device:
                raise ValueError(f"{metric}: features must be [N, {F}] on {self.flat.device}, got {tuple(f.shape)} on {f.device}")
        self._outer(metric, "gg", gen, gen, "sum_g")
        self._outer(metric, "tt", tgt, tgt, "sum_t")
        if metric in self.joint:
            if src is None:
                raise ValueError(f"joint_{metric} needs the source features")
            self._outer(metric, "ss", src, src, "sum_s")
            self._outer(metric, "gs", gen, src, None)
            self._outer(metric, "ts", tgt, src, None)

    def add_psnr(self, images: torch.Tensor, tgt: torch.Tensor):
        """calculate_metrics.py:147 per image, summed into the bank; uint8 or float images on the [0,255] scale."""
        n = images.shape[0]
        if n == 0:
            return
        acc = self.view("", "psnr")
        if self.device.type == "cuda":
            u8 = images.dtype == torch.uint8 and tgt.dtype == torch.uint8
            x = images.contiguous() if u8 else images.to(torch.float32).contiguous()
            y = tgt.contiguous() if u8 else tgt.to(torch.float32).contiguous()
            per_image = torch.empty(n, dtype=torch.float64, device=self.device)     # folded into acc in index order: reproducible
            with torch.cuda.device(self.device):
                self._context().call("vh_psnr_sum", L.PsnrArgs(x=x.data_ptr(), y=y.data_ptr(), images=n, elems=x[0].numel(),
                                                              dtype=0 if u8 else 1, acc=acc.data_ptr(), per_image=per_image.data_ptr()))
        else:
            d = images.to(torch.float32) - tgt.to(torch.float32)
            acc += (10 * torch.log10(255.0 ** 2 / (d * d).mean(dim=(1, 2, 3)))).to(torch.float64).sum()

    def add_counts(self, n_images: int, n_targets: int):
        self.view("", "counts").add_(torch.tensor([float(n_images), float(n_targets)], dtype=torch.float64).to(self.device))

    # -- end of run --------------------------------------------------------------------------------
    def all_reduce(self) -> "MomentBank":
        """The run's one data-carrying collective: SUM over ranks of the whole bank (calculate_metrics.py:176-182, :225-236)."""
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.all_reduce(self.flat, op=torch.distributed.ReduceOp.SUM)
        return self

    def finalize(self, want_psnr: bool) -> Tuple[dict, dict]:
        """(stats of the generated images, stats of the ground truth) in the reference's format: ``{metric: {mu, sigma}}`` as
        numpy fp64, ``num_images``, and ``psnr: {val}`` on the generated side."""
        host = self.flat.cpu().numpy()

        def get(metric, name):
            off, shape = self._slots[(metric, name)]
            return host[off:off + int(np.prod(shape))].reshape(shape)

        n_gen, n_tgt = (int(round(v)) for v in get("", "counts"))
        if n_gen < 2 or n_tgt < 2:
            raise ValueError(f"feature statistics need at least 2 images (got {n_gen} generated, {n_tgt} targets)")

        def gaussian(total, second, n):                      # calculate_metrics.py:176-177
            mu = total / n
            return dict(mu=mu, sigma=(second - np.outer(mu, mu) * n) / (n - 1))

        gen, ref = dict(num_images=n_gen), dict(num_images=n_tgt)
        for m in self.dims:
            gen[m] = gaussian(get(m, "sum_g"), get(m, "gg"), n_gen)
            ref[m] = gaussian(get(m, "sum_t"), get(m, "tt"), n_tgt)
            if m in self.joint:
                ss, s1 = get(m, "ss"), get(m, "sum_s")
                for side, who, n in ((gen, "g", n_gen), (ref, "t", n_tgt)):
                    cross = get(m, who + "s")
                    second = np.block([[get(m, who + who), cross], [cross.T, ss]])
                    side["joint_" + m] = gaussian(np.concatenate([get(m, "sum_" + who), s1]), second, n)
        if want_psnr:
            gen["psnr"] = dict(val=np.array([get("", "psnr")[0] / n_gen]))
        return gen, ref


def calculate_stats_for_iterable_nvs(image_iter: Iterable, detectors: Dict[str, Callable],
                                     metrics: Sequence[str] = ("fid", "joint_fid", "psnr"), device="cuda", dest_path=None):
    """Feature statistics over the records of ``generate_images_nvs`` (fields ``images``, ``tgt``, ``src``: NCHW, [0,255]).

    Returns an iterable with a length; iterating it yields ``(r, ref)`` per batch (``stats=None, images, batch_idx,
    num_batches, num_images``), and on the last batch ``r.stats`` / ``ref.stats`` hold the all_reduced statistics in the
    reference's format.  ``detectors`` maps "fid" / "fd_dinov2" to feature extractors; a ``joint_*`` metric needs its base
    metric (calculate_metrics.py:143-145).  Records are ``generate.EasyDict`` (missing attribute -> AttributeError, as dnnlib's).
    ``num_images`` of the intermediate records is this rank's running count (the reference all_reduces it every batch for its
    progress bar, calculate_metrics.py:224-228; here nothing is exchanged before the last batch, whose records carry the global
    counts).  ``dest_path``: rank 0 pickles the generated-side statistics there after the last batch (the reference calls an
    undefined ``save_stats`` at :239-240; the pickle of the stats dict is what upstream EDM2's function of that name writes)."""
    metrics = list(metrics)
    for m in metrics:
        if m.startswith("joint_") and m[len("joint_"):] not in metrics:
            raise AssertionError(f"{m} needs {m[len('joint_'):]} in metrics")
    used = {m: detectors[m] for m in metrics if m in DETECTOR_METRICS}
    joint = [m for m in used if "joint_" + m in metrics]
    want_psnr = "psnr" in metrics
    dev = torch.device(device)
    num_batches = len(image_iter)

    def batches():
        bank = MomentBank({m: d.feature_dim for m, d in used.items()}, joint, dev, allow_host=dev.type != "cuda")
        seen = 0
        for batch_idx, data in enumerate(image_iter):
            images, tgt, src = (None if data.get(k) is None else torch.as_tensor(data[k]).to(dev) for k in ("images", "tgt", "src"))
            if images is not None and tgt is not None and images.shape[0]:
                with torch.no_grad():
                    for m, det in used.items():
                        bank.add_features(m, det(images), det(tgt), det(src) if m in joint else None)
                    if want_psnr:
                        bank.add_psnr(images, tgt)
                bank.add_counts(images.shape[0], tgt.shape[0])
                seen += images.shape[0]
            common = dict(stats=None, images=images, batch_idx=batch_idx, num_batches=num_batches, num_images=seen)
            r, ref = EasyDict(common), EasyDict(common)
            if batch_idx == num_batches - 1:
                r.stats, ref.stats = bank.all_reduce().finalize(want_psnr)
                r.num_images, ref.num_images = r.stats["num_images"], ref.stats["num_images"]
                if dest_path is not None and _rank() == 0:
                    import pickle
                    os.makedirs(os.path.dirname(os.path.abspath(dest_path)), exist_ok=True)
                    with open(dest_path, "wb") as f:
                        pickle.dump(r.stats, f)
            yield r, ref

    class _Stats:
        def __len__(self):
            return num_batches

        def __iter__(self):
            return batches()

    return _Stats()


def frechet_distance(mu1: np.ndarray, sigma1: np.ndarray, mu2: np.ndarray, sigma2: np.ndarray) -> float:
    """||mu1 - mu2||^2 + tr(S1 + S2 - 2 (S1 S2)^(1/2))   (calculate_metrics.py:313-315)."""
    import scipy.linalg
    root, _ = scipy.linalg.sqrtm(sigma1 @ sigma2, disp=False)
    return float(np.real(np.square(mu1 - mu2).sum() + np.trace(sigma1 + sigma2 - 2 * root)))


def calculate_metrics_from_stats_nvs(stats: dict, ref: dict, metrics: Sequence[str] = ("fid", "joint_fid", "psnr")) -> dict:
    """Numbers from two statistics dicts (calculate_metrics.py:295-322): Fréchet distances for the detector metrics present on
    both sides, everything else (PSNR) passed through from ``stats``."""
    out = {}
    for m in metrics:
        if m in STAT_METRICS:
            if m in stats and m in ref:
                out[m] = frechet_distance(stats[m]["mu"], stats[m]["sigma"], ref[m]["mu"], ref[m]["sigma"])
        elif m in stats:
            out[m] = float(np.asarray(stats[m]["val"]).reshape(-1)[0])
    return out
