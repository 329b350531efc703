#!/usr/bin/env python3
"""Benchmark of the VIVID denoiser hot path on MI355X.

One "step" = one guided denoiser evaluation over one batch, exactly what the reference's sampler
does per call of its `denoise` closure (generate_images.py:55-62) plus the Euler update (:93-98):
    D    = net(src, x, t, labels)            encoder on 2B source rows + x-attn UNet on B rows
    Dref = gnet(src, x, t)                   unconditional guidance net (UNet only)
    x'   = x + (t' - t) * (x - lerp(Dref, D, 1.5)) / t

Workload (BASELINE.json configs[1]): "vivid-base UNet, 256x256, batch 16, CFG=1.5" = the base
architecture (model_channels=128, extra_attn=1) built at img_resolution=256,
B=16 targets = 32 dual-source rows, synthetic inputs and seeded random weights, fp32.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Every rank runs its own batch (independent samples shard with no data-path collective,
generate_images.py:199-200), so scaling is weak; value = N*K steps / max-over-ranks time.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak; bf16x3 executes 3 MFMA products per fp32 product
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # name: (img_resolution, batch, description)
    "c2": (256, 16, "vivid-base arch @256x256, batch 16, CFG 1.5 (net + uncond gnet), dual-source"),
    "base64": (64, 16, "vivid-base @64x64 (the reference's own base stage), batch 16, CFG 1.5"),
    "tiny": (64, 1, "vivid-base @64x64, batch 1, CFG 1.5 (plumbing)"),
}


def measured_traffic(precision):
    """HBM bytes per launch per kernel family from the committed rocprofv3 PMC summary of THIS command
    (profiles/, collected in separate --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE doubled per the gfx950 note in
    MI355X_MICROARCH.md).  None when no summary for this precision is committed."""
    path = os.path.join(ROOT, "profiles", f"traffic_c2_{precision}.json")
    if not os.path.exists(path):
        return {}, None
    d = json.load(open(path))
    return {k: v["hbm_bytes_per_launch"] for k, v in d.get("families", {}).items()}, os.path.relpath(path, ROOT)


def rho_schedule(num_steps=32, sigma_min=0.002, sigma_max=80.0, rho=7.0):
    idx = torch.arange(num_steps, dtype=torch.float32)
    t = (sigma_max ** (1 / rho) + idx / (num_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
    return torch.cat([t, torch.zeros(1)])


def make_inputs(R, B, seed, device):
    g = torch.Generator("cpu").manual_seed(seed)
    src = (torch.rand(2 * B, 3, R, R, generator=g) * 2 - 1).to(device)
    noise = torch.randn(B, 3, R, R, generator=g).repeat_interleave(2, dim=0).to(device)
    geo = torch.randn(2 * B, 20, generator=g)
    geo[:, [14, 15, 18, 19]] = 0
    return src, noise, geo.to(device)


def cpu_baseline(R, seconds_hint):
    """The CPU oracle (a restatement of the reference's PyTorch-CPU path; kind="port") timed on this
    host on a bounded sample: ONE guided evaluation at batch 1 of the same networks.  A batch-16
    step is 16 such evaluations (cost is linear in batch), so steps/s = 1 / (16 * t_b1)."""
    from oracle import vivid_ref as Rf
    import vivid_amd
    torch.set_grad_enabled(False)
    cores = torch.get_num_threads()
    cfg = vivid_amd.vivid_base(R)
    ucfg = vivid_amd.vivid_uncond(R)
    d, ud = cfg.to_dict(), ucfg.to_dict()
    d.pop("use_fp16"); ud.pop("use_fp16")
    net = Rf.OracleNet(Rf.make_config(**d), vivid_amd.synth_state_dict(cfg, seed=0))
    gnet = Rf.OracleNet(Rf.make_config(**ud), vivid_amd.synth_state_dict(ucfg, seed=1))
    src, noise, geo = make_inputs(R, 1, 1, "cpu")
    t = torch.full((2,), 5.0)
    x = noise * 5.0
    t0 = time.perf_counter()
    D = net(src, x, t, geo)
    ref = gnet(src, x, t)
    _ = ref.lerp(D, 1.5)
    dt = time.perf_counter() - t0
    return dict(seconds_b1=dt, cores=cores)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="override per-GPU batch (non-default runs are not the headline)")
    ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "fp32"],
                    help="bf16x3: fp32 emulated by a bf16 hi/lo split on bf16 MFMA (default); fp32: exact fp32 MFMA")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip per-kernel HIP-event timing in the timed region")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs a launcher: python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...")
        sys.exit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback for the product path)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)       # one process per GPU; wraps only in single-GPU rehearsals
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("VIVID_BENCH_BACKEND", "nccl")      # "nccl" = RCCL on ROCm; "gloo" only to rehearse ranks on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import vivid_amd
    from vivid_amd import _lib
    from vivid_amd.sampler import _context, _step

    R, B, desc = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    cfg, ucfg = vivid_amd.vivid_base(R), vivid_amd.vivid_uncond(R)
    net = vivid_amd.NVPrecond.from_config(cfg, precision=args.precision)
    net.load_state_dict(vivid_amd.synth_state_dict(cfg, seed=0), strict=True)
    net = net.to(dev)
    gnet = vivid_amd.NVPrecond.from_config(ucfg, precision=args.precision)
    gnet.load_state_dict(vivid_amd.synth_state_dict(ucfg, seed=1), strict=True)
    gnet = gnet.to(dev)
    src, noise, geo = make_inputs(R, B, 100 + rank, dev)
    t_steps = rho_schedule()
    sctx = _context(dev)
    state = {"x": (noise * float(t_steps[0])).contiguous()}

    def step(i):
        j = i % 32
        t_hat, t_next = float(t_steps[j]), float(t_steps[j + 1])
        x = state["x"]
        tt = torch.full((x.shape[0],), t_hat, device=dev)
        D = net(src, x, tt, geo)
        ref = gnet(src, x, tt)
        d_cur = torch.empty_like(D)
        x_next = torch.empty_like(x)
        _step(sctx, x, None, D, ref, 1.5, d_cur, t_hat, t_next, x_next)
        if j == 31:      # t_next = 0 ends a trajectory: restart from noise
            x_next = (noise * float(t_steps[0])).contiguous()
        state["x"] = x_next

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()

    ctxs = [net._engine.ctx, gnet._engine.ctx, sctx]
    profile = not args.no_profile
    if profile:
        for c in ctxs:
            c.profile_enable(True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64,
                            device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())

    fam = {}
    if profile:
        for c in ctxs:
            for k, v in c.profile_read().items():
                a = fam.setdefault(k, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
                for kk in a:
                    a[kk] += v[kk]
            c.profile_enable(False)
    finite = bool(torch.isfinite(state["x"]).all().item())

    if rank == 0:
        out = {
            "metric": "denoise-steps/sec", "value": world * args.steps / elapsed,
            "unit": f"guided denoiser evaluations/s (batch {B} per GPU, {R}x{R})",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16x3" if args.precision == "bf16x3" else "f32", "data": "synthetic",
            "config": {"workload": desc, "img_resolution": R, "batch_per_gpu": B, "guidance": 1.5, "precision": args.precision,
                       "global_batch": B * world, "parallelism": f"dp{world} (independent samples per rank, no collective)",
                       "params": {"net": sum(p.numel() for p in net.parameters()), "gnet": sum(p.numel() for p in gnet.parameters())}},
            "finite": finite,
        }
        if fam:
            steps = args.steps
            kern = {}
            for k, v in fam.items():
                if v["launches"] == 0:
                    continue
                s = v["ms"] / 1000.0
                kern[k] = {"ms_per_step": v["ms"] / steps, "launches_per_step": v["launches"] / steps,
                           "tflops": v["flops"] / s / 1e12 if s > 0 else 0.0, "gbs": v["bytes"] / s / 1e9 if s > 0 else 0.0,
                           "gflop_per_step": v["flops"] / steps / 1e9, "gbyte_per_step": v["bytes"] / steps / 1e9}
            dom = max(kern, key=lambda k: kern[k]["ms_per_step"])
            kd, vd = kern[dom], fam[dom]
            mfma_bound = dom in ("conv3x3", "conv1x1", "attention")
            x3 = args.precision == "bf16x3" and dom in ("conv3x3", "attention")
            if mfma_bound:
                achieved, unit = kd["tflops"], "TFLOP/s"
                peak = PEAK_BF16_MFMA_TFLOPS / 3.0 if x3 else PEAK_FP32_MFMA_TFLOPS
            else:
                achieved, peak, unit = kd["gbs"], PEAK_HBM_GBS, "GB/s"
            traffic, tsrc = measured_traffic(args.precision) if args.workload == "c2" and not args.batch else ({}, None)
            out["roofline"] = {"kernel": dom, "bound": "mfma" if mfma_bound else "hbm", "achieved": achieved, "peak": peak,
                               "unit": unit, "frac": achieved / peak, "traffic": traffic.get(dom), "traffic_source": tsrc,
                               "algorithmic_bytes_per_launch": vd["bytes"] / vd["launches"],
                               "avg_launch_ms": vd["ms"] / vd["launches"], "launches": vd["launches"],
                               "algorithmic_per_launch": (vd["flops"] if mfma_bound else vd["bytes"]) / vd["launches"],
                               "share_of_step": kd["ms_per_step"] / out["ms_per_step"],
                               "note": ("achieved = algorithmic fp32 FLOPs/s; every fp32 product is 3 bf16 MFMA products (hi*hi+hi*lo+lo*hi, "
                                        "fp32 accumulate), so peak = 2500 TF/s dense bf16 MFMA / 3; executed MFMA rate = 3 x achieved"
                                        if x3 else "fp32 operands on v_mfma_f32_32x32x2_f32; peak = fp32 matrix peak (no xf32 on gfx950)")}
            tr = traffic.get(dom)
            if tr is not None:                              # measured HBM stream of the same kernel: bytes per launch / launch time
                out["roofline"]["hbm_gbs"] = tr / (vd["ms"] / vd["launches"] / 1e3) / 1e9
                out["roofline"]["hbm_frac"] = out["roofline"]["hbm_gbs"] / PEAK_HBM_GBS
            # north_star also asks for the attention kernels' MFMA utilisation: same definition, second family
            if "attention" in kern and dom != "attention":
                ka, va = kern["attention"], fam["attention"]
                pk = PEAK_BF16_MFMA_TFLOPS / 3.0 if args.precision == "bf16x3" else PEAK_FP32_MFMA_TFLOPS
                out["roofline_attention"] = {"bound": "mfma", "achieved": ka["tflops"], "peak": pk, "unit": "TFLOP/s",
                                             "frac": ka["tflops"] / pk, "avg_launch_ms": va["ms"] / va["launches"],
                                             "traffic": traffic.get("attention"), "share_of_step": ka["ms_per_step"] / out["ms_per_step"]}
            from vivid_amd import engine as _eng
            out["config"]["conv_stagger_autotuned"] = {str(k): v for k, v in _eng._TUNED.items()}   # per device: 1 = staggered DMA issue won the start-up A/B
            out["kernels"] = kern
            tot_fl = sum(v["flops"] for v in fam.values())
            out["whole_step_tflops"] = tot_fl / elapsed / 1e12
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(R, 30)
            out["cpu_baseline"] = {"value": 1.0 / (B * cb["seconds_b1"]), "unit": out["unit"], "cores": cb["cores"],
                                   "kind": "port",
                                   "sample": f"one guided evaluation (net + uncond gnet) of the same {R}x{R} networks at batch 1 on the "
                                             f"CPU oracle took {cb['seconds_b1']:.2f} s; a batch-{B} step is {B} of them (cost linear in batch)"}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
