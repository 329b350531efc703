#!/usr/bin/env python3
"""Benchmark of the VIVID denoiser hot path on MI355X.

One "step" = one guided denoiser evaluation over one batch, exactly what the reference's sampler
does per call of its `denoise` closure (generate_images.py:55-62) plus the Euler update (:93-98):
    D    = net(src, x, t, labels)            encoder on 2B source rows + x-attn UNet on B rows
    Dref = gnet(src, x, t)                   unconditional guidance net (UNet only)
    x'   = x + (t' - t) * (x - lerp(Dref, D, 1.5)) / t

Headline workload (BASELINE.json configs[1]): "vivid-base UNet, 256x256, batch 16, CFG=1.5" = the base
architecture (model_channels=128, extra_attn=1) built at img_resolution=256,
B=16 targets = 32 dual-source rows, synthetic inputs and seeded random weights, fp32-grade arithmetic.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Every rank runs its own batch (independent samples shard with no data-path collective,
generate_images.py:199-200), so scaling is weak; value = N*K steps / max-over-ranks time.
Prints ONE JSON line on rank 0.  Besides BASELINE's metric it carries, at N = 1:
  roofline          dominant kernel family of the timed region (HIP events on the launch stream), against both peak bases
  parity            HIP vs the CPU oracle on one batch-1 guided evaluation of the SAME networks (the gate BASELINE.md 4 promises)
  cpu_baseline      that oracle evaluation timed on the host cores (after one warm-up call)
  other_workloads   short runs of the same step in exact-fp32 mode, of BASELINE configs[3] (SR net built at 1024^2, B=4) and
                    configs[4] (base + depth-warp features at 256^2, B=16), and of the reference's own cascade stages (base@64 +
                    guidance at batch 32 and batch 1, SR@256 at batch 32) with `wall_over_kernel`; each bf16x3 one carries
                    `parity.batchN_vs_batch1`; `--no-extras` skips them
  sampler_runs      whole edm_sampler runs of the reference's cascade with the library's sampler-level scheduling on and off
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak; bf16x3 executes 3 MFMA products per fp32 product
PEAK_HBM_GBS = 8000.0
EVALS_PER_IMAGE_BATCH = 63        # a 32-step Heun run = 63 denoiser evaluations (generate_images.py:74,104)

WORKLOADS = {
    # name: (kind, img_resolution, batch, guided, description)
    "c2": ("base", 256, 16, True, "vivid-base arch @256x256, batch 16, CFG 1.5 (net + uncond gnet), dual-source"),
    "c4": ("sr", 1024, 4, False, "vivid-sr arch built @1024x1024 (256->1024), batch 4, no guidance (the SR stage has none), noisy_sr 0.25"),
    "c5": ("warp", 256, 16, False, "vivid-base + depth-warp Fourier features @256x256, batch 16, one net evaluation (no guidance net)"),
    "base64": ("base", 64, 16, True, "vivid-base @64x64 (the reference's own base stage), batch 16, CFG 1.5"),
    # the reference's real presets (train_nvs.py:28-30, training/training_loop.py:236): base stage at 64^2, SR stage 64 -> 256,
    # at generate_images.py's default max_batch_size of 32
    "ref_base64_b32": ("base", 64, 32, True, "reference-true base stage: vivid-base + vivid-uncond @64x64, batch 32, CFG 1.5"),
    "ref_sr256_b32": ("sr", 256, 32, False, "reference-true SR stage: vivid-sr @256x256 (64->256), batch 32, no guidance, noisy_sr 0.25"),
    "ref_base64_b1": ("base", 64, 1, True, "reference-true base stage at batch 1 (latency case): vivid-base + vivid-uncond @64x64, CFG 1.5"),
    "tiny": ("base", 64, 1, True, "vivid-base @64x64, batch 1, CFG 1.5 (plumbing)"),
}


def measured_traffic(precision):
    """HBM bytes per launch per kernel family from a committed rocprofv3 PMC summary of THIS command (profiles/, separate --pmc
    FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md).  NOT measured in this run:
    the file names the build it was collected from."""
    path = os.path.join(ROOT, "profiles", f"traffic_c2_{precision}.json")
    if not os.path.exists(path):
        return {}, None
    d = json.load(open(path))
    return {k: v["hbm_bytes_per_launch"] for k, v in d.get("families", {}).items()}, dict(file=os.path.relpath(path, ROOT), collected_from=d.get("source"))


def rho_schedule(num_steps=32, sigma_min=0.002, sigma_max=80.0, rho=7.0):
    idx = torch.arange(num_steps, dtype=torch.float32)
    t = (sigma_max ** (1 / rho) + idx / (num_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
    return torch.cat([t, torch.zeros(1)])


def make_inputs(R, B, seed, device, kind="base"):
    g = torch.Generator("cpu").manual_seed(seed)
    src = torch.rand(2 * B, 3, R, R, generator=g) * 2 - 1
    noise = torch.randn(B, 3, R, R, generator=g).repeat_interleave(2, dim=0)
    geo = torch.randn(2 * B, 20, generator=g)
    geo[:, [14, 15, 18, 19]] = 0
    cond = None
    if kind == "warp":                                   # depth U(1,5), pose from compose_geometry
        from vivid_amd.geometry import compose_geometry
        src = torch.cat([src, torch.rand(2 * B, 1, R, R, generator=g) * 4 + 1], dim=1)
        th = 0.05 * torch.randn(2 * B, generator=g)
        Rm = torch.zeros(2 * B, 3, 3)
        Rm[:, 0, 0], Rm[:, 0, 2], Rm[:, 1, 1], Rm[:, 2, 0], Rm[:, 2, 2] = th.cos(), th.sin(), 1.0, -th.sin(), th.cos()
        K = (torch.tensor([57.7, 57.7, 32.0, 32.0]) * (R / 64)).expand(2 * B, 4)
        geo = compose_geometry(torch.cat([Rm, 0.1 * torch.randn(2 * B, 3, 1, generator=g)], dim=2), K, K, imsize=R)
    if kind == "sr":                                     # bilinear-upsampled low-res image as conditioning
        low = torch.rand(B, 3, R // 4, R // 4, generator=g) * 2 - 1
        cond = torch.nn.functional.interpolate(low, size=(R, R), mode="bilinear", align_corners=False).to(device)
    return src.to(device), noise.to(device), geo.to(device), cond


def configs_for(kind, R):
    import vivid_amd
    if kind == "sr":
        return vivid_amd.vivid_sr(R), None
    if kind == "warp":
        return vivid_amd.vivid_base(R, warp_depth_coor=True), None
    return vivid_amd.vivid_base(R), vivid_amd.vivid_uncond(R)


def build_nets(kind, R, precision, dev, guided):
    import vivid_amd
    cfg, ucfg = configs_for(kind, R)
    net = vivid_amd.NVPrecond.from_config(cfg, precision=precision)
    net.load_state_dict(vivid_amd.synth_state_dict(cfg, seed=0), strict=True)
    net = net.to(dev)
    gnet = None
    if guided:
        gnet = vivid_amd.NVPrecond.from_config(ucfg, precision=precision)
        gnet.load_state_dict(vivid_amd.synth_state_dict(ucfg, seed=1), strict=True)
        gnet = gnet.to(dev)
    return net, gnet


def host_core_allotment():
    """Cores THIS job may use: the scheduler affinity mask, tightened by the cgroup CPU quota when there is one (cgroup v2 `cpu.max`,
    v1 `cpu.cfs_quota_us / cpu.cfs_period_us`).  torch's default thread count is the machine's core count (128 on the GPU boxes), not
    the job's share of it: running the oracle on that many threads oversubscribes the allotment and mis-states `cores`."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    cores = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return dict(cores=cores, affinity=aff, cgroup_quota=quota, cpu_model=model, os_cpu_count=os.cpu_count())


def oracle_guided_eval(R):
    """The CPU oracle (a restatement of the reference's PyTorch-CPU path; kind="port") on a bounded sample: ONE guided evaluation at
    batch 1 of the headline networks, run twice - an untimed warm-up (thread pools, allocator, first-touch of 1 GB of weights), then
    the timed one.  A batch-16 step is 16 such evaluations (cost is linear in batch).  Returns the outputs too: they are the parity
    reference for the HIP nets on the same inputs."""
    from oracle import vivid_ref as Rf
    import vivid_amd
    torch.set_grad_enabled(False)
    host = host_core_allotment()
    threads_default = torch.get_num_threads()
    torch.set_num_threads(host["cores"])          # the job's allotment, not the machine's core count ("core count stated")
    cfg, ucfg = vivid_amd.vivid_base(R), vivid_amd.vivid_uncond(R)
    d, ud = cfg.to_dict(), ucfg.to_dict()
    d.pop("use_fp16"); ud.pop("use_fp16")
    net = Rf.OracleNet(Rf.make_config(**d), vivid_amd.synth_state_dict(cfg, seed=0))
    gnet = Rf.OracleNet(Rf.make_config(**ud), vivid_amd.synth_state_dict(ucfg, seed=1))
    src, noise, geo, _ = make_inputs(R, 1, 1, "cpu")
    t = torch.full((2,), 5.0)
    x = noise * 5.0
    times = []
    for _ in range(2):
        t0 = time.perf_counter()
        D = net(src, x, t, geo)
        ref = gnet(src, x, t)
        guided = ref.lerp(D, 1.5)
        times.append(time.perf_counter() - t0)
    used = torch.get_num_threads()
    torch.set_num_threads(threads_default)
    return dict(seconds_b1=times[1], seconds_b1_cold=times[0], cores=used, host=host, torch_threads_default=threads_default, D=D, guided=guided,
                inputs=(src, x, t, geo))


def rel_l2(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).norm() / b.norm())


def family_table(fam, steps, ms_per_step, precision):
    kern = {}
    for k, v in fam.items():
        if v["launches"] == 0:
            continue
        s = v["ms"] / 1000.0
        kern[k] = {"ms_per_step": v["ms"] / steps, "launches_per_step": v["launches"] / steps,
                   "tflops": v["flops"] / s / 1e12 if s > 0 else 0.0, "gbs": v["bytes"] / s / 1e9 if s > 0 else 0.0,
                   "gflop_per_step": v["flops"] / steps / 1e9, "gbyte_per_step": v["bytes"] / steps / 1e9}
    return kern


def roofline_of(dom, fam, kern, ms_per_step, precision):
    """Roofline object of one kernel family: achieved = algorithmic FLOPs (or bytes) of its launches / their summed durations."""
    kd, vd = kern[dom], fam[dom]
    mfma_bound = dom in ("conv3x3", "conv1x1", "attention")
    x3 = precision == "bf16x3" and mfma_bound
    if mfma_bound:
        achieved, unit = kd["tflops"], "TFLOP/s"
        peak = PEAK_BF16_MFMA_TFLOPS / 3.0 if x3 else PEAK_FP32_MFMA_TFLOPS
    else:
        achieved, peak, unit = kd["gbs"], PEAK_HBM_GBS, "GB/s"
    out = {"kernel": dom, "bound": "mfma" if mfma_bound else "hbm", "achieved": achieved, "peak": peak, "unit": unit,
           "frac": achieved / peak, "traffic": None,
           "algorithmic_bytes_per_launch": vd["bytes"] / vd["launches"], "avg_launch_ms": vd["ms"] / vd["launches"],
           "launches": vd["launches"], "algorithmic_per_launch": (vd["flops"] if mfma_bound else vd["bytes"]) / vd["launches"],
           "share_of_step": kd["ms_per_step"] / ms_per_step}
    if mfma_bound:
        # both readings of "peak" for fp32-grade arithmetic on gfx950 (no xf32): SURVEY 8(d) prices against the fp32 matrix peak;
        # the bf16x3 path runs on the bf16 pipe with 3 executed products per algorithmic one
        out["peak_basis"] = {"bf16_mfma_div3": {"peak": PEAK_BF16_MFMA_TFLOPS / 3.0, "frac": achieved / (PEAK_BF16_MFMA_TFLOPS / 3.0)},
                             "fp32_mfma": {"peak": PEAK_FP32_MFMA_TFLOPS, "frac": achieved / PEAK_FP32_MFMA_TFLOPS},
                             "used": "bf16_mfma_div3" if x3 else "fp32_mfma"}
        out["note"] = ("achieved = algorithmic fp32 FLOPs/s; every fp32 product is 3 bf16 MFMA products (hi*hi+hi*lo+lo*hi, fp32 accumulate), "
                       "so peak = 2500 TF/s dense bf16 MFMA / 3 and the executed MFMA rate is 3 x achieved" if x3 else
                       "fp32 operands on v_mfma_f32_32x32x2_f32; peak = fp32 matrix peak (no xf32 on gfx950)")
    return out


def batch_vs_batch1(net, gnet, src, x, tt, geo, cond, picks):
    """The timed configuration against its own batch-1 evaluations (which tests/test_hip_timed_configs.py pin to the CPU oracle):
    one evaluation at the full batch, then the samples `picks` alone.  The launcher picks kernels by grid size (tile shapes, K order,
    split-K), so this is what ties the kernels that are TIMED to the parity chain.  Returns the largest rel-L2 seen."""
    saved = [(n, n.noisy_sr) for n in (net, gnet) if n is not None]
    for n, _ in saved:
        n.noisy_sr = 0.0                      # the SR net draws randn inside forward (training/models.py:658): off, so both batches see one input
    try:
        worst = 0.0
        for n in (net, gnet):
            if n is None:
                continue
            g = geo if n is net else None
            full = n(src, x, tt, g, cond)
            for i in picks:
                one = n(src[2 * i:2 * i + 2], x[2 * i:2 * i + 2], tt[2 * i:2 * i + 2], None if g is None else g[2 * i:2 * i + 2],
                        None if cond is None else cond[i:i + 1])
                worst = max(worst, rel_l2(full[i:i + 1], one))
    finally:
        for n, v in saved:
            n.noisy_sr = v
    return worst


def run_workload(name, precision, steps, warmup, dev, rank, world, profile=True, batch=None, keep_nets=False, parity_picks=None,
                 wall_probe=False):
    """Times `steps` steps of one workload after `warmup` untimed ones; returns (result dict, (net, gnet) or None).
    parity_picks: sample indices for batch_vs_batch1 (after the timed region).  wall_probe: additionally time `steps` steps with
    per-launch events OFF (`wall_ms_unprofiled`) - the latency-bound workloads are judged on wall time over summed kernel time."""
    import vivid_amd  # noqa: F401
    from vivid_amd.sampler import _context, _step, guided_denoise
    kind, R, B, guided, desc = WORKLOADS[name]
    if batch:
        B = batch
    net, gnet = build_nets(kind, R, precision, dev, guided)
    src, noise, geo, cond = make_inputs(R, B, 100 + rank, dev, kind)
    t_steps = rho_schedule()
    sctx = _context(dev)
    state = {"x": (noise * float(t_steps[0])).contiguous(), "serial": bool(profile)}

    def step(i):
        j = i % 32
        t_hat, t_next = float(t_steps[j]), float(t_steps[j + 1])
        x = state["x"]
        tt = torch.full((x.shape[0],), t_hat, device=dev)
        # (two-stream overlap of net and gnet - the library's default for small evaluations - is held off while per-launch events
        #  are on: concurrent kernels stretch each other's durations, which would spoil the kernel table)
        D, ref = guided_denoise(net, gnet, src, x, tt, geo, cond, None, 1.5 if gnet is not None else 1, overlap=False if state["serial"] else None)
        d_cur = torch.empty_like(D)
        x_next = torch.empty_like(x)
        _step(sctx, x, None, D, ref, 1.5 if gnet is not None else 1.0, d_cur, t_hat, t_next, x_next)
        if j == 31:      # t_next = 0 ends a trajectory: restart from noise
            x_next = (noise * float(t_steps[0])).contiguous()
        state["x"] = x_next

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    ctxs = [net._engine.ctx, sctx] + ([gnet._engine.ctx] if gnet is not None else [])
    if profile:
        for c in ctxs:
            c.profile_enable(True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
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
    extra = {}
    if wall_probe:
        state["serial"] = False               # library defaults: what a caller of edm_sampler gets
        for i in range(2):
            step(warmup + steps + i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(steps):
            step(warmup + steps + i)
        torch.cuda.synchronize()
        extra["wall_ms_unprofiled"] = 1000.0 * (time.perf_counter() - t1) / steps
        extra["hipgraph"] = bool(getattr(list(net._engine.programs.values())[0].plan, "graph_captured", False))
        import vivid_amd.sampler as _vs
        extra["guidance_overlap"] = bool(gnet is not None and 2 * B * R * R <= _vs.GUIDANCE_OVERLAP_MAX_PIXELS)
    if parity_picks:
        tt = torch.full((2 * B,), 5.0, device=dev)
        picks = [i for i in parity_picks if i < B]
        extra["batch_vs_batch1"] = dict(batch=B, samples=picks, rel_l2_max=batch_vs_batch1(net, gnet, src, noise * 5.0, tt, geo, cond, picks))
    res = dict(name=name, desc=desc, R=R, B=B, guided=guided, precision=precision, steps=steps, warmup=warmup, elapsed=elapsed, extra=extra,
               ms_per_step=1000.0 * elapsed / steps, fam=fam, finite=bool(torch.isfinite(state["x"]).all().item()),
               params={"net": sum(p.numel() for p in net.parameters()), **({"gnet": sum(p.numel() for p in gnet.parameters())} if gnet is not None else {})})
    if keep_nets:
        return res, (net, gnet)
    del net, gnet, src, noise, geo, cond, state
    gc.collect()
    torch.cuda.empty_cache()
    return res, None


def sampler_run(kind, R, B, num_steps, guided, precision, dev):
    """Whole edm_sampler runs (generate_images.py:43-118) with the library's defaults and with its sampler-level scheduling switched
    off: (a) encoder features computed once per noise level, ahead on a side stream (a 32-step run calls the denoiser 63 times at 32
    distinct levels; the reference re-evaluates the encoder at every call), (b) guidance net on a second stream for small batches.
    Both are exact (bit-identical samples, tests/test_hip_denoiser.py); neither touches the headline metric, whose timed step is
    always one full evaluation."""
    import vivid_amd
    net, gnet = build_nets(kind, R, precision, dev, guided)
    src, noise, geo, cond = make_inputs(R, B, 7, dev, kind)
    kw = dict(labels=geo, gnet=gnet if guided else net, conditioning_image=cond, guidance=1.5 if guided else 1)
    out = {"workload": f"edm_sampler, {num_steps} steps = {2 * num_steps - 1} denoiser calls, {'CFG 1.5' if guided else 'no guidance'}, batch {B}, {R}x{R}",
           "calls": 2 * num_steps - 1}
    for label, env in (("reference_call_pattern", {"VIVID_FEATURE_PIPELINE": "0", "VIVID_GUIDANCE_OVERLAP": "0"}), ("library_default", {})):
        saved = {k: os.environ.get(k) for k in ("VIVID_FEATURE_PIPELINE", "VIVID_GUIDANCE_OVERLAP")}
        for k in saved:
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            vivid_amd.edm_sampler(net, src, noise, num_steps=2, **kw)          # records the programs
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            x = vivid_amd.edm_sampler(net, src, noise, num_steps=num_steps, **kw)
            torch.cuda.synchronize()
            sec = time.perf_counter() - t0
        finally:
            for k, v in saved.items():
                os.environ.pop(k, None)
                if v is not None:
                    os.environ[k] = v
        out[label] = {"seconds": sec, "denoiser_calls_per_s": (2 * num_steps - 1) / sec, "images_per_s": B / sec, "finite": bool(torch.isfinite(x).all().item())}
    out["speedup"] = out["reference_call_pattern"]["seconds"] / out["library_default"]["seconds"]
    del net, gnet
    gc.collect()
    torch.cuda.empty_cache()
    return out


def summarise(res, world):
    """The compact form used for the secondary workloads."""
    out = {"workload": res["desc"], "precision": res["precision"], "dtype": "bf16x3" if res["precision"] == "bf16x3" else "f32",
           "steps": res["steps"], "warmup": res["warmup"], "ms_per_step": res["ms_per_step"], "evals_per_s": world * res["steps"] / res["elapsed"],
           "finite": res["finite"]}
    if res["fam"]:
        kern = family_table(res["fam"], res["steps"], res["ms_per_step"], res["precision"])
        dom = max(kern, key=lambda k: kern[k]["ms_per_step"])
        r = roofline_of(dom, res["fam"], kern, res["ms_per_step"], res["precision"])
        out["roofline"] = {k: r[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "share_of_step")}
        out["kernels"] = {k: {"ms_per_step": v["ms_per_step"], "tflops": v["tflops"], "gbs": v["gbs"]} for k, v in kern.items()}
        out["whole_step_tflops"] = sum(v["flops"] for v in res["fam"].values()) / res["elapsed"] / 1e12
        ksum = sum(v["ms"] for v in res["fam"].values()) / res["steps"]
        out["kernel_ms_per_step"] = ksum
        out["launches_per_step"] = sum(v["launches"] for v in res["fam"].values()) / res["steps"]
        if "wall_ms_unprofiled" in res["extra"]:
            # wall time of a step with per-launch events off over the summed kernel durations measured with them on: 1.0 = the GPU
            # never waits for the host or for a dependent launch to start
            out["wall_ms_per_step_unprofiled"] = res["extra"]["wall_ms_unprofiled"]
            out["evals_per_s_unprofiled"] = 1000.0 / res["extra"]["wall_ms_unprofiled"]
            out["wall_over_kernel"] = res["extra"]["wall_ms_unprofiled"] / ksum
            out["hipgraph_replay"] = res["extra"].get("hipgraph")
            # small guided evaluations run net and gnet on two HIP streams (vivid_amd.sampler.guided_denoise): wall can then be BELOW the summed kernel time
            out["guidance_overlap_two_streams"] = res["extra"].get("guidance_overlap")
    if "batch_vs_batch1" in res["extra"]:
        b = res["extra"]["batch_vs_batch1"]
        out["parity"] = {f"batch{b['batch']}_vs_batch1": b["rel_l2_max"], "samples": b["samples"], "tolerance": 2e-5,
                         "pass": bool(b["rel_l2_max"] < 2e-5),
                         "note": "HIP at the timed batch vs the same nets at batch 1 (pinned to the CPU oracle by tests/test_hip_timed_configs.py)"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="override per-GPU batch (non-default runs are not the headline)")
    ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "fp32"],
                    help="bf16x3: fp32 emulated by a bf16 hi/lo split on bf16 MFMA (default); fp32: exact fp32 MFMA")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle leg (also drops the parity object)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary workloads (fp32 mode, C4, C5)")
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

    torch.set_grad_enabled(False)
    for kv in filter(None, os.environ.get("VIVID_BENCH_KNOBS", "").split(",")):       # A/B runs only: "name=value,..." -> vh_set_knob
        from vivid_amd import _lib
        _lib.set_knob(kv.split("=")[0], int(kv.split("=")[1]))
    headline = args.workload == "c2" and not args.batch
    want_parity = world == 1 and not args.no_cpu_baseline and WORKLOADS[args.workload][0] == "base" and WORKLOADS[args.workload][3]
    res, nets = run_workload(args.workload, args.precision, args.steps, args.warmup, dev, rank, world, profile=not args.no_profile,
                             batch=args.batch, keep_nets=want_parity)
    R, B = res["R"], res["B"]
    out = None
    if rank == 0:
        evals_per_s = world * args.steps / res["elapsed"]
        out = {
            "metric": "denoise-steps/sec", "value": evals_per_s,
            "unit": f"guided denoiser evaluations/s (batch {B} per GPU, {R}x{R})",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16x3" if args.precision == "bf16x3" else "f32", "data": "synthetic",
            "config": {"workload": res["desc"], "img_resolution": R, "batch_per_gpu": B, "guidance": 1.5 if res["guided"] else 1.0,
                       "precision": args.precision, "global_batch": B * world,
                       "parallelism": f"dp{world} (independent samples per rank, no collective)", "params": res["params"]},
            # denoise-steps/sec := denoiser evaluations/s; a 32-step Heun sampler run makes 63 of them per image batch
            "sampler_steps_per_s": evals_per_s / EVALS_PER_IMAGE_BATCH * 32,
            "images_per_s": evals_per_s / EVALS_PER_IMAGE_BATCH * B,
            "finite": res["finite"],
        }
        fam = res["fam"]
        if fam:
            kern = family_table(fam, args.steps, res["ms_per_step"], args.precision)
            dom = max(kern, key=lambda k: kern[k]["ms_per_step"])
            out["roofline"] = roofline_of(dom, fam, kern, res["ms_per_step"], args.precision)
            traffic, tsrc = measured_traffic(args.precision) if headline else ({}, None)
            tr = traffic.get(dom)
            if tr is not None:
                # bytes per launch come from a committed PMC profile (not from this run); the rate divides them by THIS run's launch time
                out["roofline"]["traffic"] = tr
                out["roofline"]["traffic_source"] = dict(tsrc, measured_in_this_run=False)
                out["roofline"]["traffic_over_algorithmic"] = tr / out["roofline"]["algorithmic_bytes_per_launch"]
                out["roofline"]["hbm_gbs_from_profile_bytes"] = tr / (out["roofline"]["avg_launch_ms"] / 1e3) / 1e9
            if "attention" in kern and dom != "attention":       # north_star also asks for the attention kernels' MFMA utilisation
                out["roofline_attention"] = roofline_of("attention", fam, kern, res["ms_per_step"], args.precision)
                out["roofline_attention"]["traffic"] = traffic.get("attention")
            out["kernels"] = kern
            out["whole_step_tflops"] = sum(v["flops"] for v in fam.values()) / res["elapsed"] / 1e12

    if want_parity and rank == 0:
        # CPU oracle on one batch-1 guided evaluation: the reported CPU baseline AND the parity reference for the HIP nets
        net, gnet = nets
        cb = oracle_guided_eval(R)
        src, x, t, geo = (v.to(dev) for v in cb["inputs"])
        D = net(src, x, t, geo)
        Dg = gnet(src, x, t)
        guided = Dg.lerp(D, 1.5)
        out["parity"] = {"rel_l2_D": rel_l2(D.cpu(), cb["D"]), "rel_l2_guided": rel_l2(guided.cpu(), cb["guided"]), "tolerance": 1e-3,
                         "against": "oracle/vivid_ref.py (CPU, fp32) on the same weights and inputs, batch 1, sigma 5, guidance 1.5",
                         "pass": bool(rel_l2(guided.cpu(), cb["guided"]) < 1e-3)}
        # the TIMED configuration: one evaluation at the timed batch whose sample 0 is the oracle's input; samples 0 / middle / last
        # against their own batch-1 evaluations (different tile shapes, K order and split-K at the two grid sizes)
        srcB, noiseB, geoB, _ = make_inputs(R, B, 100, dev)
        xB = noiseB * 5.0
        srcB[:2], xB[:2], geoB[:2] = src, x, geo
        tB = torch.full((2 * B,), 5.0, device=dev)
        DB, GB = net(srcB, xB, tB, geoB), gnet(srcB, xB, tB)
        out["parity"][f"batch{B}_sample0_vs_oracle"] = rel_l2(GB[:1].lerp(DB[:1], 1.5).cpu(), cb["guided"])
        out["parity"][f"batch{B}_vs_batch1"] = max(rel_l2(DB[:1], D), rel_l2(GB[:1], Dg),
                                                   batch_vs_batch1(net, gnet, srcB, xB, tB, geoB, None, [B // 2 - 1, B - 1] if B > 2 else []))
        out["parity"]["batch_samples"] = sorted({0, B // 2 - 1, B - 1}) if B > 2 else [0]
        out["parity"]["pass"] = bool(out["parity"]["pass"] and out["parity"][f"batch{B}_sample0_vs_oracle"] < 1e-3
                                     and out["parity"][f"batch{B}_vs_batch1"] < 2e-5)
        del srcB, noiseB, geoB, xB, DB, GB
        out["cpu_baseline"] = {"value": 1.0 / (B * cb["seconds_b1"]), "unit": out["unit"], "cores": cb["cores"], "kind": "port",
                               "cpu_model": cb["host"]["cpu_model"], "sched_affinity_cores": cb["host"]["affinity"],
                               "cgroup_cpu_quota": cb["host"]["cgroup_quota"], "os_cpu_count": cb["host"]["os_cpu_count"],
                               "torch_threads_default": cb["torch_threads_default"],
                               "sample": f"one guided evaluation (net + uncond gnet) of the same {R}x{R} networks at batch 1 on the CPU oracle "
                                         f"took {cb['seconds_b1']:.2f} s after one untimed warm-up call ({cb['seconds_b1_cold']:.2f} s cold); "
                                         f"a batch-{B} step is {B} of them (cost linear in batch)"}
    nets = None
    gc.collect()
    torch.cuda.empty_cache()

    if world == 1 and headline and not args.no_extras and rank == 0:
        extras = {}
        for key, (wl, prec, k, w, picks, probe) in {
                "c2_fp32": ("c2", "fp32", 2, 1, None, False),
                "c4_sr1024_b4": ("c4", "bf16x3", 3, 1, (0, 3), False),
                "c5_warp256_b16": ("c5", "bf16x3", 3, 1, (0, 7, 15), False),
                # the reference's own presets: latency-sensitive shapes, judged on wall time over summed kernel time as well
                "ref_base64_b32": ("ref_base64_b32", "bf16x3", 10, 3, (0, 15, 31), True),
                "ref_sr256_b32": ("ref_sr256_b32", "bf16x3", 5, 2, (0, 31), True),
                "ref_base64_b1": ("ref_base64_b1", "bf16x3", 20, 5, None, True)}.items():
            r2, _ = run_workload(wl, prec, k, w, dev, rank, world, profile=not args.no_profile, parity_picks=picks, wall_probe=probe)
            extras[key] = summarise(r2, world)
        out["other_workloads"] = extras
        # whole sampler runs of the reference's own cascade (base@64 with guidance, 32 steps; SR@256, 16 steps; generate_images.py:305-327)
        out["sampler_runs"] = {
            "ref_base64_b32": sampler_run("base", 64, 32, 32, True, "bf16x3", dev), "ref_sr256_b32": sampler_run("sr", 256, 32, 16, False, "bf16x3", dev),
            "ref_base64_b1": sampler_run("base", 64, 1, 32, True, "bf16x3", dev), "ref_sr256_b1": sampler_run("sr", 256, 1, 16, False, "bf16x3", dev),
            "c2_base256_b16_8steps": sampler_run("base", 256, 16, 8, True, "bf16x3", dev)}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
