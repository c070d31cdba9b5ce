#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: rendered Mpix/s (forward + backward), 1M Gaussians @ 1920x1080, SH degree 3.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

One step = one pass of the hot path on one camera view per rank: fused projection (covariance build + SH folded in),
binning + sort, raster forward, raster backward, projection backward, with the upstream gradient
dL/dimage = rand(H, W, 3; seed 1); at N > 1 followed by the RCCL all-reduce of the six parameter
gradients (data parallel by camera view).  The scene is the synthetic config-3 scene of SURVEY.md §8d
(seed 0), inputs resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      the dominant kernel's algorithmic HBM bytes per launch / its average duration (HIP events recorded on the
                launch stream inside the timed region) against the 8 TB/s HBM peak; `traffic` from profiles/ PMC data
                if a matching file exists, else null.
  cpu_baseline  the CPU oracle (oracle/torch_port.py, the reference's PyTorch CPU path restated) timed on this box's
                host cores on a bounded sample (rank 0, N = 1 only).  A reported baseline, not the target.
"""
import argparse
import importlib
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "3d-gaussian-splatting-for-novel-view-synthesis_amd"

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)

# config id -> (N, H, W, fx, mu_s)      SURVEY.md §8(d)
CONFIGS = {1: (10_000, 256, 256, 300.0, -3.0), 2: (100_000, 800, 800, 800.0, -4.5), 3: (1_000_000, 1080, 1920, 1100.0, -5.0),
           4: (3_000_000, 1080, 1920, 1100.0, -5.4), 5: (10_000_000, 2160, 3840, 2200.0, -5.8)}
NAMES = ("pos", "f_dc", "f_rest", "opacity_raw", "scale_raw", "q_raw")


def synthetic_scene(config):
    """seed 0, draws in the order pos, scale, quat, opacity, f_dc, f_rest; camera at the origin."""
    N, H, W, fx, mu_s = CONFIGS[config]
    g = torch.Generator().manual_seed(0)
    pos = torch.randn(N, 3, generator=g)
    pos[:, 2] += 5.0
    scale_raw = torch.randn(N, 3, generator=g) * 0.3 + mu_s
    q_raw = torch.randn(N, 4, generator=g)
    opacity_raw = torch.randn(N, generator=g)
    f_dc = torch.randn(N, 3, generator=g)
    f_rest = torch.randn(N, 45, generator=g) * 0.1
    if config == 1:
        f_rest = torch.zeros(N, 45)
    return dict(pos=pos, scale_raw=scale_raw, q_raw=q_raw, opacity_raw=opacity_raw, f_dc=f_dc, f_rest=f_rest), \
        dict(H=H, W=W, fx=fx, fy=fx, cx=W / 2.0, cy=H / 2.0)


def orbit_c2w(k, n_views=8, centre=(0.0, 0.0, 5.0)):
    """View k: the identity camera rotated about the scene centre by k * 360 / n_views degrees (k = 0: identity)."""
    a = 2.0 * math.pi * k / n_views
    R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    c = np.asarray(centre)
    m = np.eye(4)
    m[:3, :3] = R
    m[:3, 3] = c - R @ c
    return torch.tensor(m, dtype=torch.float32)


# Algorithmic HBM bytes per launch of each stage (fp32 SoA figures of SURVEY.md §8d, split per kernel):
#   project          16 N + 220 V (read) + 48 V (write)
#   bin              8 P (write keys) + 8 P (read sorted)
#   raster_forward   4 P (ids) + 40 P (records gathered: 36 B used of 48) + 12 HW (image)        -> 44 P + 12 HW
#   raster_backward  20 HW (dL/dO + saved per-pixel state) + 44 P (ids + records) + 36 P (2D grads) -> 80 P + 20 HW
#   project_backward 272 V + 236 N
def algorithmic_bytes(stage, N, V, P, HW):
    return {"project": 16 * N + 268 * V, "bin": 16 * P, "raster_forward": 44 * P + 12 * HW,
            "raster_backward": 80 * P + 20 * HW, "project_backward": 272 * V + 236 * N}[stage]


KERNEL_OF_STAGE = {"project": "project_kernel+bin_count_kernel+colour_kernel", "bin": "bin_scatter_kernel+split_*_kernel+list_sort_kernel",
                   "raster_forward": "raster_forward_kernel", "raster_backward": "raster_backward_kernel",
                   "project_backward": "project_backward_kernel"}


def pmc_traffic(kernel):
    """HBM bytes per launch from a committed PMC profile (profiles/pmc_traffic.json), if present for this kernel."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        v = d.get(kernel)
        return float(v["hbm_bytes_per_launch"]) if v else None
    except (OSError, ValueError, KeyError, TypeError):
        return None


def host_cores(cap=16):
    """Cores this process may really use: cgroup quota if set, else the affinity mask; capped at the GPU box's
    per-GPU CPU share (16) so that torch does not oversubscribe a container that sees the whole host."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


class _Budget:
    """SIGALRM guard: the CPU baseline must never take the benchmark line down."""

    def __init__(self, seconds):
        self.seconds = seconds

    def __enter__(self):
        import signal

        def on_alarm(signum, frame):
            raise TimeoutError(f"CPU baseline exceeded its {self.seconds}s budget")
        self.old = signal.signal(signal.SIGALRM, on_alarm)
        signal.alarm(self.seconds)

    def __exit__(self, *exc):
        import signal
        signal.alarm(0)
        signal.signal(signal.SIGALRM, self.old)
        return False


def cpu_baseline(config, sample_rows=64, sample_cols=None):
    """Time the CPU oracle (PyTorch restatement of the reference path, autograd backward) on a bounded sample of the
    benchmark workload: the same scene, forward + backward, image cropped to a centred window (principal point
    shifted accordingly).  Every one of the N Gaussians still goes through covariance build, SH and culling."""
    from oracle import torch_port as tp      # the oracle is only the baseline being timed here, never the product
    params, cam = synthetic_scene(config)
    H, W = cam["H"], cam["W"]
    ch = min(sample_rows, H)
    cw = W if sample_cols is None else min(sample_cols, W)
    y0, x0 = (H - ch) // 2, (W - cw) // 2
    cores = host_cores()
    torch.set_num_threads(cores)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    gimg = torch.rand(ch, cw, 3, generator=torch.Generator().manual_seed(1))
    t0 = time.perf_counter()
    img = tp.render_fused(p["pos"], p["f_dc"], p["f_rest"], p["opacity_raw"], p["scale_raw"], p["q_raw"], torch.eye(4),
                          ch, cw, cam["fx"], cam["fy"], cam["cx"] - x0, cam["cy"] - y0)
    t1 = time.perf_counter()
    img.backward(gimg)
    t2 = time.perf_counter()
    return {"value": ch * cw / (t2 - t0) / 1e6, "unit": "Mpix/s", "cores": cores, "kind": "port",
            "sample": f"oracle/torch_port.py fwd+bwd, config {config} scene (all {len(params['pos'])} Gaussians), "
                      f"centred {cw}x{ch} crop of the {W}x{H} image; fwd {t1 - t0:.1f}s bwd {t2 - t1:.1f}s",
            "seconds": t2 - t0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=3, help="synthetic scene of SURVEY.md §8d (default 3: 1M @ 1080p)")
    ap.add_argument("--forward-only", action="store_true", help="time forward-only inference instead of fwd+bwd")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=256)
    ap.add_argument("--cpu-budget", type=int, default=150, help="seconds allowed for the CPU baseline leg")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--exchange", default="factored", choices=("factored", "allreduce"),
                    help="N > 1: SH gradients as logit gradients + local rebuild (DESIGN.md §7), or one all-reduce of all six tensors")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # CPU baseline first (rank 0, N = 1 only), before anything touches the GPU, under a time budget
    cb = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        try:
            with _Budget(args.cpu_budget):
                cb = cpu_baseline(args.config, sample_rows=args.cpu_rows)
        except Exception as e:
            cb = {"value": None, "unit": "Mpix/s", "cores": host_cores(), "kind": "port",
                  "sample": f"failed: {type(e).__name__}: {e}"}
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    dev = torch.device("cuda", 0 if args.single_device else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    gs = importlib.import_module(PKG)
    ops = importlib.import_module(PKG + ".ops")
    dp = importlib.import_module(PKG + ".dp")

    params_cpu, cam = synthetic_scene(args.config)
    N = params_cpu["pos"].shape[0]
    H, W = cam["H"], cam["W"]
    need_grad = not args.forward_only
    params = {k: params_cpu[k].to(dev).requires_grad_(need_grad) for k in NAMES}
    c2w = orbit_c2w(rank % 8).to(dev)                       # data parallel by camera view: rank r renders view r
    gimg = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)
    cam_args = (H, W, cam["fx"], cam["fy"], cam["cx"], cam["cy"])

    info = {"allreduce": None}

    def step():
        if need_grad:
            for p in params.values():
                p.grad = None
            if world > 1 and args.exchange == "factored":
                with dp.FactoredExchange(params, world_views=world) as ex:
                    img = gs.render_gaussians(*[params[k] for k in NAMES], c2w, *cam_args)
                    img.backward(gimg)
                ex.finish()
                info["allreduce"] = "factored: all-reduce of pos/opacity/scale/rotation gradients (44 B per Gaussian) + all-gather " \
                                    "of colour-logit gradients (12 B per Gaussian and view) + local SH rebuild"
            else:
                img = gs.render_gaussians(*[params[k] for k in NAMES], c2w, *cam_args)
                img.backward(gimg)
            if world > 1 and args.exchange == "allreduce":
                grads = [params[k].grad for k in NAMES]
                info["allreduce"] = "one flat buffer, single collective" if dp._common_base(grads) is not None else "2 buckets"
                dp.allreduce_gradients(grads, world_views=world)
        else:
            with torch.no_grad():
                img = gs.render_gaussians(*[params[k] for k in NAMES], c2w, *cam_args)
        return img

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    stats = gs.render_stats()
    # Untimed calibration pass with an event pair around EVERY library call: per-stage breakdown, and which call dominates.
    # (Each event pair costs ~10 us of stream time, so the timed region below only brackets the dominant call.)
    cal = ops.StageTimer()
    ops.set_stage_timer(cal)
    fence()
    for _ in range(3):
        step()
    fence()
    ops.set_stage_timer(None)
    cal_stage = cal.totals_ms()
    dom = max(cal_stage, key=lambda k: cal_stage[k][1])
    timer = ops.StageTimer(only=[dom])
    ops.set_stage_timer(timer)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    ops.set_stage_timer(None)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        _, V, P = stats
        HW = H * W
        ms = elapsed / args.steps * 1e3
        value = world * HW * args.steps / elapsed / 1e6
        stage = dict(cal_stage)
        stage.update(timer.totals_ms())              # the dominant call: measured live inside the timed region
        per_stage = {k: {"launches": n, "avg_ms": t / n, "alg_bytes": algorithmic_bytes(k, N, V, P, HW),
                         "gbs": algorithmic_bytes(k, N, V, P, HW) / (t / n * 1e-3) / 1e9,
                         "timed_region": k == dom} for k, (n, t) in stage.items()}
        ach = per_stage[dom]["gbs"]
        fwd_b = 16 * N + 268 * V + 52 * P + 12 * HW
        bwd_b = 236 * N + 272 * V + 80 * P + 20 * HW
        alg_total = fwd_b + (bwd_b if need_grad else 0)
        out = {
            "metric": "rendered Mpix/s (fwd+bwd), 1M Gaussians @1080p SH3" if need_grad and args.config == 3 else
                      f"rendered Mpix/s ({'fwd+bwd' if need_grad else 'forward only'}), config {args.config}",
            "value": value, "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"SURVEY §8d config {args.config}: {N} Gaussians, {W}x{H}, SH degree 3, "
                                   f"{'forward+backward' if need_grad else 'forward only'}, one camera view per GPU per step"
                                   + (", gradient exchange over RCCL" if world > 1 and need_grad else ""),
                       "N": N, "V": V, "P": P, "tiles": math.ceil(H / 16) * math.ceil(W / 16),
                       "parallelism": f"dp{world} by camera view", "allreduce": info["allreduce"]},
            "fps": world * args.steps / elapsed,
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(KERNEL_OF_STAGE[dom]), "kernel": KERNEL_OF_STAGE[dom],
                         "avg_launch_ms": per_stage[dom]["avg_ms"], "alg_bytes_per_launch": per_stage[dom]["alg_bytes"],
                         "note": "the raster kernels are VALU-bound, not HBM-bound: SQ_ACTIVE_INST_VALU busy 87 % (forward) / 98 % "
                                 "(backward) of the kernel's cycles, profiles/r01_v14_sq_counters.txt; DESIGN.md section 6"},
            "pipeline_roofline": {"alg_bytes_per_step": alg_total, "achieved": alg_total / (ms * 1e-3) / 1e9,
                                  "frac": alg_total / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "unit": "GB/s"},
            "stages": per_stage,
        }
        if cb is not None:
            out["cpu_baseline"] = cb
            if cb.get("value"):
                out["gpu_over_cpu"] = value / cb["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
