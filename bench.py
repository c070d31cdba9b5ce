#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: rendered Mpix/s (forward + backward), 1M Gaussians @ 1920x1080, SH degree 3.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Either an external launcher started the ranks (`python -m torch.distributed.run ...
bench.py --gpus N`, WORLD_SIZE set), or bench.py starts them itself: called without a launcher it re-runs itself under
`torch.distributed.run` as a CHILD process, before this process has made any GPU call, and exits with the child's code.
Fewer visible GPUs than N is an error (exit code 2), never a silent 1-GPU run.

One step = one pass of the hot path on one camera view per rank: fused projection (covariance build + SH folded in),
binning + sort, raster forward, raster backward, projection backward, with the upstream gradient
dL/dimage = rand(H, W, 3; seed 1); at N > 1 followed by the exchange of the parameter gradients over
RCCL (data parallel by camera view).  The scene is the synthetic config-3 scene of SURVEY.md 8d (seed 0),
inputs resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      the dominant kernel's algorithmic HBM bytes per launch / its average duration (HIP events recorded on the
                launch stream inside the timed region) against the 8 TB/s HBM peak; `traffic` from profiles/ PMC data.
  cpu_baseline  the CPU oracle (oracle/torch_port.py, the reference's PyTorch CPU path restated) timed on this box's
                host cores on a bounded sample: a CROP of the frame (rank 0, N = 1 only).  A reported baseline, not the target.
  sustained     >= 1 s of back-to-back steps after the timed region: ms/step mean, min and max over 10 windows.
  three_call_sequence  the same step through the reference's own three calls (build_sigma_from_params, evaluate_sh, render).
  waiting_path  the same step through the call that waits for every frame's pair count (the reference-style call).
  N = 1 extras  (outside the timed region; --no-extras skips them) forward_only (the reference's FPS protocol,
                scripts/render_trained.py:319-381: frame by frame, and software-pipelined render_frames), train_step
                (render + L1/SSIM loss + backward + clip + Adam, scripts/train.py:446-569), config4 (3 M Gaussians, a training iteration
                over 8 views on this one GPU), config5 (10 M Gaussians, 4K).
  N > 1         exchange: compute_ms (the same step without the exchange), exchange_ms (step - compute = exposed exchange
                time), exchange_alone_ms (the collectives by themselves), bytes all-reduced / all-gathered per step, nranks.
"""
import argparse
import importlib
import json
import math
import os
import socket
import subprocess
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
           4: (3_000_000, 1080, 1920, 1100.0, -5.4), 5: (10_000_000, 2160, 3840, 2200.0, -5.8),
           # 6 is not a BASELINE.json config: config 2's scene with the footprints of a TRAINED scene (log-scale mean -2.0 instead of
           # -4.5: radii of ~75 px instead of ~6; every Gaussian covers hundreds of lists and the lists saturate early)
           6: (100_000, 800, 800, 800.0, -2.0)}
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


def pmc_traffic(kernel, config=3):
    """HBM bytes per launch from a committed PMC profile of THIS scene (profiles/pmc_traffic_config<N>.json: separate rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE passes, tools/profile_all.sh), if present for this kernel; else null."""
    path = os.path.join(ROOT, "profiles", f"pmc_traffic_config{config}.json")
    try:
        with open(path) as f:
            d = json.load(f)
        v = d.get(kernel)
        if v is None:                     # summaries made since round 2 keep template arguments: raster_backward_kernel<false>
            hits = [x for k, x in d.items() if k.split("<")[0] == kernel and isinstance(x, dict)]
            v = max(hits, key=lambda x: x.get("hbm_bytes_per_launch", 0.0)) if hits else None
        return float(v["hbm_bytes_per_launch"]) if v else None
    except (OSError, ValueError, KeyError, TypeError):
        return None


def valu_figures(kernel, config=3):
    """SURVEY.md 8(d)'s second roofline figure for the raster kernels: VALU work, from COMMITTED profiles of this scene --
    profiles/valu_config<N>.json (tools/valu_summary.py over `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE`, tools/sq_counters.sh)
    and profiles/r03_raster_sim_config<N>.json (tools/raster_sim.py: the loop iterations and the (pixel, Gaussian) evaluations that land
    inside an ellipse, counted by replaying the kernels' queue construction on the CPU).  None when the files are absent."""
    try:
        with open(os.path.join(ROOT, "profiles", f"valu_config{config}.json")) as f:
            v = json.load(f)
        k = next((x for name, x in v.items() if name.split("<")[0] == kernel and isinstance(x, dict)), None)
        if k is None:
            return None
        out = {"kernel": kernel, "valu_wave_instructions_per_launch": k["insts_valu"], "valu_busy": k["busy"], "source": v.get("_source")}
        try:
            with open(os.path.join(ROOT, "profiles", f"r03_raster_sim_config{config}.json")) as f:
                sim = json.load(f)
            out.update({"loop_iterations_per_launch": sim["iterations"], "useful_lane_fraction": sim["useful_lane_fraction"],
                        "queue_imbalance": sim["queue_imbalance"],
                        "valu_wave_instructions_per_iteration": k["insts_valu"] / sim["iterations"],
                        "valu_lane_instructions_per_composited_pixel_gaussian": k["insts_valu"] * 64 / sim["inside_evaluations"]})
        except (OSError, ValueError, KeyError):
            pass
        return out
    except (OSError, ValueError, KeyError, StopIteration):
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
    out = {"value": ch * cw / (t2 - t0) / 1e6, "unit": "Mpix/s", "cores": cores, "kind": "port",
           "sample": f"oracle/torch_port.py fwd+bwd, config {config} scene (all {len(params['pos'])} Gaussians), "
                     f"centred {cw}x{ch} CROP of the {W}x{H} image; fwd {t1 - t0:.1f}s bwd {t2 - t1:.1f}s",
           "seconds": t2 - t0}
    # the FULL frame (SURVEY 8d asks for it "when RAM allows"): the PyTorch restatement needs 45 GB and minutes for it; the plain-C
    # restatement of the same algorithm (oracle/gs_oracle.c: float64, OpenMP, per-tile lists and front-to-back compositing with the
    # reference's early exit) does it in seconds -- a second figure, not the reference's own speed
    try:
        os.environ.setdefault("OMP_NUM_THREADS", str(cores))
        from oracle import c_oracle
        s_np = {k: v.numpy() for k, v in params.items()}
        s_np["c2w"] = np.eye(4)
        gfull = np.random.default_rng(1).uniform(0, 1, (H, W, 3))
        t3 = time.perf_counter()
        st, _, _, (V, P) = c_oracle.render(s_np, H, W, cam["fx"], cam["fy"], cam["cx"], cam["cy"], grad_image=gfull)
        t4 = time.perf_counter()
        if st == 0:
            out["full_frame_c"] = {"value": H * W / (t4 - t3) / 1e6, "unit": "Mpix/s", "cores": cores, "kind": "port (plain C, float64, OpenMP)",
                                   "sample": f"oracle/gs_oracle.c fwd+bwd, the whole {W}x{H} frame, V={V} P={P}", "seconds": t4 - t3}
    except Exception as e:                                      # never take the line down
        out["full_frame_c"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """No launcher started us but --gpus N > 1: start the N ranks as children (torch.distributed.run).  This process has made
    no GPU call (torch.cuda.device_count() does not initialise the device on ROCm), so nothing GPU-initialised is re-executed."""
    rehearsal = args.single_device or args.launch_check
    if not rehearsal:
        visible = torch.cuda.device_count()
        if visible < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {visible} GPU(s) are visible: refusing to run (a 1-GPU run "
                             f"must not be reported as {args.gpus})\n")
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def launch_check(args, rank, world):
    """Plumbing check without a GPU (tests): the ranks rendezvous, all-reduce one number and report the group's size."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(args.backend if args.backend != "nccl" or torch.cuda.is_available() else "gloo")
    t = torch.ones(1)
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "nranks": dist.get_world_size(), "sum": float(t.item()),
                          "backend": dist.get_backend()}))
    dist.barrier()
    dist.destroy_process_group()
    return 0


def host_queue_ms(render_pass, ops, fence, steps=100):
    """Milliseconds of HOST time to queue one step (Python + autograd + the library calls), measured with the GPU idle at the start and
    never waited for inside the loop (fewer steps than a DeferredChecks block holds before it looks at its oldest frames).  When this
    exceeds the GPU's time per step, the host is the bound."""
    fence()
    with ops.deferred_checks() as chk:
        t0 = time.perf_counter()
        for _ in range(steps):
            render_pass()
        t1 = time.perf_counter()
        fence()
    chk.verify()
    return (t1 - t0) / steps * 1e3


def timed(fn, steps, fence):
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    fence()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=3, help="synthetic scene of SURVEY.md §8d (default 3: 1M @ 1080p)")
    ap.add_argument("--forward-only", action="store_true", help="time forward-only inference instead of fwd+bwd")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip forward_only / train_step / config5 / sustained")
    ap.add_argument("--cpu-rows", type=int, default=256)
    ap.add_argument("--cpu-budget", type=int, default=150, help="seconds allowed for the CPU baseline leg")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--exchange", default="factored", choices=("factored", "allreduce"),
                    help="N > 1: SH gradients as logit gradients + local rebuild (DESIGN.md §7), or one all-reduce of all six tensors")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--launch-check", action="store_true", help="only check the rank launch + process group (no GPU needed)")
    ap.add_argument("--deterministic", action="store_true", help="bitwise reproducible gradients (ops.set_deterministic): rows per "
                    "(list, Gaussian) pair stored and added in a fixed order instead of float atomics")
    ap.add_argument("--views-per-rank", type=int, default=1, help="camera views every rank renders per step (one gradient exchange per step: "
                    "the exchange amortises over the views; BASELINE.json's config 4 is 1 view per GPU)")
    ap.add_argument("--separate-calls", action="store_true", help="ablation: a deferred frame goes through the separate library calls "
                    "(gsplat_project, gsplat_bin, ...) instead of the two composite entries")
    ap.add_argument("--wait-counts", action="store_true", help="every forward pass waits for its pair count (exact buffers) instead of "
                    "sizing them from earlier frames (ops.deferred_checks)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.launch_check:
        raise SystemExit(launch_check(args, rank, world))
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
    if not args.single_device and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) are visible")
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
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")

    gs = importlib.import_module(PKG)
    ops = importlib.import_module(PKG + ".ops")
    dp = importlib.import_module(PKG + ".dp")

    if args.deterministic:
        ops.set_deterministic(True)
    if args.separate_calls:
        ops._composite = False
    params_cpu, cam = synthetic_scene(args.config)
    N = params_cpu["pos"].shape[0]
    H, W = cam["H"], cam["W"]
    need_grad = not args.forward_only
    params = {k: params_cpu[k].to(dev).requires_grad_(need_grad) for k in NAMES}
    del params_cpu
    vpr = max(1, args.views_per_rank)
    c2ws = [orbit_c2w((rank * vpr + i) % 8).to(dev) for i in range(vpr)]      # data parallel by camera view: rank r renders views r V .. r V + V - 1
    c2w = c2ws[0]
    gimg = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)
    cam_args = (H, W, cam["fx"], cam["fy"], cam["cx"], cam["cy"])

    info = {}

    def checked(fn):
        # the way training.Trainer.step renders: no wait for the frame's pair count (buffers sized from earlier frames, SH colour
        # inside the projection kernel); the frame's checks (off-screen exception, buffer capacity) are made inside the step, once
        # (ops.run_deferred: the pass is repeated if a frame outgrew the buffers).  --wait-counts: the reference-style call that
        # waits for the counters in the middle of the forward pass.
        return fn() if args.wait_counts else ops.run_deferred(fn)

    def render_pass():
        if need_grad:
            for p in params.values():
                p.grad = None
            for c in c2ws:                                   # (gradients of the rank's views accumulate in .grad)
                gs.render_gaussians(*[params[k] for k in NAMES], c, *cam_args).backward(gimg)
        else:
            with torch.no_grad():
                for c in c2ws:
                    gs.render_gaussians(*[params[k] for k in NAMES], c, *cam_args)

    def local_step():
        checked(render_pass)

    def step(mode=None):
        mode = mode or args.exchange
        if not need_grad or world == 1:
            return local_step()
        if mode == "factored":
            def factored_pass():
                for p in params.values():
                    p.grad = None
                with dp.FactoredExchange(params, world_views=world * vpr) as ex:
                    for c in c2ws:                           # view k's logit gradients are gathered while view k + 1 renders
                        gs.render_gaussians(*[params[k] for k in NAMES], c, *cam_args).backward(gimg)
                return ex
            # (a repeat would re-issue this rank's collectives: it cannot happen in the timed region -- every rank renders the same view
            # in every step, and the warm-up steps left it the capacity of that view; training.Trainer.step, where views change, agrees
            # on a repeat across the ranks)
            ex = checked(factored_pass)
            ex.finish()
            info[mode] = "factored: all-reduce of pos/opacity/scale/rotation gradients (44 B per Gaussian) + all-gather " \
                                "of colour-logit gradients (12 B per Gaussian and view) + local SH rebuild"
        else:
            checked(render_pass)
            grads = [params[k].grad for k in NAMES]
            info[mode] = "all-reduce of all six gradient tensors (236 B per Gaussian): " + \
                         ("one flat buffer, single collective" if dp._common_base(grads) is not None else "2 buckets")
            dp.allreduce_gradients(grads, world_views=world * vpr)

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for _ in range(args.warmup):
        step()
    stats = gs.render_stats()
    p_binned = ops.binned_pairs()            # (list, Gaussian) pairs really binned on this scene (the extras render other scenes later)
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
    # the dominant call is bracketed in every 8th step of the timed region (an event pair costs ~10 us of stream time and, with the
    # composite entries, one more library call: the steps in between run exactly as the product does)
    timer = ops.StageTimer(only=[dom], every=8)
    ops.set_stage_timer(timer)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    ops.set_stage_timer(None)
    elapsed = max_over_ranks(elapsed)
    ms = elapsed / args.steps * 1e3

    extras = {}
    if not args.no_extras:
        # sustained: >= 1 s of back-to-back steps, ten windows
        per = max(2, int(math.ceil(100.0 / ms)))
        wins = [max_over_ranks(timed(step, per, fence)) for _ in range(10)]
        extras["sustained"] = {"seconds": sum(wins) * per / 1e3, "steps": 10 * per, "ms_per_step": sum(wins) / 10, "min_ms": min(wins),
                               "max_ms": max(wins), "windows": 10}
    if not args.wait_counts:
        extras["host_queue_ms"] = min(host_queue_ms(render_pass, ops, fence) for _ in range(3))
        extras["host_calls_per_step"] = {"forward": "gsplat_forward_deferred (1 call, 1 arena)", "backward": "gsplat_backward (1 call)"} \
            if ops._composite else {"forward": "gsplat_project + gsplat_bin + gsplat_rasterize_forward", "backward": "gsplat_rasterize_backward + gsplat_project_backward"}
    if not args.no_extras and not args.wait_counts and world == 1:
        # the same step through the call that WAITS for every frame's pair count in the middle of the forward pass (exact buffers,
        # exceptions raised by the call itself: what a caller that only switched the import gets)
        def waiting_step():
            render_pass()
        for _ in range(3):
            waiting_step()
        wms = timed(waiting_step, max(20, args.steps // 4), fence)
        extras["waiting_path"] = {"what": "render_gaussians() outside deferred_checks(): the host waits for the frame's counters",
                                  "ms_per_step": wms, "mpix_per_s": H * W / (wms * 1e-3) / 1e6}
    if world > 1 and need_grad:
        # the same step without the exchange, and the collectives alone, on buffers of the step's sizes
        compute_ms = max_over_ranks(timed(local_step, args.steps, fence))
        small = torch.zeros(11 * N + 64 * 4, dtype=torch.float32, device=dev)        # pos 3 + opacity 1 + scale 3 + quat 4 floats
        full = torch.zeros(59 * N + 64 * 6, dtype=torch.float32, device=dev) if args.exchange == "allreduce" else None
        logit = torch.zeros(N, 3, dtype=torch.float32, device=dev)
        gathered = torch.empty(world * N, 3, dtype=torch.float32, device=dev)

        def collectives():
            if args.exchange == "factored":
                ws = [dist.all_gather_into_tensor(gathered, logit, async_op=True) for _ in range(vpr)]
                ws.append(dist.all_reduce(small, async_op=True))
                for w_ in ws:
                    w_.wait()
            else:
                dist.all_reduce(full)
        alone_ms = None
        if args.backend == "nccl":
            for _ in range(2):
                collectives()
            alone_ms = max_over_ranks(timed(collectives, args.steps, fence))
        # the other form of the exchange, same step (both are built; the first run on real links decides which is the default)
        alt = "allreduce" if args.exchange == "factored" else "factored"
        alt_ms = alt_err = None
        try:
            for _ in range(2):
                step(alt)
            alt_ms = max_over_ranks(timed(lambda: step(alt), args.steps, fence))
        except Exception as e:                           # never take the headline line down
            alt_err = f"{type(e).__name__}: {e}"
        extras["exchange"] = {
            "mode": args.exchange, "nranks": dist.get_world_size(), "backend": dist.get_backend(),
            "other_mode": {"mode": alt, "what": info.get(alt), "step_ms": alt_ms, "error": alt_err},
            "step_ms": ms, "compute_ms": compute_ms, "exchange_ms": ms - compute_ms, "exchange_alone_ms": alone_ms,
            "allreduce_bytes_per_step": (44 if args.exchange == "factored" else 236) * N,
            "allgather_bytes_per_rank_per_step": 12 * N * vpr if args.exchange == "factored" else 0,
            "allgather_bytes_received_per_step": 12 * N * world * vpr if args.exchange == "factored" else 0,
            "note": "exchange_ms = step - compute: the part of the exchange the step does not hide; the SH rebuild "
                    "(gsplat_sh_accumulate) counts as exchange"}
    if world == 1 and rank == 0 and not args.no_extras and args.config == 3 and need_grad:
        extras.update(single_gpu_extras(gs, ops, params, cam, cam_args, dev, fence))

    if rank == 0:
        _, V, P = stats
        HW = H * W
        value = world * vpr * HW * args.steps / elapsed / 1e6
        stage = dict(cal_stage)
        stage.update(timer.totals_ms())              # the dominant call: measured live inside the timed region
        per_stage = {k: {"launches": n, "avg_ms": t / n, "alg_bytes": algorithmic_bytes(k, N, V, P, HW),
                         "gbs": algorithmic_bytes(k, N, V, P, HW) / (t / n * 1e-3) / 1e9,
                         "timed_region": k == dom} for k, (n, t) in stage.items()}
        ach = per_stage[dom]["gbs"]
        fwd_b = 16 * N + 268 * V + 52 * P + 12 * HW
        bwd_b = 236 * N + 272 * V + 80 * P + 20 * HW
        alg_total = (fwd_b + (bwd_b if need_grad else 0)) * vpr
        out = {
            "metric": "rendered Mpix/s (fwd+bwd), 1M Gaussians @1080p SH3" if need_grad and args.config == 3 else
                      f"rendered Mpix/s ({'fwd+bwd' if need_grad else 'forward only'}), config {args.config}",
            "value": value, "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"SURVEY §8d config {args.config}: {N} Gaussians, {W}x{H}, SH degree 3, "
                                   f"{'forward+backward' if need_grad else 'forward only'}, {vpr if vpr > 1 else 'one'} camera view{'s' if vpr > 1 else ''} per GPU per step"
                                   + ((", gradient exchange over " + ("RCCL" if args.backend == "nccl" else args.backend + " (rehearsal)"))
                                      if world > 1 and need_grad else ""),
                       "N": N, "V": V, "P": P, "P_binned": p_binned,
                       "tiles": math.ceil(H / 16) * math.ceil(W / 16),
                       "parallelism": f"dp{world} by camera view", "allreduce": info.get(args.exchange),
                       "counts": "waited for in every forward pass" if args.wait_counts else
                                 "not waited for: buffers from earlier frames, checks once per step (ops.deferred_checks, as Trainer.step)"},
            "fps": world * vpr * args.steps / elapsed, "views_per_rank": vpr,
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(KERNEL_OF_STAGE[dom], args.config), "kernel": KERNEL_OF_STAGE[dom],
                         "avg_launch_ms": per_stage[dom]["avg_ms"], "alg_bytes_per_launch": per_stage[dom]["alg_bytes"],
                         "note": "the raster kernels are VALU-bound, not HBM-bound (DESIGN.md section 6): the HBM fraction of this "
                                 "kernel is small by construction; pipeline_roofline prices the whole step"},
            "valu": valu_figures(KERNEL_OF_STAGE[dom], args.config),
            "pipeline_roofline": {"alg_bytes_per_step": alg_total, "achieved": alg_total / (ms * 1e-3) / 1e9,
                                  "frac": alg_total / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "unit": "GB/s"},
            "stages": per_stage,
        }
        out.update(extras)
        if cb is not None:
            out["cpu_baseline"] = cb
            if cb.get("value"):
                out["gpu_over_cpu"] = value / cb["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def single_gpu_extras(gs, ops, params, cam, cam_args, dev, fence):
    """N = 1, config 3, after the headline region: the other numbers BASELINE.json's configs name, each measured here so that
    the driver's line carries them (they do not enter `value`)."""
    out = {}
    H, W = cam["H"], cam["W"]
    plist = [params[k] for k in NAMES]
    # -- forward only (config 3 of BASELINE.json; the reference's protocol: scripts/render_trained.py:319-381)
    cams = [orbit_c2w(k % 8).to(dev) for k in range(32)]
    with torch.no_grad():
        def frame_by_frame():
            for c in cams:
                gs.render_gaussians(*plist, c, *cam_args)

        def pipelined():
            gs.render_frames(*plist, cams, *cam_args, on_frame=lambda k, im: None)
        frame_by_frame()
        ms_seq = timed(frame_by_frame, 2, fence) / len(cams)
        pipelined()
        ms_pipe = timed(pipelined, 2, fence) / len(cams)
    out["forward_only"] = {"workload": "config 3, forward only, 32 orbit views (8 distinct poses)", "frame_by_frame_ms": ms_seq,
                           "frame_by_frame_fps": 1e3 / ms_seq, "render_frames_ms": ms_pipe, "render_frames_fps": 1e3 / ms_pipe,
                           "mpix_per_s": H * W / (ms_pipe * 1e-3) / 1e6}
    # -- the reference's OWN call sequence through the drop-in interface (scripts/train.py:463,502,505-508): build_sigma_from_params +
    #    evaluate_sh + render, autograd through all three, every call waiting for its frame's counters like the reference's code would --
    #    what a caller gets who only switched the import
    try:
        gimg3 = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)
        eye3 = torch.eye(4, device=dev)

        def three_call():
            for p in params.values():
                p.grad = None
            sigma = gs.build_sigma_from_params(params["scale_raw"], params["q_raw"])
            color = gs.evaluate_sh(params["f_dc"], params["f_rest"], params["pos"], eye3)
            gs.render(params["pos"], color, params["opacity_raw"], sigma, eye3, *cam_args).backward(gimg3)
        if all(p.requires_grad for p in params.values()):
            for _ in range(3):
                three_call()
            ms3 = timed(three_call, 20, fence)
            out["three_call_sequence"] = {"workload": "config 3, forward + backward through gaussian_splatting's own three calls "
                                                      "(build_sigma_from_params, evaluate_sh, render), each waiting for the frame's counters",
                                          "ms_per_step": ms3, "mpix_per_s": H * W / (ms3 * 1e-3) / 1e6}
        for p in params.values():
            p.grad = None
    except Exception as e:
        out["three_call_sequence"] = {"error": f"{type(e).__name__}: {e}"}
    # -- one training iteration (scripts/train.py:446-569): render + L1/SSIM loss + backward + clip + Adam, one view
    model_mod = importlib.import_module(PKG + ".model")
    training = importlib.import_module(PKG + ".training")
    model = model_mod.GaussianModel({k: params[k].detach() for k in NAMES}, device=dev)
    trainer = training.Trainer(model, training.TrainConfig())
    view = {"image": torch.rand(H, W, 3, generator=torch.Generator().manual_seed(2)).to(dev), "c2w": cams[0], "H": H, "W": W,
            "fx": cam["fx"], "fy": cam["fy"], "cx": cam["cx"], "cy": cam["cy"]}
    it = [1]

    def train_step():
        trainer.step(it[0], [view])            # iterations 1, 2, ...: no densification (every 100), no opacity reset
        it[0] += 1
    for _ in range(3):
        train_step()
    out["train_step"] = {"workload": "config 3, Trainer.step: fused render + fused L1/SSIM loss + backward (the Adam step of f_rest inside it) + clip + fused Adam, one view",
                         "ms": timed(train_step, 20, fence)}
    del trainer, model
    # -- config 4 of BASELINE.json on ONE GPU: the 3 M-Gaussian scene, a training iteration over 8 orbit views (what 8 ranks with one
    #    view each do between two exchanges, here one after the other)
    try:
        p4_cpu, cam4 = synthetic_scene(4)
        model4 = model_mod.GaussianModel({k: p4_cpu[k] for k in NAMES}, device=dev)
        del p4_cpu
        tr4 = training.Trainer(model4, training.TrainConfig())
        g4 = torch.Generator().manual_seed(3)
        views4 = [{"image": torch.rand(cam4["H"], cam4["W"], 3, generator=g4).to(dev), "c2w": orbit_c2w(k).to(dev), "H": cam4["H"],
                   "W": cam4["W"], "fx": cam4["fx"], "fy": cam4["fy"], "cx": cam4["cx"], "cy": cam4["cy"]} for k in range(8)]
        it4 = [1]

        def train4():
            tr4.step(it4[0], views4)
            it4[0] += 1
        for _ in range(2):
            train4()
        ms4 = timed(train4, 5, fence)
        out["config4"] = {"workload": "config 4: 3 M Gaussians, 1920x1080, Trainer.step over 8 orbit views on one GPU (render + loss + backward per "
                                      "view, then clip + Adam)", "ms_per_iteration": ms4, "ms_per_view": ms4 / 8,
                          "mpix_per_s": 8 * cam4["H"] * cam4["W"] / (ms4 * 1e-3) / 1e6}
        del tr4, model4, views4
    except Exception as e:
        out["config4"] = {"error": f"{type(e).__name__}: {e}"}
    # -- footprints of a trained scene (config 6 above: config 2's scene at log-scale mean -2.0): where the front end, not the rasterizer, is the bound
    try:
        p6_cpu, cam6 = synthetic_scene(6)
        p6 = {k: p6_cpu[k].to(dev).requires_grad_(True) for k in NAMES}
        H6, W6 = cam6["H"], cam6["W"]
        g6 = torch.rand(H6, W6, 3, generator=torch.Generator().manual_seed(1)).to(dev)
        a6 = (H6, W6, cam6["fx"], cam6["fy"], cam6["cx"], cam6["cy"])
        eye6 = torch.eye(4, device=dev)

        def pass6():
            for p in p6.values():
                p.grad = None
            gs.render_gaussians(*[p6[k] for k in NAMES], eye6, *a6).backward(g6)

        def step6():
            ops.run_deferred(pass6)
        for _ in range(3):
            step6()
        _, V6, P6 = gs.render_stats()
        pb6 = ops.binned_pairs()
        cal = ops.StageTimer()
        ops.set_stage_timer(cal)
        for _ in range(3):
            step6()
        fence()
        ops.set_stage_timer(None)
        st6 = {k: t / n for k, (n, t) in cal.totals_ms().items()}
        ms6 = timed(step6, 10, fence)
        out["big_footprint"] = {"workload": "config 2's scene (100 k Gaussians, 800x800, SH 3) with log-scale mean -2.0: footprints of ~75 px, forward + backward",
                                "ms_per_step": ms6, "mpix_per_s": H6 * W6 / (ms6 * 1e-3) / 1e6, "V": V6, "P": P6, "P_binned": pb6, "stage_ms": st6,
                                "front_end_ms": st6.get("project", 0.0) + st6.get("bin", 0.0)}
        del p6, g6
    except Exception as e:
        out["big_footprint"] = {"error": f"{type(e).__name__}: {e}"}
    # -- config 5 (10 M Gaussians, 3840 x 2160, forward + backward)
    for p in params.values():
        p.grad = None
    try:
        p5_cpu, cam5 = synthetic_scene(5)
        p5 = {k: p5_cpu[k].to(dev).requires_grad_(True) for k in NAMES}
        del p5_cpu
        H5, W5 = cam5["H"], cam5["W"]
        g5 = torch.rand(H5, W5, 3, generator=torch.Generator().manual_seed(1)).to(dev)
        eye = torch.eye(4, device=dev)
        a5 = (H5, W5, cam5["fx"], cam5["fy"], cam5["cx"], cam5["cy"])

        def pass5():
            for p in p5.values():
                p.grad = None
            gs.render_gaussians(*[p5[k] for k in NAMES], eye, *a5).backward(g5)

        def step5():
            ops.run_deferred(pass5)
        for _ in range(2):
            step5()
        _, V5, P5 = gs.render_stats()
        cal = ops.StageTimer()
        ops.set_stage_timer(cal)
        for _ in range(2):
            step5()
        fence()
        ops.set_stage_timer(None)
        st = {k: t / n for k, (n, t) in cal.totals_ms().items()}
        ms5 = timed(step5, 5, fence)
        N5, HW5 = p5["pos"].shape[0], H5 * W5
        dom5 = max(st, key=st.get)
        alg5 = (16 + 236) * N5 + (268 + 272) * V5 + 132 * P5 + 32 * HW5
        out["config5"] = {"workload": "config 5: 10 M Gaussians, 3840x2160, SH 3, forward + backward", "ms_per_step": ms5,
                          "mpix_per_s": HW5 / (ms5 * 1e-3) / 1e6, "V": V5, "P": P5, "stage_ms": st,
                          "dominant": {"kernel": KERNEL_OF_STAGE[dom5], "avg_ms": st[dom5],
                                       "frac": algorithmic_bytes(dom5, N5, V5, P5, HW5) / (st[dom5] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "traffic": pmc_traffic(KERNEL_OF_STAGE[dom5], 5)},
                          "pipeline_frac": alg5 / (ms5 * 1e-3) / 1e9 / HBM_PEAK_GBS}
    except Exception as e:                        # never take the headline line down
        out["config5"] = {"error": f"{type(e).__name__}: {e}"}
    return out


if __name__ == "__main__":
    main()
