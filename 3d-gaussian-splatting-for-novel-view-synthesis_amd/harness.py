"""Benchmark / render harness and checkpoint formats of the reference ("next" row 3).

    create_orbit_trajectory(center, radius=3.0, num_frames=60, elevation=0.0)       scripts/render_trained.py:28-75
    save_checkpoint / load_checkpoint (dict with iteration + the six tensors)         scripts/train.py:197-219
    save_parameter_files (pos_{it}.pt ... q_rot_{it}.pt)                              scripts/train.py:590-597
    load_parameters (checkpoint file -> loose files -> latest checkpoint -> error)    scripts/render_trained.py:116-182
    benchmark_orbit / format_report (1 warm-up frame, synchronize + wall clock per frame, mean/median/min/max/std ms and
                                     FPS)                                            scripts/render_trained.py:319-381
Host-side Python only (file formats and timing protocol); rendering goes through the fused HIP path.
"""
import time
from pathlib import Path

import numpy as np
import torch

from . import ops

PARAM_KEYS = ("pos", "opacity_raw", "f_dc", "f_rest", "scale_raw", "q_raw")
_FILE_STEM = {"pos": "pos", "opacity_raw": "opacity_raw", "f_dc": "f_dc", "f_rest": "f_rest", "scale_raw": "scale_raw",
              "q_raw": "q_rot"}          # the reference saves q_raw as q_rot_{iteration}.pt


def create_orbit_trajectory(center, radius=3.0, num_frames=60, elevation=0.0):
    """Circular orbit around `center` in the z-up world of the reference: camera-to-world matrices [num_frames, 4, 4]
    with columns (right, -up, forward, position)."""
    center = np.asarray(center, dtype=np.float64)
    ang = 2.0 * np.pi * np.arange(num_frames) / num_frames
    cam = center + np.stack([radius * np.cos(ang), radius * np.sin(ang), np.full(num_frames, radius * np.sin(elevation))], 1)
    fwd = center - cam
    fwd = fwd / (np.linalg.norm(fwd, axis=1, keepdims=True) + 1e-8)
    right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
    right = right / (np.linalg.norm(right, axis=1, keepdims=True) + 1e-8)
    up = np.cross(right, fwd)
    up = up / (np.linalg.norm(up, axis=1, keepdims=True) + 1e-8)
    c2w = np.tile(np.eye(4), (num_frames, 1, 1))
    c2w[:, :3, 0], c2w[:, :3, 1], c2w[:, :3, 2], c2w[:, :3, 3] = right, -up, fwd, cam
    return c2w


def _as_dict(model):
    return {k: (model[k] if isinstance(model, dict) else getattr(model, k)) for k in PARAM_KEYS}


def save_checkpoint(path, model, iteration):
    """checkpoint_*.pt exactly as GaussianModel.save_checkpoint writes it (scripts/train.py:197-208)."""
    p = _as_dict(model)
    ck = {'iteration': iteration}
    ck.update({k: p[k].detach().cpu() for k in PARAM_KEYS})
    torch.save(ck, path)


def save_parameter_files(output_dir, model, iteration):
    """The six loose tensors the reference writes next to every checkpoint (scripts/train.py:590-597)."""
    p = _as_dict(model)
    out = Path(output_dir)
    for k in PARAM_KEYS:
        torch.save(p[k].detach().cpu(), out / f"{_FILE_STEM[k]}_{iteration}.pt")


def load_checkpoint(path, device="cuda"):
    """-> (dict of the six tensors on `device`, iteration).  Uses the safe loader (tensors and plain Python values)."""
    ck = torch.load(path, map_location="cpu", weights_only=True)
    return {k: ck[k].detach().to(device) for k in PARAM_KEYS}, ck.get('iteration', 0)


def load_parameters(checkpoint_dir, iteration='final', device="cuda"):
    """The lookup order of scripts/render_trained.py:116-182: checkpoint file, then the six loose files, then the latest
    checkpoint in the directory, else FileNotFoundError."""
    d = Path(checkpoint_dir)
    if iteration == 'final':
        path, suffix = d / 'checkpoint_final.pt', 'final'
    else:
        path, suffix = d / f'checkpoint_{int(iteration):06d}.pt', str(int(iteration))
    if path.exists():
        return load_checkpoint(path, device)[0]
    loose = {k: d / f"{_FILE_STEM[k]}_{suffix}.pt" for k in PARAM_KEYS}
    if all(f.exists() for f in loose.values()):
        return {k: torch.load(f, map_location="cpu", weights_only=True).detach().to(device) for k, f in loose.items()}
    available = sorted(d.glob('checkpoint_*.pt'))
    if available:
        return load_checkpoint(available[-1], device)[0]
    raise FileNotFoundError(f"Could not find checkpoint files for iteration {iteration} in {d} "
                            f"(looked for {path.name} and {', '.join(f.name for f in loose.values())})")


def benchmark_orbit(params, c2ws, H, W, fx, fy, cx, cy, fused=True, on_frame=None):
    """Per-frame render times over a trajectory with the reference's protocol: one un-timed warm-up frame, then for each
    frame synchronize -> wall clock -> (SH + render) -> synchronize -> wall clock.  `fused=False` issues the reference's
    own call sequence (evaluate_sh + render with a pre-built sigma, covariance build outside the timed region)."""
    dev = params["pos"].device
    kw = dict(pix_guard=32, chi_square_clip=6.25, alpha_cutoff=1 / 128.)
    sigma = None if fused else ops.build_sigma_from_params(params["scale_raw"], params["q_raw"])

    def frame(c2w):
        if fused:
            return ops.render_gaussians(params["pos"], params["f_dc"], params["f_rest"], params["opacity_raw"], params["scale_raw"],
                                        params["q_raw"], c2w, H, W, fx, fy, cx, cy, **kw)
        col = ops.evaluate_sh(params["f_dc"], params["f_rest"], params["pos"], c2w)
        return ops.render(params["pos"], col, params["opacity_raw"], sigma, c2w, H, W, fx, fy, cx, cy, **kw)

    times = []
    with torch.no_grad():
        cams = [torch.as_tensor(np.asarray(c), dtype=torch.float32, device=dev) for c in c2ws]
        if cams:
            frame(cams[0])
            torch.cuda.synchronize(dev)
        for i, c2w in enumerate(cams):
            torch.cuda.synchronize(dev)
            t0 = time.time()
            img = frame(c2w)
            torch.cuda.synchronize(dev)
            times.append(time.time() - t0)
            if on_frame is not None:
                on_frame(i, img)
    t = np.asarray(times)
    fps = 1.0 / t
    return {"frames": len(t), "mean_ms": t.mean() * 1e3, "median_ms": float(np.median(t)) * 1e3, "min_ms": t.min() * 1e3,
            "max_ms": t.max() * 1e3, "std_ms": t.std() * 1e3, "fps_mean": fps.mean(), "fps_median": float(np.median(fps)),
            "fps_min": fps.min(), "fps_max": fps.max(), "times": t}


def throughput_orbit(params, c2ws, H, W, fx, fy, cx, cy, on_frame=None):
    """Frames per second over a trajectory when frames need not be timed one by one: ops.render_frames pipelines them over
    two streams (frame k + 1's projection / binning overlaps frame k's rasterisation).  Not the reference's protocol
    (benchmark_orbit is): a serving-style number."""
    dev = params["pos"].device
    args = (params["pos"], params["f_dc"], params["f_rest"], params["opacity_raw"], params["scale_raw"], params["q_raw"])
    cams = [torch.as_tensor(np.asarray(c), dtype=torch.float32, device=dev) for c in c2ws]
    ops.render_frames(*args, cams[:2], H, W, fx, fy, cx, cy, on_frame=lambda k, im: None)       # warm-up
    torch.cuda.synchronize(dev)
    t0 = time.time()
    ops.render_frames(*args, cams, H, W, fx, fy, cx, cy, on_frame=on_frame or (lambda k, im: None))
    torch.cuda.synchronize(dev)
    dt = time.time() - t0
    return {"frames": len(cams), "seconds": dt, "fps": len(cams) / dt if dt > 0 else float("inf")}


def format_report(stats, H, W, n_gaussians, scale_factor=1.0):
    """The metrics block scripts/render_trained.py:361-381 prints."""
    bar = "=" * 60
    return "\n".join([
        "", bar, "RENDERING PERFORMANCE METRICS", bar, f"Resolution: {W}x{H} (scale_factor={scale_factor})",
        f"Number of Gaussians: {n_gaussians:,}", f"Number of frames rendered: {stats['frames']}", "", "Render Time per Frame:",
        f"  Mean:   {stats['mean_ms']:.2f} ms", f"  Median: {stats['median_ms']:.2f} ms", f"  Min:    {stats['min_ms']:.2f} ms",
        f"  Max:    {stats['max_ms']:.2f} ms", f"  Std:    {stats['std_ms']:.2f} ms", "", "FPS (Frames Per Second):",
        f"  Mean:   {stats['fps_mean']:.2f} FPS", f"  Median: {stats['fps_median']:.2f} FPS", f"  Min:    {stats['fps_min']:.2f} FPS",
        f"  Max:    {stats['fps_max']:.2f} FPS", bar])
