#!/usr/bin/env python3
"""Per-stage times of the fused forward+backward on a benchmark scene (HIP events around every library call).
    [GSPLAT_MI355X_LIB=other.so] python tools/time_stages.py [config] [iterations]"""
import importlib
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench

PKG = "3d-gaussian-splatting-for-novel-view-synthesis_amd"
gs = importlib.import_module(PKG)
ops = importlib.import_module(PKG + ".ops")
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
deferred = "--deferred" in sys.argv          # no wait for the counters (ops.deferred_checks), checks once at the end
params, cam = bench.synthetic_scene(cfg)
dev = torch.device("cuda:0")
p = {k: v.to(dev).requires_grad_(True) for k, v in params.items()}
args = [p[k] for k in bench.NAMES] + [torch.eye(4, device=dev), cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"]]
gimg = torch.rand(cam["H"], cam["W"], 3, device=dev)
import contextlib
import time
t = ops.StageTimer()
gs.render_gaussians(*args).backward(gimg)          # learns the pair capacity
with (ops.deferred_checks() if deferred else contextlib.nullcontext()) as chk:
    for it in range(iters + 3):
        if it == 3:
            ops.set_stage_timer(t)
        for q in p.values():
            q.grad = None
        gs.render_gaussians(*args).backward(gimg)
    torch.cuda.synchronize()
    ops.set_stage_timer(None)
    t0 = time.perf_counter()
    for it in range(iters * 3):
        for q in p.values():
            q.grad = None
        gs.render_gaussians(*args).backward(gimg)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / (iters * 3) * 1e3
if deferred:
    chk.verify()
tot = t.totals_ms()
print("deferred" if deferred else "waiting ", "  ".join(f"{k} {v[1] / v[0] * 1e3:.1f}" for k, v in tot.items()), " | sum",
      f"{sum(v[1] / v[0] for v in tot.values()) * 1e3:.1f} us | wall {wall:.3f} ms/step without event pairs")
