#!/usr/bin/env python3
"""Diagnostic: time raster_backward with atomics and/or the wave reduction disabled (results are wrong in those runs)."""
import importlib, sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench
PKG = "3d-gaussian-splatting-for-novel-view-synthesis_amd"
gs = importlib.import_module(PKG); ops = importlib.import_module(PKG + ".ops"); abi = importlib.import_module(PKG + "._abi")
lib = abi.lib()
params, cam = bench.synthetic_scene(3)
dev = torch.device("cuda:0")
p = {k: v.to(dev).requires_grad_(True) for k, v in params.items()}
args = [p[k] for k in bench.NAMES] + [torch.eye(4, device=dev), cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"]]
gimg = torch.rand(cam["H"], cam["W"], 3, device=dev)
for bits, name in ((0, "full"), (1, "no atomics"), (2, "no reduction"), (3, "neither")):
    lib.gsplat_debug_set_ablation(bits)
    t = ops.StageTimer(); 
    for it in range(8):
        if it == 3: ops.set_stage_timer(t)
        for q in p.values(): q.grad = None
        gs.render_gaussians(*args).backward(gimg)
    torch.cuda.synchronize(); ops.set_stage_timer(None)
    tot = t.totals_ms()
    print(f"{name:14s} raster_backward {tot['raster_backward'][1] / tot['raster_backward'][0] * 1e3:7.1f} us   raster_forward {tot['raster_forward'][1] / tot['raster_forward'][0] * 1e3:7.1f} us")
lib.gsplat_debug_set_ablation(0)
