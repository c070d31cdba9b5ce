#!/usr/bin/env python3
"""Time the reference's own three-call sequence (build_sigma_from_params + evaluate_sh + render, autograd through all three)
against the fused entry on the benchmark scene (diagnostic; needs a GPU):  python tools/unfused_step.py [config]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench
import gsplat_amd as gs

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
params, cam = bench.synthetic_scene(cfg)
dev = torch.device("cuda:0")
p = {k: v.to(dev).requires_grad_(True) for k, v in params.items()}
H, W = cam["H"], cam["W"]
c2w = torch.eye(4, device=dev)
gimg = torch.rand(H, W, 3, device=dev)
cargs = (H, W, cam["fx"], cam["fy"], cam["cx"], cam["cy"])


def three_call():
    sigma = gs.build_sigma_from_params(p["scale_raw"], p["q_raw"])
    color = gs.evaluate_sh(p["f_dc"], p["f_rest"], p["pos"], c2w)
    return gs.render(p["pos"], color, p["opacity_raw"], sigma, c2w, *cargs)


def fused():
    return gs.render_gaussians(p["pos"], p["f_dc"], p["f_rest"], p["opacity_raw"], p["scale_raw"], p["q_raw"], c2w, *cargs)


for name, fn in (("three-call", three_call), ("fused", fused)):
    for _ in range(3):
        for t in p.values():
            t.grad = None
        fn().backward(gimg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        for t in p.values():
            t.grad = None
        fn().backward(gimg)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per forward+backward")
