#!/usr/bin/env python3
"""Host-side cost of one forward+backward on the smallest benchmark scene (config 1), where the GPU work is tiny: the
latency floor of a step (~0.3 ms) and a cProfile of where the Python time goes (diagnostic; needs a GPU)."""
import sys, time, cProfile, pstats, torch
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import bench, gsplat_amd as gs
params, cam = bench.synthetic_scene(1)
dev = torch.device("cuda:0")
p = {k: v.to(dev).requires_grad_(True) for k, v in params.items()}
c2w = torch.eye(4, device=dev)
gimg = torch.rand(cam["H"], cam["W"], 3, device=dev)
cargs = (cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
def step():
    for t in p.values(): t.grad = None
    img = gs.render_gaussians(*[p[k] for k in bench.NAMES], c2w, *cargs)
    img.backward(gimg)
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 200 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
