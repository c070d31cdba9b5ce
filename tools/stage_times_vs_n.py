"""Stage times of config 3s scene truncated / repeated to N Gaussians: the part of every stage that does not depend on the scene
(diagnostic; needs a GPU):  python tools/stage_times_vs_n.py"""
import importlib, sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench
gs = importlib.import_module(bench.PKG); ops = importlib.import_module(bench.PKG + ".ops")
dev = torch.device("cuda:0")
params, cam = bench.synthetic_scene(3)
full = {k: torch.cat([v, v + 0.0], 0).to(dev) for k, v in params.items()}     # 2 M
gimg = torch.rand(cam["H"], cam["W"], 3, device=dev)
for n in (250_000, 500_000, 786_432, 983_040, 1_000_000, 1_179_648, 1_500_000, 2_000_000):
    p = {k: full[k][:n].clone().requires_grad_(True) for k in bench.NAMES}
    args = [p[k] for k in bench.NAMES] + [torch.eye(4, device=dev), cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"]]
    gs.render_gaussians(*args).backward(gimg)
    t = ops.StageTimer()
    with ops.deferred_checks() as chk:
        for it in range(13):
            if it == 3: ops.set_stage_timer(t)
            for q in p.values(): q.grad = None
            gs.render_gaussians(*args).backward(gimg)
        torch.cuda.synchronize(); ops.set_stage_timer(None)
    chk.verify()
    tot = t.totals_ms()
    print(n, "waves", (n + 63) // 64, "  ".join(f"{k} {v[1] / v[0] * 1e3:.1f}" for k, v in tot.items()), flush=True)
