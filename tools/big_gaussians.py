import importlib, sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
gs = importlib.import_module("3d-gaussian-splatting-for-novel-view-synthesis_amd")
ops = importlib.import_module("3d-gaussian-splatting-for-novel-view-synthesis_amd.ops")
dev = torch.device("cuda:0")
for mu in (-4.5, -3.5, -3.0, -2.5, -2.0):
    bench.CONFIGS[9] = (100_000, 800, 800, 800.0, mu)
    params, cam = bench.synthetic_scene(9)
    p = {k: v.to(dev).requires_grad_(True) for k, v in params.items()}
    args = [p[k] for k in bench.NAMES] + [torch.eye(4, device=dev), cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"]]
    gimg = torch.rand(cam["H"], cam["W"], 3, device=dev)
    def one():
        for q in p.values(): q.grad = None
        gs.render_gaussians(*args).backward(gimg)
    one(); stats = gs.render_stats()
    t = ops.StageTimer(); ops.set_stage_timer(t)
    for _ in range(3): gs.run_deferred(one)
    torch.cuda.synchronize(); ops.set_stage_timer(None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): gs.run_deferred(one)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
    tot = t.totals_ms()
    print(f"mu_s {mu}: V {stats[1]} P {stats[2]}  step {dt:.3f} ms  " + "  ".join(f"{k} {v[1] / v[0] * 1e3:.0f}" for k, v in tot.items()), "finite", all(bool(torch.isfinite(q.grad).all()) for q in p.values()))
