"""Next row 2 end to end: training.Trainer.step against the same iteration written with the oracle
(oracle/torch_port.py render + loss, float64) and torch.optim.Adam / clip_grad_norm_ at the reference's settings
(scripts/train.py:394-401, 446-569)."""
import datetime
import importlib

import numpy as np
import pytest
import torch

from oracle import scenes
from oracle import torch_port as tp

PKG = "3d-gaussian-splatting-for-novel-view-synthesis_amd"
NAMES = ("pos", "opacity_raw", "f_dc", "f_rest", "scale_raw", "q_raw")
pytestmark = pytest.mark.gpu


def _scene():
    s = scenes.case_g1()
    rng = np.random.default_rng(5)
    cams = [s["c2w"], scenes._camera(rng)]
    targets = [rng.uniform(0, 1, (s["H"], s["W"], 3)).astype(np.float32) for _ in cams]
    views = [dict(image=t, c2w=c, H=s["H"], W=s["W"], fx=s["fx"], fy=s["fy"], cx=s["cx"], cy=s["cy"]) for t, c in zip(targets, cams)]
    return s, views


def _oracle_iteration(P, opt, views, pos_lr):
    opt.param_groups[0]['lr'] = pos_lr
    opt.zero_grad()
    total = 0
    for v in views:
        img = tp.render_fused(P["pos"], P["f_dc"], P["f_rest"], P["opacity_raw"], P["scale_raw"], P["q_raw"],
                              torch.tensor(v["c2w"], dtype=torch.float64), v["H"], v["W"], v["fx"], v["fy"], v["cx"], v["cy"])
        loss = tp.compute_loss(img, torch.tensor(v["image"], dtype=torch.float64), 0.8, 0.2)[0]
        total = total + loss / len(views)
    total.backward()
    grads = {k: P[k].grad.clone() for k in NAMES}
    torch.nn.utils.clip_grad_norm_(P["pos"], max_norm=1.0)
    opt.step()
    return float(total), grads


def test_training_iterations_match_the_oracle_loop():
    model_mod = importlib.import_module(PKG + ".model")
    training = importlib.import_module(PKG + ".training")
    optim = importlib.import_module(PKG + ".optim")
    s, views = _scene()
    init = {k: torch.tensor(s[k]) for k in NAMES}
    model = model_mod.GaussianModel(init, device="cuda:0")
    # no densification / reset in this test: pure optimisation
    cfg = training.TrainConfig(densify_until_iter=0, opacity_reset_interval=10 ** 9)
    tr = training.Trainer(model, cfg)
    P = {k: torch.nn.Parameter(init[k].double()) for k in NAMES}

    class M:
        pass
    mm = M()
    for k in NAMES:
        setattr(mm, k, P[k])
    opt = torch.optim.Adam(optim.reference_param_groups(mm), lr=0.01, eps=1e-15)
    for it in (1, 2, 3):                       # iteration 0 would also trigger nothing here, but keep clear of the modulo rules
        before = {k: getattr(model, k).detach().cpu().double() for k in NAMES}
        out = tr.step(it, views)
        ref_loss, ref_grads = _oracle_iteration(P, opt, views, optim.position_lr(it))
        assert abs(float(out["loss"]) - ref_loss) <= 2e-5 * abs(ref_loss), (it, float(out["loss"]), ref_loss)
        assert out["gaussians"] == len(s["pos"]) and not out["densified"]
        if it == 1:
            # first Adam step = lr * sign(g) wherever |g| >> eps: compare where the reference gradient is well above fp32 noise
            for k, lr in zip(NAMES, (optim.position_lr(1), 0.05, 0.0025, 0.0025 / 20, 0.005, 0.001)):
                g = ref_grads[k]
                solid = g.abs() > 1e-4 * g.abs().max()
                moved = getattr(model, k).detach().cpu().double() - before[k]
                ref_moved = P[k].detach() - before[k]
                assert solid.float().mean() > 0.2, k
                err = (moved - ref_moved)[solid].abs().max()
                assert err <= 2e-3 * lr + 2.4e-7 * max(1.0, float(before[k].abs().max())), (k, float(err), lr)


def test_loss_goes_down_and_densification_keeps_training():
    model_mod = importlib.import_module(PKG + ".model")
    training = importlib.import_module(PKG + ".training")
    gs = importlib.import_module(PKG)
    s, views = _scene()
    dev = "cuda:0"
    truth = {k: torch.tensor(s[k], device=dev) for k in NAMES}
    with torch.no_grad():                      # targets = renders of the true scene; start from a perturbed copy
        for v in views:
            v["image"] = gs.render_gaussians(truth["pos"], truth["f_dc"], truth["f_rest"], truth["opacity_raw"], truth["scale_raw"],
                                             truth["q_raw"], torch.tensor(v["c2w"], device=dev), v["H"], v["W"], v["fx"], v["fy"],
                                             v["cx"], v["cy"]).cpu().numpy()
    g = torch.Generator().manual_seed(3)
    init = {k: torch.tensor(s[k]) for k in NAMES}
    init["f_dc"] = init["f_dc"] + 0.5 * torch.randn(init["f_dc"].shape, generator=g)
    init["opacity_raw"] = init["opacity_raw"] - 0.5
    model = model_mod.GaussianModel(init, device=dev)
    cfg = training.TrainConfig(densification_interval=10, densify_until_iter=25, opacity_reset_interval=15, max_grad=1e-4)
    tr = training.Trainer(model, cfg)
    losses, counts, densified = [], [], []
    for it in range(40):
        out = tr.step(it, views)
        losses.append(float(out["loss"]))
        counts.append(out["gaussians"])
        densified.append(out["densified"])
    assert all(np.isfinite(losses))
    assert densified[0] and densified[10] and densified[20] and not densified[30] and sum(densified) == 3
    assert len(set(counts)) > 1                       # the model was actually resized
    for k in NAMES:
        p = getattr(model, k)
        assert p.shape[0] == counts[-1] and torch.isfinite(p).all()
    assert np.mean(losses[-5:]) < 0.8 * np.mean(losses[:5]), (losses[:5], losses[-5:])


def _dp_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    model_mod = importlib.import_module(PKG + ".model")
    training = importlib.import_module(PKG + ".training")
    s, views = _scene()
    model = model_mod.GaussianModel({k: torch.tensor(s[k]) for k in NAMES}, device="cuda:0")
    tr = training.Trainer(model, training.TrainConfig(densification_interval=2, densify_until_iter=3, max_grad=1e-4))
    out = None
    for it in (1, 2, 3):                     # iteration 2 densifies: the replicas must stay identical through it
        out = tr.step(it, [views[rank]], global_views=world)
    q.put((rank, {k: getattr(model, k).detach().cpu().numpy() for k in NAMES}, out["gaussians"]))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_training_matches_single_process():
    """Two ranks (gloo, both on cuda:0), one view each, factored gradient exchange, against one process rendering both
    views: same parameters after three iterations including a densification, and bit-identical replicas."""
    import socket
    import torch.multiprocessing as mp
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r, (params, n)) for r, params, n in (q.get(timeout=150) for _ in range(2)))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    model_mod = importlib.import_module(PKG + ".model")
    training = importlib.import_module(PKG + ".training")
    s, views = _scene()
    model = model_mod.GaussianModel({k: torch.tensor(s[k]) for k in NAMES}, device="cuda:0")
    tr = training.Trainer(model, training.TrainConfig(densification_interval=2, densify_until_iter=3, max_grad=1e-4))
    for it in (1, 2, 3):
        out = tr.step(it, views)
    assert got[0][1] == got[1][1] == out["gaussians"]
    for k in NAMES:
        assert np.array_equal(got[0][0][k], got[1][0][k]), k                       # replicas bit-identical
        ref = getattr(model, k).detach().cpu().numpy()
        # Adam normalises: compare the movement where it is well above the fp32 noise of the gradient sums
        err = np.abs(got[0][0][k] - ref)
        assert np.quantile(err, 0.99) <= 1e-4 * max(1.0, np.abs(ref).max()), (k, float(np.quantile(err, 0.99)))


def _rccl_worker(port, q):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    gs = importlib.import_module(PKG)
    dp = importlib.import_module(PKG + ".dp")
    s, views = _scene()
    res = {}
    for mode in ("plain", "factored"):
        p = {k: torch.tensor(s[k], device="cuda:0").requires_grad_(True) for k in NAMES}
        ex = dp.FactoredExchange(p, world_views=2, force_collectives=True) if mode == "factored" else None
        if ex is not None:
            ex.__enter__()
        for v in views:
            img = gs.render_gaussians(*[p[k] for k in ("pos", "f_dc", "f_rest", "opacity_raw", "scale_raw", "q_raw")],
                                      torch.tensor(v["c2w"], device="cuda:0"), v["H"], v["W"], v["fx"], v["fy"], v["cx"], v["cy"])
            (img * torch.tensor(v["image"], device="cuda:0")).sum().backward()
        if ex is not None:
            ex.__exit__(None, None, None)
            ex.finish()                                   # all_reduce + all_gather_into_tensor over RCCL (one rank)
        else:
            dp.allreduce_gradients([p[k].grad for k in NAMES], world_views=2)
        res[mode] = {k: p[k].grad.cpu().numpy() for k in NAMES}
    q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_collectives_run_over_rccl():
    """The collectives of the data-parallel exchange through the real RCCL backend (a one-rank group is what a one-GPU box
    allows): same gradients as the plain path."""
    import socket
    import torch.multiprocessing as mp
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    proc = ctx.Process(target=_rccl_worker, args=(port, q))
    proc.start()
    res = q.get(timeout=150)
    proc.join(timeout=120)
    assert proc.exitcode == 0
    for k in NAMES:
        scale = np.abs(res["plain"][k]).max()
        assert np.abs(res["factored"][k] - res["plain"][k]).max() <= 2e-5 * scale, k


def test_trainer_step_does_not_wait_per_view_in_steady_state():
    """Two views per iteration: after the first iteration (which learns the pair capacity) no forward pass waits for its
    counters -- the per-frame checks are made once per iteration (ops.deferred_checks)."""
    model_mod = importlib.import_module(PKG + ".model")
    training = importlib.import_module(PKG + ".training")
    ops = importlib.import_module(PKG + ".ops")
    s, views = _scene()
    model = model_mod.GaussianModel({k: torch.tensor(s[k]) for k in NAMES}, device="cuda:0")
    tr = training.Trainer(model, training.TrainConfig(densify_until_iter=0, opacity_reset_interval=10 ** 9))
    tr.step(1, views)
    before = dict(ops.forward_modes)
    losses = [float(tr.step(it, views)["loss"]) for it in (2, 3, 4)]
    assert ops.forward_modes["waited"] == before["waited"] and ops.forward_modes["deferred"] == before["deferred"] + 6
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_views_on_two_streams_train_like_views_on_one():
    """TrainConfig.view_streams: the views of an iteration alternate between two HIP streams (the front of view k + 1 beside the
    rasterisation of view k).  With deterministic gradients the parameters after three iterations are bit-identical to the
    one-stream loop's: same kernels, same accumulation order."""
    model_mod = importlib.import_module(PKG + ".model")
    training = importlib.import_module(PKG + ".training")
    ops = importlib.import_module(PKG + ".ops")
    s, views = _scene()
    views = views + views[::-1]                                  # four views per iteration
    ops.set_deterministic(True)
    try:
        out = []
        for streams, in_kernel in ((1, False), (2, False), (1, True), (2, True)):      # (in_kernel: the views' gradients summed by the projection
            #  backward itself, ops.accumulate_grads -- the same sums in the same order as autograd's accumulation)
            model = model_mod.GaussianModel({k: torch.tensor(s[k]) for k in NAMES}, device="cuda:0")
            tr = training.Trainer(model, training.TrainConfig(densify_until_iter=0, opacity_reset_interval=10 ** 9, view_streams=streams,
                                                              sum_views_in_kernel=in_kernel))
            losses = [float(tr.step(it, views)["loss"]) for it in (1, 2, 3)]
            torch.cuda.synchronize()
            out.append((losses, {k: getattr(model, k).detach().clone() for k in NAMES}))
    finally:
        ops.set_deterministic(False)
    for other in out[1:]:
        assert out[0][0] == other[0], (out[0][0], other[0])
        for k in NAMES:
            assert torch.equal(out[0][1][k], other[1][k]), k


def test_adam_step_of_f_rest_inside_the_backward_pass_is_the_optimisers_step():
    """TrainConfig.fold_rest_step: an iteration of one view applies the Adam step of f_rest inside the projection backward
    (gsplat_backward_adam_rest).  Same arithmetic as the optimiser's kernel on the gradient the backward would have written: with
    deterministic gradients every parameter -- and both moments of f_rest -- are bit-identical after three iterations.  A pass that
    outgrew its pair buffers steps nothing (the kernel reads the frame's counters itself) and is repeated; an off-screen view raises
    the reference's exception with every parameter untouched."""
    model_mod = importlib.import_module(PKG + ".model")
    training = importlib.import_module(PKG + ".training")
    ops = importlib.import_module(PKG + ".ops")
    s, views = _scene()
    one = views[:1]
    ops.set_deterministic(True)
    try:
        res = []
        for fold in (False, True):
            model = model_mod.GaussianModel({k: torch.tensor(s[k]) for k in NAMES}, device="cuda:0")
            tr = training.Trainer(model, training.TrainConfig(densify_until_iter=0, opacity_reset_interval=10 ** 9, fold_rest_step=fold))
            tr.step(1, one)                                    # (the first frame of a scene waits for its counters: the ordinary backward)
            calls = ops.composite_calls["backward"]
            losses = [float(tr.step(it, one)["loss"]) for it in (2, 3, 4)]
            assert ops.composite_calls["backward"] == calls + 3
            st = tr.optimizer.state[model.f_rest]
            assert st['step'] == 4 and (model.f_rest.grad is None) == fold
            if fold:                                           # an iteration whose buffers are too small: one garbage pass, one good one
                import types
                key = ops.capacity_key(torch.device("cuda:0"), types.SimpleNamespace(H=one[0]["H"], W=one[0]["W"]), len(s["pos"]))
                keep = ops._ws.capacity[key]
            torch.cuda.synchronize()
            res.append((losses, {k: getattr(model, k).detach().clone() for k in NAMES}, st['exp_avg'].clone(), st['exp_avg_sq'].clone()))
            if fold:
                ops._ws.capacity[key] = 100
                before = dict(ops.forward_modes)
                tr.step(5, one)
                assert ops.forward_modes["deferred"] == before["deferred"] + 2 and tr.optimizer.state[model.f_rest]['step'] == 5
                folded_after_redo = model.f_rest.detach().clone()
                ops._ws.capacity[key] = keep
            else:
                tr.step(5, one)
                plain_after = model.f_rest.detach().clone()
    finally:
        ops.set_deterministic(False)
    assert res[0][0] == res[1][0]
    for k in NAMES:
        assert torch.equal(res[0][1][k], res[1][1][k]), k
    assert torch.equal(res[0][2], res[1][2]) and torch.equal(res[0][3], res[1][3])
    assert torch.equal(plain_after, folded_after_redo)        # the garbage pass stepped nothing
    # off screen: the reference's exception, nothing moved
    g = scenes.case_g10()
    model = model_mod.GaussianModel({k: torch.tensor(g[k]) for k in NAMES}, device="cuda:0")
    tr = training.Trainer(model, training.TrainConfig(densify_until_iter=0, opacity_reset_interval=10 ** 9))
    target = np.random.default_rng(9).uniform(0, 1, (g["H"], g["W"], 3)).astype(np.float32)
    good = dict(image=target, c2w=g["c2w"], H=g["H"], W=g["W"], fx=g["fx"], fy=g["fy"], cx=g["cx"] + 40.0, cy=g["cy"])
    tr.step(1, [good])
    tr.step(2, [good])
    before = {k: getattr(model, k).detach().clone() for k in NAMES}
    step_before = tr.optimizer.state[model.f_rest]['step']
    with pytest.raises(Exception, match="off-screen"):
        tr.step(3, [dict(good, cx=g["cx"])])
    torch.cuda.synchronize()
    assert all(torch.equal(before[k], getattr(model, k).detach()) for k in NAMES)
    assert tr.optimizer.state[model.f_rest]['step'] == step_before


def test_config4_training_iteration():
    """BASELINE.json config 4 as it says: the 3 M-Gaussian scene at 1080p through Trainer.step (render + L1/SSIM loss + backward
    + clip + Adam), two views.  Finite, every parameter moves, and the loss of view 0 equals the oracle's compute_loss of the
    oracle's render on a window of the frame (all 3 M Gaussians, float64)."""
    model_mod = importlib.import_module(PKG + ".model")
    training = importlib.import_module(PKG + ".training")
    gs = importlib.import_module(PKG)
    losses = importlib.import_module(PKG + ".losses")
    s = scenes.synthetic_scene(4)
    H, W = s["H"], s["W"]
    g = torch.Generator().manual_seed(3)
    views = [dict(image=torch.rand(H, W, 3, generator=g), c2w=torch.tensor(scenes.orbit_c2w(k, 8)), H=H, W=W, fx=s["fx"], fy=s["fy"],
                  cx=s["cx"], cy=s["cy"]) for k in range(2)]
    init = {k: torch.tensor(s[k]) for k in NAMES}
    # window check first (parameters untouched): 1920 x 48 rows of view 0, loss on the window
    h, y0 = 48, 516
    win = (h, W, s["fx"], s["fy"], s["cx"], s["cy"] - y0)
    dev = torch.device("cuda:0")
    with torch.no_grad():
        img = gs.render_gaussians(*[init[k].to(dev) for k in ("pos", "f_dc", "f_rest", "opacity_raw", "scale_raw", "q_raw")],
                                  views[0]["c2w"].to(dev), *win)
        _, vals = losses.compute_loss_device(img, views[0]["image"][y0:y0 + h].to(dev), 0.8, 0.2)
    torch.set_num_threads(16)
    with torch.no_grad():
        ref = tp.render_fused(*[init[k].double() for k in ("pos", "f_dc", "f_rest", "opacity_raw", "scale_raw", "q_raw")],
                              views[0]["c2w"].double(), *win)
        ref_total, ref_l1, ref_ssim = tp.compute_loss(ref, views[0]["image"][y0:y0 + h].double(), 0.8, 0.2)
    got = vals.cpu().numpy()          # (l1, 1 - ssim, total)
    print("config 4 window loss: HIP", got, "oracle", float(ref_l1), float(ref_ssim), float(ref_total))
    assert abs(got[2] - float(ref_total)) <= 2e-5 * abs(float(ref_total)) + 1e-6
    # the training iteration at full size
    model = model_mod.GaussianModel(init, device="cuda:0")
    before = {k: getattr(model, k).detach().clone() for k in NAMES}
    tr = training.Trainer(model, training.TrainConfig())
    out = tr.step(1, views)
    torch.cuda.synchronize()
    assert np.isfinite(float(out["loss"])) and 0.0 < float(out["loss"]) < 1.0 and out["gaussians"] == 3_000_000
    for k in NAMES:
        p = getattr(model, k)
        assert torch.isfinite(p).all() and torch.isfinite(p.grad).all(), k
        moved = float((p.detach() - before[k]).abs().max())
        assert moved > 0.0, f"{k} did not move"
    out2 = tr.step(2, views)           # steady state: no waiting forward, still finite
    assert np.isfinite(float(out2["loss"]))


def test_trainer_repeats_a_pass_that_outgrew_the_pair_buffers():
    """The pair capacity kept from earlier frames is too small for this iteration's views (as after a densification or a new
    camera): the pass is repeated with larger buffers and the iteration's result is the one of an undisturbed iteration."""
    model_mod = importlib.import_module(PKG + ".model")
    training = importlib.import_module(PKG + ".training")
    ops = importlib.import_module(PKG + ".ops")
    s, views = _scene()

    def run(shrink):
        model = model_mod.GaussianModel({k: torch.tensor(s[k]) for k in NAMES}, device="cuda:0")
        tr = training.Trainer(model, training.TrainConfig(densify_until_iter=0, opacity_reset_interval=10 ** 9))
        tr.step(1, views)
        if shrink:
            import types
            key = ops.capacity_key(torch.device("cuda:0"), types.SimpleNamespace(H=views[0]["H"], W=views[0]["W"]), len(s["pos"]))
            ops._ws.capacity[key] = 100                       # far below the ~1600 pairs of a view
        before = dict(ops.forward_modes)
        out = tr.step(2, views)
        torch.cuda.synchronize()
        if shrink:                                            # one garbage pass + one good pass, nothing waited for
            assert ops.forward_modes["deferred"] == before["deferred"] + 4 and ops.forward_modes["waited"] == before["waited"]
            assert ops._ws.capacity[key] > 1600
        return float(out["loss"]), {k: getattr(model, k).detach().clone() for k in NAMES}

    loss_a, pa = run(False)
    loss_b, pb = run(True)
    assert abs(loss_a - loss_b) <= 1e-6 * abs(loss_a)
    for k in NAMES:      # same update up to the summation order of the gradient atomics
        assert float((pa[k] - pb[k]).abs().max()) <= 1e-5 * max(1.0, float(pa[k].abs().max())), k


def _dp_offscreen_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    model_mod = importlib.import_module(PKG + ".model")
    training = importlib.import_module(PKG + ".training")
    s = scenes.case_g10()                                        # 48 Gaussians whose centres project into the guard band LEFT of a 32 x 32 image
    model = model_mod.GaussianModel({k: torch.tensor(s[k]) for k in NAMES}, device="cuda:0")
    tr = training.Trainer(model, training.TrainConfig(densify_until_iter=0, opacity_reset_interval=10 ** 9))
    target = np.random.default_rng(9).uniform(0, 1, (s["H"], s["W"], 3)).astype(np.float32)
    good = dict(image=target, c2w=s["c2w"], H=s["H"], W=s["W"], fx=s["fx"], fy=s["fy"], cx=s["cx"] + 40.0, cy=s["cy"])   # shifted: on screen
    tr.step(1, [good], global_views=world)                      # a good iteration first: leaves the pair capacity (later frames do not wait)
    bad = dict(good, cx=s["cx"]) if rank == 1 else good         # rank 1: survivors (guard band), nothing on screen
    before = {k: getattr(model, k).detach().clone() for k in NAMES}
    try:
        tr.step(2, [bad], global_views=world)
        q.put((rank, "ok", ""))
    except Exception as e:
        unchanged = all(torch.equal(before[k], getattr(model, k).detach()) for k in NAMES)
        q.put((rank, type(e).__name__, str(e) + ("" if unchanged else " [parameters changed]")))
    dist.barrier()
    dist.destroy_process_group()


def test_an_offscreen_view_on_one_rank_raises_on_every_rank():
    """Data parallel, real renderer: rank 1's view has survivors but nothing on screen.  The reference raises
    Exception("All projected points are off-screen") for such a view; here EVERY rank raises it, in the same step, within the
    timeout (nobody waits in a collective), and no rank has stepped its optimiser."""
    import socket
    import torch.multiprocessing as mp
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_offscreen_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r, (name, msg)) for r, name, msg in (q.get(timeout=150) for _ in range(2)))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ops = importlib.import_module(PKG + ".ops")
    assert got[1] == ("Exception", ops.OFFSCREEN_MSG), got
    assert got[0][0] == "Exception" and got[0][1].startswith(ops.OFFSCREEN_MSG) and "changed" not in got[0][1], got


def test_a_model_whose_sh_tensors_are_converted_by_the_render_is_not_silently_left_out_of_the_exchange():
    """dp.FactoredExchange.owns() compares the caller's OWN f_dc / f_rest tensors (before the host's dtype / layout conversion):
    a float64 model still goes through the factored exchange, and Trainer.step would raise if a view did not."""
    gs = importlib.import_module(PKG)
    dp = importlib.import_module(PKG + ".dp")
    s, views = _scene()
    v = views[0]
    p = {k: torch.tensor(s[k], dtype=torch.float64, device="cuda:0").requires_grad_(True) for k in NAMES}
    ex = dp.FactoredExchange(p, world_views=1)
    with ex:
        img = gs.render_gaussians(*[p[k] for k in ("pos", "f_dc", "f_rest", "opacity_raw", "scale_raw", "q_raw")],
                                  torch.tensor(v["c2w"], device="cuda:0"), v["H"], v["W"], v["fx"], v["fy"], v["cx"], v["cy"])
        (img * torch.tensor(v["image"], device="cuda:0")).sum().backward()
    assert ex.n_added == 1 and p["f_dc"].grad is None            # the SH gradients went to the exchange, not into .grad
    ex.finish()
    assert p["f_dc"].grad is not None and float(p["f_rest"].grad.abs().max()) > 0
