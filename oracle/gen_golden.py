#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference implementation.

TEST INFRASTRUCTURE.  Runs only in the build container, where the reference checkout is
mounted read-only at /root/reference (it never travels to the GPU box; the .npz fixtures
do).  The reference is imported unmodified:

    gaussian_splatting.build_sigma_from_params   gaussian_splatting/gaussian.py:71
    gaussian_splatting.evaluate_sh               gaussian_splatting/spherical_harmonics.py:70
    gaussian_splatting.render                    gaussian_splatting/render.py:62

Each fixture stores (a) the float32-representable inputs, (b) the float64 reference image,
(c) the six parameter gradients of L = sum(image * Wrand) in float64, (d) the float32
reference image (the reference's own fp32-vs-fp64 disagreement = tolerance floor), and
(e) per-stage intermediates read from render()'s frame locals at return.

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [case ...]
"""
import os
import sys
import hashlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("GS_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import gaussian_splatting as ref  # noqa: E402  (the reference)
from oracle import scenes  # noqa: E402

REF_RENDER_MOD = sys.modules["gaussian_splatting.render"]
OUT = os.path.join(ROOT, "tests", "golden")
PARAMS = ["pos", "scale_raw", "q_raw", "opacity_raw", "f_dc", "f_rest"]


class _Locals:
    """sys.setprofile hook: copy render()'s frame locals when it returns (or raises)."""

    def __init__(self):
        self.snap = None
        self.code = REF_RENDER_MOD.render.__code__

    def __call__(self, frame, event, arg):
        if event == "return" and frame.f_code is self.code:
            self.snap = dict(frame.f_locals)


def _t(x, dtype, grad=False):
    t = torch.tensor(np.asarray(x), dtype=dtype)
    return t.requires_grad_(grad)


def run_fused(s, dtype, wrand=None, capture=False):
    """The reference call sequence of scripts/train.py:463,502,505 on one view."""
    p = {k: _t(s[k], dtype, grad=wrand is not None) for k in PARAMS}
    c2w = _t(s["c2w"], dtype)
    sigma = ref.build_sigma_from_params(p["scale_raw"], p["q_raw"])
    color = ref.evaluate_sh(p["f_dc"], p["f_rest"], p["pos"], c2w)
    hook = _Locals()
    if capture:
        sys.setprofile(hook)
    try:
        img = ref.render(p["pos"], color, p["opacity_raw"], sigma, c2w, s["H"], s["W"], s["fx"], s["fy"],
                         s["cx"], s["cy"], **s["kwargs"])
    finally:
        sys.setprofile(None)
    grads = None
    if wrand is not None:
        loss = (img * _t(wrand, dtype)).sum()
        loss.backward()
        grads = {k: (p[k].grad if p[k].grad is not None else torch.zeros_like(p[k])).numpy() for k in PARAMS}
    return img.detach().numpy(), grads, sigma.detach().numpy(), color.detach().numpy(), hook.snap


def intermediates(snap):
    """Reduce render()'s locals to arrays indexed like the final (depth-sorted, on-screen) list."""
    out = {}
    if snap is None or "on_screen" not in snap or "inverse_covariance" not in snap:
        return out
    n0 = snap["opacity_mask"].numel()
    ids = torch.arange(n0)[snap["opacity_mask"]][snap["in_guard"]][snap["keep"]][snap["order"]][snap["on_screen"]]
    out["im_ids"] = ids.numpy().astype(np.int32)                       # original Gaussian index
    out["im_u"] = snap["u"].detach().numpy()
    out["im_v"] = snap["v"].detach().numpy()
    out["im_cov2d"] = snap["sigma_camera"].detach().numpy()            # after eigen clamp
    out["im_conic"] = snap["inverse_covariance"].detach().numpy()      # after det / diag clamp
    out["im_evals"] = snap["evals"][snap["on_screen"]].detach().numpy()
    out["im_opacity"] = snap["opacity"].detach().numpy()
    out["im_color"] = snap["color"].detach().numpy()
    out["im_tile_rect"] = torch.stack([snap["umin_tile"], snap["vmin_tile"], snap["umax_tile"],
                                       snap["vmax_tile"]], 1).numpy().astype(np.int32)
    out["im_pair_gauss"] = snap["gaussian_ids"].numpy().astype(np.int32)   # index into the final list
    out["im_tile_ids"] = snap["unique_tile_ids"].numpy().astype(np.int32)
    out["im_tile_start"] = snap["start"].numpy().astype(np.int32)
    out["im_tile_end"] = snap["end"].numpy().astype(np.int32)
    return out


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  wrote {os.path.relpath(path, ROOT)}  {os.path.getsize(path) / 1024:.0f} KiB")


def scene_arrays(s):
    d = {k: np.asarray(s[k], dtype=np.float32) for k in PARAMS}
    d["c2w"] = np.asarray(s["c2w"], dtype=np.float32)
    d["cam"] = np.array([s["H"], s["W"], s["fx"], s["fy"], s["cx"], s["cy"]], dtype=np.float64)
    kw = s["kwargs"]
    d["kw_names"] = np.array(sorted(kw.keys()))
    d["kw_vals"] = np.array([kw[k] for k in sorted(kw.keys())], dtype=np.float64)
    return d


def gen_case(name):
    print(name)
    s = scenes.CASES[name]()
    rng = np.random.default_rng(int(hashlib.sha1(name.encode()).hexdigest()[:8], 16))
    wrand = rng.uniform(0.0, 1.0, (s["H"], s["W"], 3))
    arrs = scene_arrays(s)
    arrs["wrand"] = wrand
    if name == "g10_offscreen":
        try:
            run_fused(s, torch.float64)
            raise SystemExit("expected the reference to raise for " + name)
        except Exception as e:  # the reference raises a bare Exception (render.py:236)
            arrs["raises"] = np.array(str(e))
            print("  reference raised:", e)
        save(name, **arrs)
        return
    img64, g64, sigma64, color64, snap = run_fused(s, torch.float64, wrand, capture=True)
    img32, g32, _, _, _ = run_fused(s, torch.float32, wrand)
    arrs["image"] = img64
    arrs["image_f32"] = img32.astype(np.float32)
    arrs["sigma"] = sigma64
    arrs["color"] = color64
    for k in PARAMS:
        arrs["grad_" + k] = g64[k]
        arrs["grad32_" + k] = g32[k].astype(np.float32)
        if not np.isfinite(g64[k]).all():
            raise SystemExit(f"{name}: non-finite reference gradient for {k} (degenerate eigenvalues?)")
    arrs.update(intermediates(snap))
    d = np.abs(img64 - img32)
    nv = len(arrs.get("im_ids", []))
    npairs = len(arrs.get("im_pair_gauss", []))
    print(f"  image mean {img64.mean():.4f} max {img64.max():.4f}  V={nv} P={npairs} "
          f"fp32-vs-fp64 max {d.max():.2e} (> 1e-5: {(d > 1e-5).sum()} values)")
    for k in PARAMS:
        den = np.linalg.norm(g64[k]) + 1e-300
        print(f"    grad {k:12s} |g|={den:.3e}  fp32 rel-L2 err {np.linalg.norm(g64[k] - g32[k]) / den:.2e}")
    save(name, **arrs)


def gen_unfused():
    """G11: render() called directly with arbitrary colour / covariance (the un-fused boundary)."""
    name = "g11_unfused"
    print(name)
    rng = np.random.default_rng(211)
    s = scenes._base(rng, 500, 64, 96, 80.0, 80.0, 48.0, 32.0)
    N = 500
    A = rng.normal(0, 0.12, (N, 3, 3))
    sigma = (A @ A.transpose(0, 2, 1) + 1e-4 * np.eye(3)).astype(np.float32)
    sigma = 0.5 * (sigma + sigma.transpose(0, 2, 1))
    color = rng.uniform(-0.4, 1.6, (N, 3)).astype(np.float32)     # outside [0,1]: exercises the output clamp mask
    wrand = rng.uniform(0.0, 1.0, (s["H"], s["W"], 3))
    outs = {}
    for dtype, tag in ((torch.float64, ""), (torch.float32, "32")):
        pos = _t(s["pos"], dtype, True)
        col = _t(color, dtype, True)
        opa = _t(s["opacity_raw"], dtype, True)
        sig = _t(sigma, dtype, True)
        img = ref.render(pos, col, opa, sig, _t(s["c2w"], dtype), s["H"], s["W"], s["fx"], s["fy"], s["cx"], s["cy"])
        (img * _t(wrand, dtype)).sum().backward()
        outs["image" + ("_f32" if tag else "")] = img.detach().numpy()
        for k, t in (("pos", pos), ("color", col), ("opacity_raw", opa), ("sigma", sig)):
            outs[f"grad{tag}_{k}"] = t.grad.numpy()
    arrs = scene_arrays(s)
    arrs.update(outs, sigma_in=sigma, color_in=color, wrand=wrand)
    print(f"  image mean {outs['image'].mean():.4f}  clamped-high values {(outs['image'] >= 1).sum()} "
          f"clamped-low {(outs['image'] <= 0).sum()}")
    save(name, **arrs)


def gen_pieces():
    """Stand-alone goldens for the small exported functions (forward + backward), float64."""
    print("pieces")
    rng = np.random.default_rng(311)
    N = 64
    scale_raw = rng.normal(-2, 1.5, (N, 3)).astype(np.float32)
    scale_raw[:4] = -15.0                                  # exp() < 1e-6 -> clamp_min branch
    q_raw = rng.normal(0, 1, (N, 4)).astype(np.float32)
    f_dc = rng.normal(0, 1, (N, 3)).astype(np.float32)
    f_rest = rng.normal(0, 0.5, (N, 45)).astype(np.float32)
    pts = rng.normal(0, 2, (N, 3)).astype(np.float32)
    c2w = scenes._camera(rng)
    m2 = rng.normal(0, 1, (N, 2, 2)).astype(np.float32)
    w_sigma = rng.normal(0, 1, (N, 3, 3))
    w_col = rng.normal(0, 1, (N, 3))
    w_rot = rng.normal(0, 1, (N, 3, 3))
    d = torch.float64
    sr, qr = _t(scale_raw, d, True), _t(q_raw, d, True)
    sig = ref.build_sigma_from_params(sr, qr)
    (sig * _t(w_sigma, d)).sum().backward()
    fd, fr, pt = _t(f_dc, d, True), _t(f_rest, d, True), _t(pts, d, True)
    col = ref.evaluate_sh(fd, fr, pt, _t(c2w, d))
    (col * _t(w_col, d)).sum().backward()
    qq = _t(q_raw, d, True)
    rot = ref.quat_to_rotmat(qq)
    (rot * _t(w_rot, d)).sum().backward()
    inv = ref.inv2x2(_t(m2, d))
    H, W, Hs, Ws = 540, 960, 1080, 1920
    si = np.array(ref.scale_intrinsics(H, W, Hs, Ws, 1100.0, 1090.0, 961.5, 538.25))
    uv, x, y, z = ref.project_points(_t(pts, d), _t(c2w, d), 500.0, 510.0, 320.0, 240.0)
    save("pieces", scale_raw=scale_raw, q_raw=q_raw, f_dc=f_dc, f_rest=f_rest, points=pts, c2w=c2w, m2=m2,
         w_sigma=w_sigma, w_col=w_col, w_rot=w_rot, sigma=sig.detach().numpy(), grad_scale_raw=sr.grad.numpy(),
         grad_q_raw=qr.grad.numpy(), color=col.detach().numpy(), grad_f_dc=fd.grad.numpy(),
         grad_f_rest=fr.grad.numpy(), grad_points=pt.grad.numpy(), rot=rot.detach().numpy(),
         grad_q_rot=qq.grad.numpy(), inv2x2=inv.numpy(), scale_intrinsics=si,
         harmonics_names=np.array(sorted(ref.HARMONICS)), harmonics_vals=np.array(
             [ref.HARMONICS[k] for k in sorted(ref.HARMONICS)]),
         proj_uv=uv.numpy(), proj_xyz=torch.stack([x, y, z], 1).numpy())


def gen_loss():
    """Next-row fixture: compute_loss (gaussian_splatting/losses.py:158-185) forward + d/dpred, float64."""
    print("loss")
    import importlib
    losses = importlib.import_module("gaussian_splatting.losses")
    rng = np.random.default_rng(411)
    out = {}
    for tag, shape in (("a", (45, 70, 3)), ("b", (2, 40, 52, 3)), ("c", (9, 7, 3))):
        tgt = rng.uniform(0, 1, shape).astype(np.float32)
        pred = np.clip(tgt + rng.normal(0, 0.15, shape), 0, 1).astype(np.float32)
        pred[..., :3, :, :][..., :2, :] = tgt[..., :3, :, :][..., :2, :]          # some exactly equal pixels: sign(0) = 0
        p = _t(pred, torch.float64, True)
        total, parts = losses.compute_loss(p, _t(tgt, torch.float64), 0.8, 0.2)
        total.backward()
        out.update({f"pred_{tag}": pred, f"target_{tag}": tgt, f"grad_{tag}": p.grad.numpy(),
                    f"vals_{tag}": np.array([parts["l1"], parts["ssim"], parts["total"]])})
        p2 = _t(pred, torch.float64, True)
        t2, _ = losses.compute_loss(p2, _t(tgt, torch.float64), 0.3, 1.7)
        t2.backward()
        out.update({f"grad2_{tag}": p2.grad.numpy(), f"total2_{tag}": np.array(float(t2))})
        print(f"  {tag} {shape}: l1 {parts['l1']:.5f} ssim-loss {parts['ssim']:.5f} total {parts['total']:.5f}")
    save("loss", **out)


def gen_optim():
    """Next-row fixture: the optimiser step at the reference's call sites (scripts/train.py:394-401 Adam groups with
    eps = 1e-15, :446-457 position-LR schedule, :536 clip_grad_norm_(pos, 1.0), :538 optimizer.step()), run with
    torch.optim.Adam itself in float64 for three iterations on small tensors."""
    print("optim")
    rng = np.random.default_rng(511)
    n = 37
    shapes = {"pos": (n, 3), "opacity_raw": (n,), "f_dc": (n, 3), "f_rest": (n, 45), "scale_raw": (n, 3), "q_raw": (n, 4)}
    init = {k: rng.normal(0, 1, s).astype(np.float32) for k, s in shapes.items()}
    iters = [0, 1, 299, 300, 29999, 30000]        # delay phase, its end, decay end
    grads = {k: np.stack([rng.normal(0, 3.0 if k == "pos" else 1.0, s) * (10.0 ** rng.integers(-3, 1)) for _ in iters]).astype(np.float32)
             for k, s in shapes.items()}
    P = {k: torch.nn.Parameter(_t(init[k], torch.float64)) for k in shapes}
    position_lr_init, position_lr_final, delay_mult, max_steps = 0.00016, 0.0000016, 0.01, 30000
    feature_lr, opacity_lr, scaling_lr, rotation_lr = 0.0025, 0.05, 0.005, 0.001
    opt = torch.optim.Adam([
        {'params': [P["pos"]], 'lr': position_lr_init, 'name': 'pos'},
        {'params': [P["opacity_raw"]], 'lr': opacity_lr, 'name': 'opacity'},
        {'params': [P["f_dc"]], 'lr': feature_lr, 'name': 'f_dc'},
        {'params': [P["f_rest"]], 'lr': feature_lr / 20.0, 'name': 'f_rest'},
        {'params': [P["scale_raw"]], 'lr': scaling_lr, 'name': 'scale'},
        {'params': [P["q_raw"]], 'lr': rotation_lr, 'name': 'rotation'},
    ], lr=0.01, eps=1e-15)
    out = {"iters": np.array(iters)}
    lrs, norms = [], []
    for j, it in enumerate(iters):
        if it < max_steps:
            plr = position_lr_init * (position_lr_final / position_lr_init) ** (it / max_steps)
        else:
            plr = position_lr_final
        if it < delay_mult * max_steps:
            plr *= 0.01
        opt.param_groups[0]['lr'] = plr
        lrs.append(plr)
        for k in shapes:
            P[k].grad = _t(grads[k][j], torch.float64)
        norms.append(float(torch.nn.utils.clip_grad_norm_(P["pos"], max_norm=1.0)))
        opt.step()
        for k in shapes:
            out[f"after{j}_{k}"] = P[k].detach().numpy().copy()
        out[f"clipped{j}_pos"] = P["pos"].grad.numpy().copy()
    out["pos_lr"] = np.array(lrs)
    out["pos_grad_norm"] = np.array(norms)
    for k in shapes:
        out["init_" + k] = init[k]
        out["grads_" + k] = grads[k]
    print("  position lr per iteration:", lrs, " pos grad norms:", [round(x, 3) for x in norms])
    save("optim", **out)


def gen_harness():
    """Next-row fixtures: create_orbit_trajectory (scripts/render_trained.py:28-75) outputs, and a checkpoint + the six
    loose tensor files written by the reference's own GaussianModel.save_checkpoint / training-loop lines."""
    print("harness")
    import importlib.util
    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, "scripts", name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    rt = load("render_trained")
    out = {}
    for tag, args in (("a", dict(center=[0.3, -1.2, 0.5], radius=3.0, num_frames=12, elevation=0.0)),
                      ("b", dict(center=[0.0, 0.0, 0.0], radius=4.5, num_frames=7, elevation=0.4))):
        out["orbit_" + tag] = rt.create_orbit_trajectory(**args)
        out["orbit_args_" + tag] = np.array(list(args["center"]) + [args["radius"], args["num_frames"], args["elevation"]])
    save("harness", **out)
    tr = load("train")
    rng = np.random.default_rng(611)
    n = 11
    model = object.__new__(tr.GaussianModel)              # only the tensors matter to save_checkpoint
    shapes = {"pos": (n, 3), "opacity_raw": (n,), "f_dc": (n, 3), "f_rest": (n, 45), "scale_raw": (n, 3), "q_raw": (n, 4)}
    for k, sh in shapes.items():
        setattr(model, k, torch.nn.Parameter(_t(rng.normal(0, 1, sh).astype(np.float32), torch.float32)))
    ck_dir = os.path.join(OUT, "ref_checkpoint")
    os.makedirs(ck_dir, exist_ok=True)
    model.save_checkpoint(os.path.join(ck_dir, "checkpoint_001000.pt"), 1000)
    # the loose files of scripts/train.py:590-597
    it = 2000
    torch.save(model.pos.cpu(), os.path.join(ck_dir, f"pos_{it}.pt")); torch.save(model.opacity_raw.cpu(), os.path.join(ck_dir, f"opacity_raw_{it}.pt"))
    torch.save(model.f_dc.cpu(), os.path.join(ck_dir, f"f_dc_{it}.pt")); torch.save(model.f_rest.cpu(), os.path.join(ck_dir, f"f_rest_{it}.pt"))
    torch.save(model.scale_raw.cpu(), os.path.join(ck_dir, f"scale_raw_{it}.pt")); torch.save(model.q_raw.cpu(), os.path.join(ck_dir, f"q_rot_{it}.pt"))
    np.savez(os.path.join(ck_dir, "expected.npz"), **{k: getattr(model, k).detach().numpy() for k in shapes})
    print("  wrote", ck_dir)


def gen_data():
    """Next-row fixtures: a tiny scene in the reference's on-disk layout, and what the reference's own loaders return for
    it (GaussianDataset, load_point_cloud, initialize_gaussians_from_pointcloud; gaussian_splatting/data_loader.py)."""
    print("data")
    from PIL import Image
    dl = sys.modules.get("gaussian_splatting.data_loader") or __import__("gaussian_splatting.data_loader", fromlist=["x"])
    rng = np.random.default_rng(711)
    root = os.path.join(OUT, "ref_dataset")
    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    for i in range(3):
        Image.fromarray(rng.integers(0, 256, (12, 18, 3), dtype=np.uint8)).save(os.path.join(root, "images", f"{i:04d}.png"))
    np.save(os.path.join(root, "cam_meta.npy"), {'fx': 20.5, 'fy': 21.0, 'cx': 9.25, 'cy': 5.75, 'height': 12, 'width': 18}, allow_pickle=True)
    np.save(os.path.join(root, "poses.npy"), np.stack([scenes._camera(rng) for _ in range(3)]))
    pts = rng.normal(0, 2, (40, 3))
    pts[5] = [np.nan, 0, 0]; pts[9] = [5000.0, 1, 1]
    with open(os.path.join(root, "pointcloud.ply"), "w") as f:
        f.write(f"ply\nformat ascii 1.0\nelement vertex {len(pts)}\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
        for x, y, z in pts:
            f.write(f"{x:.7g} {y:.7g} {z:.7g}\n")
    out = {}
    for sf in (1.0, 0.5):
        ds = dl.GaussianDataset(root, scale_factor=sf)
        tag = "full" if sf == 1.0 else "half"
        for i in range(len(ds)):
            smp = ds[i]
            out[f"{tag}_image{i}"] = smp['image'].numpy()
            out[f"{tag}_c2w{i}"] = smp['c2w'].numpy()
            out[f"{tag}_intr{i}"] = np.array([smp['fx'], smp['fy'], smp['cx'], smp['cy'], smp['H'], smp['W']], dtype=np.float64)
    cloud = dl.load_point_cloud(os.path.join(root, "pointcloud.ply"))
    out["cloud"] = cloud.numpy()
    torch.manual_seed(5)
    init = dl.initialize_gaussians_from_pointcloud(cloud, num_sh_bands=3)
    for k, v in init.items():
        out["init_" + k] = v.numpy()
    rgbpts = torch.cat([cloud, torch.rand(len(cloud), 3) * 255], 1)
    out["rgb_points"] = rgbpts.numpy()
    torch.manual_seed(6)
    out["init_rgb_f_dc"] = dl.initialize_gaussians_from_pointcloud(rgbpts)["f_dc"].numpy()
    save("data", **out)


def gen_densify():
    """Next-row fixture: adaptive density control.  Runs the reference's own GaussianModel.densify_and_prune
    (scripts/train.py:89-195) on CPU with torch.manual_seed fixed (the split draws randn_like), and the opacity-reset
    expression of the training loop (scripts/train.py:564-569, inline code there, evaluated here with the same torch calls)."""
    print("densify")
    import importlib.util
    spec = importlib.util.spec_from_file_location("train", os.path.join(REF, "scripts", "train.py"))
    tr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tr)
    rng = np.random.default_rng(711)
    n = 60
    shapes = {"pos": (n, 3), "opacity_raw": (n,), "f_dc": (n, 3), "f_rest": (n, 45), "scale_raw": (n, 3), "q_raw": (n, 4)}
    init = {k: rng.normal(0, 1, s).astype(np.float32) for k, s in shapes.items()}
    init["opacity_raw"] = rng.normal(-2.0, 3.0, n).astype(np.float32)          # a good share below sigmoid^-1(0.01) = -4.6
    init["scale_raw"] = rng.normal(-5.1, 0.6, (n, 3)).astype(np.float32)       # max exp() on both sides of 0.01
    gpos = (rng.normal(0, 1, (n, 3)) * 10.0 ** rng.integers(-4, 0, (n, 1))).astype(np.float32)   # norms on both sides of 0.01
    gopa = rng.normal(0, 1, n).astype(np.float32)
    out = {"init_" + k: v for k, v in init.items()}
    out["grad_pos"], out["grad_opacity_raw"] = gpos, gopa
    # name -> (kwargs, use grads)
    cases = {
        "split_only": (dict(opacity_threshold=0.01, max_grad=0.01, scale_threshold=1e-6), True),
        "clone_only": (dict(opacity_threshold=0.01, max_grad=0.01, scale_threshold=10.0), True),
        "prune_only": (dict(opacity_threshold=0.05, max_grad=1e9, scale_threshold=0.01), True),
        "nothing": (dict(opacity_threshold=1e-9, max_grad=1e9, scale_threshold=0.01), True),
        "no_grads": (dict(opacity_threshold=0.01, max_grad=0.01, scale_threshold=0.01), False),
        "both": (dict(opacity_threshold=0.01, max_grad=0.01, scale_threshold=0.01), True),
    }
    for name, (kw, use_grads) in cases.items():
        model = tr.GaussianModel({k: _t(v, torch.float32) for k, v in init.items()}, device="cpu")
        grads = {"pos": _t(gpos, torch.float32), "opacity_raw": _t(gopa, torch.float32)} if use_grads else None
        torch.manual_seed(1234)
        try:
            model.densify_and_prune(grads, **kw)
            raised = ""
        except Exception as e:                       # the reference's own failure when split and clone are both due
            raised = type(e).__name__
        out[f"{name}_kwargs"] = np.array([kw["opacity_threshold"], kw["max_grad"], kw["scale_threshold"]])
        out[f"{name}_raised"] = np.array(raised)
        for k in shapes:
            out[f"{name}_{k}"] = getattr(model, k).detach().numpy().copy()
        print(f"  {name}: {n} -> {model.get_num_gaussians()} Gaussians" + (f"  (reference raised {raised})" if raised else ""))
    # opacity reset, scripts/train.py:564-569
    o_raw = _t(init["opacity_raw"], torch.float32).clone()
    opacity = torch.sigmoid(o_raw)
    mask = opacity < 0.01
    o_raw[mask] = torch.logit(torch.clamp(opacity[mask] + 0.01, 0, 1))
    out["reset_opacity_raw"] = o_raw.numpy()
    out["reset_count"] = np.array(int(mask.sum()))
    save("densify", **out)


def gen_config1():
    """G12: config 1 at full size (10k Gaussians, 256x256, f_rest = 0): image + gradient digests."""
    name = "g13_config1_full"
    print(name)
    s = scenes.synthetic_scene(1)
    wrand = np.random.default_rng(1).uniform(0, 1, (s["H"], s["W"], 3)).astype(np.float32)
    img64, g64, _, _, snap = run_fused(s, torch.float64, wrand.astype(np.float64), capture=True)
    im = intermediates(snap)
    blk = img64.reshape(s["H"] // 2, 2, s["W"] // 2, 2, 3).mean(axis=(1, 3))
    arrs = dict(image_blockmean=blk.astype(np.float32), image_rows8=img64[::8].astype(np.float32),
                wrand_seed=np.array(1),
                input_digest=np.array([float(np.abs(s[k]).astype(np.float64).sum()) for k in PARAMS]),
                V=np.array(len(im["im_ids"])), P=np.array(len(im["im_pair_gauss"])))
    for k in PARAMS:
        g = g64[k]
        arrs["gnorm_" + k] = np.array(np.linalg.norm(g))
        arrs["grad_" + k + "_head"] = g[:1024].astype(np.float32)
    print(f"  V={arrs['V']} P={arrs['P']} image mean {img64.mean():.4f}")
    save(name, **arrs)


if __name__ == "__main__":
    torch.set_num_threads(8)
    want = sys.argv[1:]
    for n in scenes.CASES:
        if not want or n in want:
            gen_case(n)
    if not want or "g11_unfused" in want:
        gen_unfused()
    if not want or "pieces" in want:
        gen_pieces()
    if not want or "g13_config1_full" in want:
        gen_config1()
    if not want or "loss" in want:
        gen_loss()
    if not want or "optim" in want:
        gen_optim()
    if not want or "harness" in want:
        gen_harness()
    if not want or "data" in want:
        gen_data()
    if not want or "densify" in want:
        gen_densify()
