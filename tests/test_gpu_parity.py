"""GPU parity tests: the HIP path (through the C ABI) against the goldens produced by the real reference and
against the CPU oracle on seeded inputs.  Tolerances are the stated fp32 ones in tests/util.py."""
import numpy as np
import pytest
import torch

from oracle import scenes
from oracle import torch_port as tp
from tests import util

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
F32 = torch.float32


def _fused(gs, d, grad=True, dtype=F32):
    p = util.tensors(d, dtype, device=DEV, grad=grad)
    c2w = torch.tensor(d["c2w"], dtype=dtype, device=DEV)
    img = gs.render_gaussians(p["pos"], p["f_dc"], p["f_rest"], p["opacity_raw"], p["scale_raw"], p["q_raw"], c2w,
                              *util.cam_args(d), **d["kwargs"])
    if grad:
        (img * torch.tensor(d["wrand"], dtype=dtype, device=DEV)).sum().backward()
    return img, p


@pytest.mark.parametrize("name", util.RENDER_CASES)
def test_fused_vs_reference_golden(gs, name):
    d = util.load(name)
    img, p = _fused(gs, d)
    assert img.shape == (d["H"], d["W"], 3) and img.dtype == F32 and img.device.type == "cuda"
    util.check_image(img.detach().cpu().numpy(), d["image"], cal=d["image_f32"])
    for k in util.PARAMS:
        util.check_grad(p[k].grad.cpu().numpy(), d["grad_" + k], k, cal=d["grad32_" + k])


@pytest.mark.parametrize("name", ["g1_generic", "g2_ragged", "g6_huge", "g7_tiny", "g12_kwargs"])
def test_three_call_sequence_vs_reference_golden(gs, name):
    """build_sigma_from_params + evaluate_sh + render, chained by torch autograd exactly like scripts/train.py."""
    d = util.load(name)
    p = util.tensors(d, F32, device=DEV, grad=True)
    c2w = torch.tensor(d["c2w"], dtype=F32, device=DEV)
    sigma = gs.build_sigma_from_params(p["scale_raw"], p["q_raw"])
    color = gs.evaluate_sh(p["f_dc"], p["f_rest"], p["pos"], c2w)
    assert np.abs(sigma.detach().cpu().numpy() - d["sigma"]).max() <= 2e-6 * np.abs(d["sigma"]).max()
    assert np.abs(color.detach().cpu().numpy() - d["color"]).max() < 2e-6
    img = gs.render(p["pos"], color, p["opacity_raw"], sigma, c2w, *util.cam_args(d), **d["kwargs"])
    (img * torch.tensor(d["wrand"], dtype=F32, device=DEV)).sum().backward()
    util.check_image(img.detach().cpu().numpy(), d["image"], cal=d["image_f32"])
    for k in util.PARAMS:
        util.check_grad(p[k].grad.cpu().numpy(), d["grad_" + k], k, cal=d["grad32_" + k])


def test_unfused_boundary_vs_reference_golden(gs):
    d = util.load("g11_unfused")
    t = {k: torch.tensor(d[k], dtype=F32, device=DEV, requires_grad=True) for k in ("pos", "opacity_raw")}
    col = torch.tensor(d["color_in"], dtype=F32, device=DEV, requires_grad=True)
    sig = torch.tensor(d["sigma_in"], dtype=F32, device=DEV, requires_grad=True)
    img = gs.render(t["pos"], col, t["opacity_raw"], sig, torch.tensor(d["c2w"], device=DEV), *util.cam_args(d))
    (img * torch.tensor(d["wrand"], dtype=F32, device=DEV)).sum().backward()
    util.check_image(img.detach().cpu().numpy(), d["image"])
    for k, g in (("pos", t["pos"].grad), ("color", col.grad), ("opacity_raw", t["opacity_raw"].grad), ("sigma", sig.grad)):
        util.check_grad(g.cpu().numpy(), d["grad_" + k], k)


@pytest.mark.parametrize("name", util.EMPTY_CASES)
def test_empty_scene_is_zero_image_with_zero_grads(gs, name):
    d = util.load(name)
    img, p = _fused(gs, d)
    assert img.shape == (d["H"], d["W"], 3) and float(img.detach().abs().max()) == 0.0 and img.requires_grad
    for k in util.PARAMS:
        assert p[k].grad is not None and float(p[k].grad.abs().max()) == 0.0


def test_all_offscreen_raises_like_the_reference(gs):
    d = util.load("g10_offscreen")
    with pytest.raises(Exception, match=str(d["raises"])):
        _fused(gs, d, grad=False)


def test_zero_gaussians(gs):
    z = lambda *s: torch.zeros(*s, device=DEV)
    img = gs.render(z(0, 3), z(0, 3), z(0), z(0, 3, 3), torch.eye(4, device=DEV), 20, 30, 10., 10., 15., 10.)
    assert img.shape == (20, 30, 3) and float(img.abs().max()) == 0.0


def test_no_grad_and_tensor_hw(gs):
    """render_trained.py / inference.py call under no_grad; H, W may be 0-d tensors (train.py:499)."""
    d = util.load("g1_generic")
    with torch.no_grad():
        p = util.tensors(d, F32, device=DEV)
        c2w = torch.tensor(d["c2w"], device=DEV)
        img = gs.render_gaussians(p["pos"], p["f_dc"], p["f_rest"], p["opacity_raw"], p["scale_raw"], p["q_raw"], c2w,
                                  torch.tensor(d["H"]), torch.tensor(d["W"]), d["fx"], d["fy"], d["cx"], d["cy"])
    assert not img.requires_grad
    util.check_image(img.cpu().numpy(), d["image"], cal=d["image_f32"])


def test_float64_inputs_round_trip(gs):
    d = util.load("g2_ragged")
    img, p = _fused(gs, d, dtype=torch.float64)
    assert img.dtype == torch.float64 and p["pos"].grad.dtype == torch.float64
    util.check_image(img.detach().cpu().numpy(), d["image"], cal=d["image_f32"])
    util.check_grad(p["f_rest"].grad.cpu().numpy(), d["grad_f_rest"], "f_rest", cal=d["grad32_f_rest"])


def test_forward_is_deterministic_and_order_is_depth_then_index(gs):
    d = util.load("g3_occlusion")
    a, _ = _fused(gs, d, grad=False)
    b, _ = _fused(gs, d, grad=False)
    assert torch.equal(a, b)
    # permuting the Gaussians must not change the image (order inside a tile is by depth)
    perm = np.random.default_rng(5).permutation(len(d["pos"]))
    d2 = dict(d)
    for k in util.PARAMS:
        d2[k] = d[k][perm]
    c, _ = _fused(gs, d2, grad=False)
    assert float((a - c).abs().max()) < 2e-6


def test_config1_full_vs_reference_digest(gs):
    """Config 1 (10k Gaussians, 256x256, f_rest = 0) at full size against digests of the real reference."""
    d = dict(np.load(util.GOLDEN + "/g13_config1_full.npz"))
    s = scenes.synthetic_scene(1)
    dig = np.array([float(np.abs(s[k]).astype(np.float64).sum()) for k in util.PARAMS])
    if not np.allclose(dig, d["input_digest"], rtol=1e-12):
        pytest.skip("torch RNG stream differs from the one the fixture was generated with")
    img, p = _fused(gs, dict(s, wrand=np.random.default_rng(1).uniform(0, 1, (256, 256, 3)).astype(np.float32)))
    im = img.detach().cpu().numpy().astype(np.float64)
    assert np.abs(im.reshape(128, 2, 128, 2, 3).mean(axis=(1, 3)) - d["image_blockmean"]).max() < 1e-3
    util.check_image(im[::8], d["image_rows8"])
    for k in util.PARAMS:
        g = p[k].grad.cpu().numpy()
        assert abs(np.linalg.norm(g) - float(d["gnorm_" + k])) <= 1e-3 * float(d["gnorm_" + k]), k
        ref = d["grad_" + k + "_head"]
        assert np.abs(g[:1024] - ref).max() <= util.GRAD_TOL_MAX * np.abs(ref).max(), k


@pytest.mark.parametrize("n,hw,fx,mu", [(20000, (208, 304), 330.0, -3.6)])
def test_seeded_scene_vs_cpu_oracle(gs, n, hw, fx, mu):
    """A mid-size seeded scene (ragged image size, many tiles, long lists): HIP fp32 vs the oracle in float64."""
    g = torch.Generator().manual_seed(7)
    pos = torch.randn(n, 3, generator=g)
    pos[:, 2] += 5.0
    s = dict(pos=pos, scale_raw=torch.randn(n, 3, generator=g) * 0.4 + mu, q_raw=torch.randn(n, 4, generator=g),
             opacity_raw=torch.randn(n, generator=g) * 1.5, f_dc=torch.randn(n, 3, generator=g),
             f_rest=torch.randn(n, 45, generator=g) * 0.2)
    H, W = hw
    c2w = torch.tensor(scenes.orbit_c2w(1, 24))
    # Depth ties: fp32 cannot order two overlapping Gaussians whose camera depths agree to ~1e-6 (the reference's
    # argsort is not even stable there), so drop one Gaussian of every near-tie pair (|dz| < 2e-5).
    zc = tp.to_camera(s["pos"].double(), c2w.double())[2]
    zs, order = torch.sort(zc)
    drop = order[1:][(zs[1:] - zs[:-1]) < 2e-5]
    keep = torch.ones(n, dtype=torch.bool)
    keep[drop] = False
    s = {k: v[keep].contiguous() for k, v in s.items()}
    w = torch.rand(H, W, 3, generator=g)
    cam = (H, W, fx, fx * 1.03, W / 2 - 3.5, H / 2 + 2.25)
    p64 = {k: v.double().requires_grad_(True) for k, v in s.items()}
    ref = tp.render_fused(p64["pos"], p64["f_dc"], p64["f_rest"], p64["opacity_raw"], p64["scale_raw"], p64["q_raw"],
                          c2w.double(), *cam)
    (ref * w.double()).sum().backward()
    p = {k: v.to(DEV).requires_grad_(True) for k, v in s.items()}
    img = gs.render_gaussians(p["pos"], p["f_dc"], p["f_rest"], p["opacity_raw"], p["scale_raw"], p["q_raw"], c2w.to(DEV), *cam)
    (img * w.to(DEV)).sum().backward()
    # Threshold flips (q <= chi_square_clip, alpha >= alpha_cutoff) are inherent to fp32 and their density grows with
    # the number of Gaussian evaluations per pixel (~20 here).  Calibrated on the oracle's own fp32-vs-fp64 disagreement
    # (= the reference's pure-PyTorch fp32 path), image and gradients (tests/util.py).
    img32, g32 = _oracle(s, cam, c2w, w, torch.float32)
    r64 = ref.detach().numpy()
    got = img.detach().cpu().numpy()
    util.check_image(got, r64, cal=img32)
    assert np.abs(got - r64).mean() < 2e-6
    for k in util.PARAMS:
        util.check_grad(p[k].grad.cpu().numpy(), p64[k].grad.numpy(), k, cal=g32[k])


@pytest.mark.parametrize("n,longest,hw", [(6000, 4096, (32, 48)), (11000, 8192, (32, 48)), (6000, 4096, (512, 640))])
def test_long_lists_take_the_large_sort_paths(gs, n, longest, hw):
    """Thousands of Gaussians on the same pixels: lists of 4096-8191 entries use the largest LDS sort class, longer ones are
    sorted in global memory.  Third case: the same hot spot in a large, otherwise empty image -- the average list is short, the class of
    4096+ entries gets no launch of its own (gsplat_bin) and the launch of the 1024 .. 4095 class sorts the one long list in global memory."""
    (H, W), f = hw, 40.0
    g = torch.Generator().manual_seed(11)
    z = torch.rand(n, generator=g, dtype=torch.float64) * 4 + 2          # distinct depths
    zs, order = torch.sort(z)
    keep = torch.ones(n, dtype=torch.bool)
    keep[order[1:][(zs[1:] - zs[:-1]) < 2e-5]] = False
    uv = torch.rand(n, 2, generator=g, dtype=torch.float64) * 7.0 + 4.5            # all centres inside tile (0, 0)
    pos = torch.stack([(uv[:, 0] - W / 2) / f * z, (uv[:, 1] - H / 2) / f * z, z], 1).float()
    s = dict(pos=pos, scale_raw=torch.randn(n, 3, generator=g) * 0.2 - 1.6, q_raw=torch.randn(n, 4, generator=g),
             opacity_raw=torch.randn(n, generator=g) * 0.3 - 3.6,                  # opacity ~ 0.027: thousands of layers contribute
             f_dc=torch.randn(n, 3, generator=g), f_rest=torch.randn(n, 45, generator=g) * 0.2)
    s = {k: v[keep].contiguous() for k, v in s.items()}
    cam = (H, W, f, f, W / 2.0, H / 2.0)
    c2w = torch.eye(4)
    w = torch.rand(H, W, 3, generator=g)
    p64 = {k: v.double().requires_grad_(True) for k, v in s.items()}
    stages = {}
    ref = tp.render_fused(p64["pos"], p64["f_dc"], p64["f_rest"], p64["opacity_raw"], p64["scale_raw"], p64["q_raw"],
                          c2w.double(), *cam, stages=stages)
    assert int((stages["tile_end"] - stages["tile_start"]).max()) > longest
    (ref * w.double()).sum().backward()
    p = {k: v.to(DEV).requires_grad_(True) for k, v in s.items()}
    img = gs.render_gaussians(p["pos"], p["f_dc"], p["f_rest"], p["opacity_raw"], p["scale_raw"], p["q_raw"], c2w.to(DEV), *cam)
    (img * w.to(DEV)).sum().backward()
    # thousands of layers per pixel: thousands of threshold decisions per value -> calibrated on the oracle in float32
    img32, g32 = _oracle(s, cam, c2w, w, torch.float32)
    util.check_image(img.detach().cpu().numpy(), ref.detach().numpy(), cal=img32)
    for k in util.PARAMS:
        util.check_grad(p[k].grad.cpu().numpy(), p64[k].grad.numpy(), k, cal=g32[k])


def _oracle(s, cam, c2w, w, dtype, z_order_eps=None, **kw):
    """(image, gradients) of the oracle evaluated in `dtype` (float32 = the reference's own fp32 arithmetic: the calibration)."""
    q = {k: v.to(dtype) for k, v in s.items()}
    if z_order_eps is not None:            # make (depth, index) the unique order in float64: the HIP path's tie rule
        q["pos"] = q["pos"].clone()
        q["pos"][:, 2] += torch.arange(len(q["pos"]), dtype=dtype) * z_order_eps
    q = {k: v.detach().clone().requires_grad_(True) for k, v in q.items()}
    img = tp.render_fused(q["pos"], q["f_dc"], q["f_rest"], q["opacity_raw"], q["scale_raw"], q["q_raw"], c2w.to(dtype), *cam, **kw)
    (img * w.to(dtype)).sum().backward()
    return img.detach().double().numpy(), {k: (v.grad.double().numpy() if v.grad is not None else np.zeros(v.shape)) for k, v in q.items()}


def _vs_oracle(gs, s, cam, c2w, w, z_order_eps=None, calibrate=True):
    ref, g64 = _oracle(s, cam, c2w, w, torch.float64, z_order_eps)
    img32, g32 = _oracle(s, cam, c2w, w, torch.float32) if calibrate else (None, None)
    p = {k: v.to(DEV).requires_grad_(True) for k, v in s.items()}
    img = gs.render_gaussians(p["pos"], p["f_dc"], p["f_rest"], p["opacity_raw"], p["scale_raw"], p["q_raw"], c2w.to(DEV), *cam)
    (img * w.to(DEV)).sum().backward()
    util.check_image(img.detach().cpu().numpy(), ref, cal=img32)
    for k in util.PARAMS:
        util.check_grad(p[k].grad.cpu().numpy(), g64[k], k, cal=g32[k] if calibrate else None)
    return img


def test_equal_depths_are_ordered_by_index(gs):
    """Every Gaussian at the same camera depth: the per-list distribution sort sees one dense bucket and takes its exact
    fallback, and the order inside a list must be the Gaussian index.  Lists of 1000+ entries."""
    n, H, W, f = 3000, 32, 48, 40.0
    g = torch.Generator().manual_seed(21)
    z = torch.full((n,), 4.0, dtype=torch.float64)
    uv = torch.stack([torch.rand(n, generator=g, dtype=torch.float64) * (W - 8) + 4,
                      torch.rand(n, generator=g, dtype=torch.float64) * (H - 8) + 4], 1)
    pos = torch.stack([(uv[:, 0] - W / 2) / f * z, (uv[:, 1] - H / 2) / f * z, z], 1).float()
    assert float((pos[:, 2] - 4.0).abs().max()) == 0.0
    s = dict(pos=pos, scale_raw=torch.randn(n, 3, generator=g) * 0.2 - 1.2, q_raw=torch.randn(n, 4, generator=g),
             opacity_raw=torch.randn(n, generator=g) * 0.3 - 3.4, f_dc=torch.randn(n, 3, generator=g),
             f_rest=torch.randn(n, 45, generator=g) * 0.2)
    w = torch.rand(H, W, 3, generator=g)
    # (no float32 calibration here: with every depth equal the float32 oracle's order is its argsort's whim)
    _vs_oracle(gs, s, (H, W, f, f, W / 2.0, H / 2.0), torch.eye(4), w, z_order_eps=1e-10, calibrate=False)
    assert gs.render_stats()[2] > 0


def test_huge_gaussians_cover_many_lists(gs):
    """Gaussians whose rectangle spans more than 32 half-tile lists are binned by the whole wave (for_each_list)."""
    n, H, W, f = 400, 120, 160, 100.0
    g = torch.Generator().manual_seed(22)
    z = torch.rand(n, generator=g, dtype=torch.float64) * 3 + 3
    zs, order = torch.sort(z)
    keep = torch.ones(n, dtype=torch.bool)
    keep[order[1:][(zs[1:] - zs[:-1]) < 2e-5]] = False
    uv = torch.stack([torch.rand(n, generator=g, dtype=torch.float64) * W, torch.rand(n, generator=g, dtype=torch.float64) * H], 1)
    pos = torch.stack([(uv[:, 0] - W / 2) / f * z, (uv[:, 1] - H / 2) / f * z, z], 1).float()
    scale = torch.randn(n, 3, generator=g) * 0.3 - 2.0
    scale[::8] += 2.3                                  # every 8th: sigma ~ 1.3 world units = 30+ pixels -> 75+ px radius
    s = dict(pos=pos, scale_raw=scale, q_raw=torch.randn(n, 4, generator=g), opacity_raw=torch.randn(n, generator=g) - 1.0,
             f_dc=torch.randn(n, 3, generator=g), f_rest=torch.randn(n, 45, generator=g) * 0.2)
    s = {k: v[keep].contiguous() for k, v in s.items()}
    w = torch.rand(H, W, 3, generator=g)
    _vs_oracle(gs, s, (H, W, f, f * 1.1, W / 2.0 + 1.5, H / 2.0 - 2.0), torch.tensor(scenes.orbit_c2w(0, 8)), w)


def test_factored_sh_gradient_exchange_matches_the_plain_backward(gs):
    """dp.FactoredExchange in one process, two views: logit gradients + gsplat_sh_accumulate must give the gradients of
    the ordinary backward summed over the views, and sh_accumulate must match the oracle's SH basis."""
    import importlib
    dp = importlib.import_module("3d-gaussian-splatting-for-novel-view-synthesis_amd.dp")
    ops = importlib.import_module("3d-gaussian-splatting-for-novel-view-synthesis_amd.ops")
    d = util.load("g1_generic")
    rng = np.random.default_rng(9)
    cams = [torch.tensor(d["c2w"], device=DEV), torch.tensor(scenes._camera(rng), device=DEV)]
    ws = [torch.tensor(d["wrand"], device=DEV), torch.rand(d["H"], d["W"], 3, device=DEV)]
    names = ("pos", "f_dc", "f_rest", "opacity_raw", "scale_raw", "q_raw")

    def run(exchange):
        p = util.tensors(d, F32, device=DEV, grad=True)
        ex = dp.FactoredExchange(p, world_views=2) if exchange else None
        if ex is not None:
            ex.__enter__()
        for c2w, w in zip(cams, ws):
            img = gs.render_gaussians(*[p[k] for k in names], c2w, *util.cam_args(d), **d["kwargs"])
            (img * w).sum().backward()
        if ex is not None:
            ex.__exit__(None, None, None)
            assert p["f_dc"].grad is None and p["f_rest"].grad is None and len(ex.logits) == 2
            ex.finish()
        else:
            dp.allreduce_gradients([p[k].grad for k in names], world_views=2)
        return {k: p[k].grad.detach().cpu().numpy() for k in names}

    plain, fact = run(False), run(True)
    for k in names:
        scale = np.abs(plain[k]).max()
        assert np.abs(fact[k] - plain[k]).max() <= 2e-5 * scale, (k, np.abs(fact[k] - plain[k]).max(), scale)
    # the accumulate kernel on its own against the oracle's basis
    n, v = 333, 3
    g = torch.Generator().manual_seed(4)
    pos, eyes, logits = torch.randn(n, 3, generator=g), torch.randn(v, 3, generator=g) * 3, torch.randn(v, n, 3, generator=g)
    logits[1, ::5] = 0.0                                 # Gaussians a view did not bin
    g_dc, g_rest = ops.sh_accumulate(pos.to(DEV), eyes.to(DEV), logits.to(DEV), 0.5)
    acc = torch.zeros(n, 16, 3, dtype=torch.float64)
    for k in range(v):
        dd = pos.double() - eyes[k].double()
        dd = dd / (dd.norm(dim=-1, keepdim=True) + 1e-8)
        acc += tp.sh_basis(dd).unsqueeze(-1) * logits[k].double().unsqueeze(1)
    acc *= 0.5
    assert (g_dc.cpu().double() - acc[:, 0, :]).abs().max() < 1e-5
    assert (g_rest.cpu().double() - acc[:, 1:, :].transpose(1, 2).reshape(n, 45)).abs().max() < 1e-5


# (207, 339: the two of 400 further seeds that round 2 left outside the bounds; 794: the one of 769 that round 3's sweep found -- a
#  needle whose conic lost four digits in a d - b^2, fixed in gs_math.h; asserted with the general bounds)
@pytest.mark.parametrize("seed", list(range(10)) + list(range(100, 150)) + [207, 339, 794])
def test_random_scenes_vs_oracle(gs, seed):
    """Randomised image sizes, cameras, anisotropies and opacities against the float64 oracle: exercises ragged list grids,
    partial coarse bins, masks of thin rotated ellipses, chunk and group boundaries of the binning and raster kernels."""
    rng = np.random.default_rng(1000 + seed)
    H, W = int(rng.integers(9, 150)), int(rng.integers(9, 200))
    n = int(rng.integers(1, 1800))
    f = float(rng.uniform(40, 160))
    cam = (H, W, f, f * float(rng.uniform(0.9, 1.1)), W / 2 + float(rng.uniform(-5, 5)), H / 2 + float(rng.uniform(-5, 5)))
    c2w = torch.tensor(scenes._camera(rng, tilt=0.3))
    s = scenes._base(rng, n, H, W, cam[2], cam[3], cam[4], cam[5], mu_s=float(rng.uniform(-3.2, -1.2)), sd_s=float(rng.uniform(0.2, 1.0)),
                     op_mu=float(rng.uniform(-2, 3)), op_sd=1.5, spread=1.3, c2w=c2w.numpy())
    t = {k: torch.tensor(s[k]) for k in util.PARAMS}
    # drop near depth ties (fp32 cannot order them; the reference's argsort is unstable there)
    zc = tp.to_camera(t["pos"].double(), c2w.double())[2]
    zs, order = torch.sort(zc)
    keep = torch.ones(n, dtype=torch.bool)
    keep[order[1:][(zs[1:] - zs[:-1]) < 2e-5]] = False
    t = {k: v[keep].contiguous() for k, v in t.items()}
    w = torch.tensor(rng.uniform(0, 1, (H, W, 3)).astype(np.float32))
    p64 = {k: v.double().requires_grad_(True) for k, v in t.items()}
    try:
        ref = tp.render_fused(p64["pos"], p64["f_dc"], p64["f_rest"], p64["opacity_raw"], p64["scale_raw"], p64["q_raw"], c2w.double(), *cam)
    except Exception as e:                                  # all off-screen: the HIP path must raise the same
        with pytest.raises(Exception, match=str(e)):
            gs.render_gaussians(*[t[k].to(DEV) for k in ("pos", "f_dc", "f_rest", "opacity_raw", "scale_raw", "q_raw")], c2w.to(DEV), *cam)
        return
    (ref * w.double()).sum().backward()
    p = {k: v.to(DEV).requires_grad_(True) for k, v in t.items()}
    img = gs.render_gaussians(p["pos"], p["f_dc"], p["f_rest"], p["opacity_raw"], p["scale_raw"], p["q_raw"], c2w.to(DEV), *cam)
    (img * w.to(DEV)).sum().backward()
    # every bound beyond SURVEY 8c's is K_CAL x what the oracle in float32 (the reference's own fp32 arithmetic) does on this seed
    img32, g32 = _oracle(t, cam, c2w, w, torch.float32)
    known = KNOWN_CORNERS.get(seed, {})
    util.check_image(img.detach().cpu().numpy(), ref.detach().numpy(), cal=img32, what=f"seed {seed} image", frac=known.get("image_bulk_frac"))
    band = None
    if known.get("flip_band"):
        # float64 gradients with the chi-square clip moved by -+ 1e-4 relative: what flipping the pixels that sit within fp32's
        # reach of the clip does to every gradient entry (fp32 evaluates q of a needle-shaped Gaussian to ~5e-5 relative)
        lo = _oracle(t, cam, c2w, w, torch.float64, chi_square_clip=6.25 * (1 - known["flip_band"]))[1]
        hi = _oracle(t, cam, c2w, w, torch.float64, chi_square_clip=6.25 * (1 + known["flip_band"]))[1]
        band = {k: np.abs(hi[k] - lo[k]) for k in util.PARAMS}
    for k in util.PARAMS:
        g64 = p64[k].grad.numpy()
        if np.abs(g64).max() > 0:
            util.check_grad(p[k].grad.cpu().numpy(), g64, k, cal=g32[k], band=None if band is None else band[k])
        else:
            assert float(p[k].grad.abs().max()) == 0.0


# The two seeds of 2 x 400 further ones (tests/stress_random_scenes.py, profiles/r02_final_stress400_summary.txt) that lie outside
# the general bounds, kept here with their cause and their own stated bound:
#   339  ONE pixel (76, 62) at the far end of a needle-shaped Gaussian (2D eigenvalues 2.7 and 4563, 2.5 sigma = 169 px) has
#        q = 6.25 within fp32's reach and is inside the clip in float64, outside in the kernel (the float32 oracle happens to keep
#        it); at the far end of a needle a pixel weighs ~ (169 px)^2 in the second moments, and the rotation gradient of a needle is
#        the tiny off-diagonal of dL/dcov2d times the eigenvalue gap: 1.5 % of that Gaussian's q_raw gradient
#        (tools/moments_check.py 339 801).  Bound: the general one outside the band that moving the clip by -+ 1e-4 opens.
#   207  a 42 x 95 image with 6 of 11 970 values between 1.0e-5 and 1.8e-5 (no flip: q of large thin Gaussians carries ~1e-4
#        relative noise in fp32 -- terms of ~2000 cancel to <= 6.25 -- which the CPU reference's double-accumulating sums partly
#        hide).  Bound: bulk fraction 6e-4 at 1e-5, every value below 5e-3 as everywhere.
KNOWN_CORNERS = {339: {"flip_band": 1e-4}, 207: {"image_bulk_frac": 1 - 6e-4}}


def test_render_frames_is_the_frame_by_frame_result(gs):
    """The two-stream software pipeline of render_frames must return exactly the images of render_gaussians."""
    d = util.load("g1_generic")
    rng = np.random.default_rng(17)
    p = util.tensors(d, F32, device=DEV)
    names = ("pos", "f_dc", "f_rest", "opacity_raw", "scale_raw", "q_raw")
    cams = [torch.tensor(d["c2w"], device=DEV)] + [torch.tensor(scenes._camera(rng), device=DEV) for _ in range(6)]
    behind = np.eye(4, dtype=np.float32)
    behind[2, 3] = 50.0                                   # camera beyond the scene: everything culled -> zero image
    cams.insert(3, torch.tensor(behind, device=DEV))
    with torch.no_grad():
        ref = [gs.render_gaussians(*[p[k] for k in names], c, *util.cam_args(d), **d["kwargs"]) for c in cams]
        got = gs.render_frames(*[p[k] for k in names], cams, *util.cam_args(d), **d["kwargs"])
        seen = []
        assert gs.render_frames(*[p[k] for k in names], cams, *util.cam_args(d), **d["kwargs"],
                                on_frame=lambda k, im: seen.append((k, im.clone()))) is None
    torch.cuda.synchronize()
    assert len(got) == len(ref) == len(seen) and [k for k, _ in seen] == list(range(len(cams)))
    assert float(ref[3].abs().max()) == 0.0
    for a, b, (_, c) in zip(ref, got, seen):
        assert torch.equal(a, b) and torch.equal(a, c)
    assert gs.render_frames(*[p[k] for k in names], [], *util.cam_args(d)) == []


def test_no_grad_with_parameters_saves_nothing_for_backward(gs, monkeypatch):
    """Evaluation renders of a model's nn.Parameters under torch.no_grad() (render_trained.py:323): needs_input_grad is still
    True there, but no accum / grad2d buffer may be allocated or written."""
    import importlib
    ops = importlib.import_module("3d-gaussian-splatting-for-novel-view-synthesis_amd.ops")
    d = util.load("g1_generic")
    p = {k: torch.nn.Parameter(v) for k, v in util.tensors(d, F32, device=DEV).items()}
    seen = []
    real = ops._forward_impl
    monkeypatch.setattr(ops, "_forward_impl", lambda *a: (seen.append(a[-1]), real(*a))[1])
    args = (p["pos"], p["f_dc"], p["f_rest"], p["opacity_raw"], p["scale_raw"], p["q_raw"], torch.tensor(d["c2w"], device=DEV), *util.cam_args(d))
    with torch.no_grad():
        a = gs.render_gaussians(*args)
    b = gs.render_gaussians(*args)
    assert seen == [False, True] and torch.equal(a, b.detach()) and b.requires_grad and not a.requires_grad


def test_near_depth_ties_stay_in_and_tie_insensitive_results_match(gs):
    """Nothing is dropped here: thousands of Gaussians whose camera depths differ by less than fp32 resolves (the reference's
    argsort is not even stable there; the HIP order is (fp32 depth, index)).  What does NOT depend on how a tie is broken must
    still match the float64 oracle: the image on every pixel that no two members of a tie cluster both cover, and all six
    gradients of a loss that weights only those pixels."""
    n, H, W, f = 12000, 96, 144, 120.0
    g = torch.Generator().manual_seed(31)
    c2w = torch.tensor(scenes.orbit_c2w(1, 24))                                    # rotated: the fp32 depth is really rounded
    zc = 4.0 + torch.rand(n, generator=g, dtype=torch.float64) * 1.0               # mean gap 8e-5: 4 in 10 Gaussians have a near tie
    uv = torch.stack([torch.rand(n, generator=g, dtype=torch.float64) * W, torch.rand(n, generator=g, dtype=torch.float64) * H], 1)
    cam_pts = torch.stack([(uv[:, 0] - W / 2) / f * zc, (uv[:, 1] - H / 2) / f * zc, zc], 1)
    pos = (cam_pts @ c2w[:3, :3].double().t() + c2w[:3, 3].double()).float()
    s = dict(pos=pos, scale_raw=torch.randn(n, 3, generator=g) * 0.3 - 2.6, q_raw=torch.randn(n, 4, generator=g),
             opacity_raw=torch.randn(n, generator=g) - 0.5, f_dc=torch.randn(n, 3, generator=g), f_rest=torch.randn(n, 45, generator=g) * 0.2)
    cam = (H, W, f, f, W / 2.0, H / 2.0)
    stages = {}
    with torch.no_grad():
        tp.render_fused(*[s[k].double() for k in ("pos", "f_dc", "f_rest", "opacity_raw", "scale_raw", "q_raw")], c2w.double(), *cam, stages=stages)
    ids, u, v, con = stages["ids"], stages["u"].numpy(), stages["v"].numpy(), stages["conic"].numpy()
    z = tp.to_camera(s["pos"].double()[ids], c2w.double())[2].numpy()              # ascending: the oracle's depth order
    assert (np.diff(z) >= 0).all()
    new_cluster = np.concatenate([[True], np.diff(z) >= 2e-5])
    cluster = np.cumsum(new_cluster) - 1
    sizes = np.bincount(cluster)
    tied = np.nonzero(sizes[cluster] > 1)[0]
    assert len(tied) > 0.25 * len(z)                                               # ties are common in this scene
    cover = {}                                                                     # cluster -> per-pixel count of covering members
    masked = np.zeros((H, W), bool)
    ys, xs = np.mgrid[0:H, 0:W]
    for i in tied:
        du, dv = xs - u[i], ys - v[i]
        inside = con[i, 0] * du * du + 2 * con[i, 1] * du * dv + con[i, 2] * dv * dv <= 6.25 * 1.001 + 1e-3
        c = cover.setdefault(int(cluster[i]), np.zeros((H, W), np.int32))
        c += inside
    for c in cover.values():
        masked |= c >= 2
    print(f"{len(tied)} of {len(z)} visible Gaussians sit in near-tie clusters; {int(masked.sum())} of {H * W} pixels are covered by two of one cluster")
    assert 0 < masked.sum() < 0.5 * H * W
    w = torch.rand(H, W, 3, generator=g) * torch.tensor(~masked).unsqueeze(-1)
    ref, g64 = _oracle(s, cam, c2w, w, torch.float64)
    img32, g32 = _oracle(s, cam, c2w, w, torch.float32)
    p = {k: t.to(DEV).requires_grad_(True) for k, t in s.items()}
    img = gs.render_gaussians(p["pos"], p["f_dc"], p["f_rest"], p["opacity_raw"], p["scale_raw"], p["q_raw"], c2w.to(DEV), *cam)
    (img * w.to(DEV)).sum().backward()
    got = img.detach().cpu().numpy()
    util.check_image(got[~masked], ref[~masked], cal=img32[~masked], what="image off the tied pixels")
    for k in util.PARAMS:
        util.check_grad(p[k].grad.cpu().numpy(), g64[k], k, cal=g32[k])


def _ops():
    import importlib
    return importlib.import_module("3d-gaussian-splatting-for-novel-view-synthesis_amd.ops")


def _cap_key(ops, d):
    """Where ops keeps the pair capacity of golden case d's frames: per (device, image size, Gaussian-count bucket)."""
    import types
    return ops.capacity_key(torch.device("cuda:0"), types.SimpleNamespace(H=d["H"], W=d["W"]), len(d["pos"]))


def test_deferred_checks_render_the_same_frame_without_waiting(gs):
    """ops.deferred_checks(): buffers from the capacity kept from earlier frames, counters read once in verify(), SH colour
    inside the projection kernel -- the image is bit-identical to the waiting path, the gradients agree to summation order."""
    ops = _ops()
    d = util.load("g1_generic")
    a, pa = _fused(gs, d)                                        # waits; leaves a pair capacity for the device
    before = dict(ops.forward_modes)
    with ops.deferred_checks() as chk:
        b, pb = _fused(gs, d)
        c, _ = _fused(gs, d, grad=False)
    assert ops.forward_modes["deferred"] == before["deferred"] + 2 and ops.forward_modes["waited"] == before["waited"]
    counts = chk.verify()
    assert len(counts) == 2 and counts[0].n_visible == len(d["im_ids"]) and counts[0].n_pairs == len(d["im_pair_gauss"])
    assert torch.equal(a, b) and torch.equal(a, c)
    for k in util.PARAMS:
        assert float((pa[k].grad - pb[k].grad).abs().max()) <= 1e-5 * float(pa[k].grad.abs().max()), k
    util.check_image(b.detach().cpu().numpy(), d["image"], cal=d["image_f32"])


def test_deferred_checks_report_overflow_offscreen_and_empty_scenes(gs):
    ops = _ops()
    d = util.load("g1_generic")
    _fused(gs, d, grad=False)
    key = _cap_key(ops, d)
    real = ops._ws.capacity[key]
    try:
        ops._ws.capacity[key] = 64                              # far fewer pairs than the frame has (1616)
        with ops.deferred_checks() as chk:
            img, p = _fused(gs, d)                              # memory-safe garbage
        torch.cuda.synchronize()
        with pytest.raises(ops.PairCapacityExceeded):
            chk.verify()
        assert ops._ws.capacity[key] >= len(d["im_pair_gauss"]) # raised: the repeat fits
        with ops.deferred_checks() as chk:
            img, p = _fused(gs, d)
        chk.verify()
        util.check_image(img.detach().cpu().numpy(), d["image"], cal=d["image_f32"])
        for k in util.PARAMS:
            util.check_grad(p[k].grad.cpu().numpy(), d["grad_" + k], k, cal=d["grad32_" + k])
    finally:
        ops._ws.capacity[key] = max(real, ops._ws.capacity[key])
    off = util.load("g10_offscreen")
    with ops.deferred_checks() as chk:
        _fused(gs, off, grad=False)                             # the reference raises here; deferred: in verify()
    with pytest.raises(Exception, match=str(off["raises"])):
        chk.verify()
    for name in util.EMPTY_CASES:                               # nothing survives: zero image, zero gradients, no exception
        e = util.load(name)
        with ops.deferred_checks() as chk:
            img, p = _fused(gs, e)
        assert chk.verify()[0].n_survivors == 0 and float(img.detach().abs().max()) == 0.0
        for k in util.PARAMS:
            assert float(p[k].grad.abs().max()) == 0.0


@pytest.mark.parametrize("name", ["g1_generic", "g3_occlusion", "g6_huge"])
def test_deterministic_backward_matches_the_goldens_and_repeats_bitwise(gs, name):
    """Deterministic mode on the reference's goldens: occlusion (lists that saturate early leave rows unwritten -> must read as
    zero) and huge Gaussians (rectangles of more than 32 lists: the un-masked slot rule)."""
    d = util.load(name)
    old = gs.set_deterministic(True)
    try:
        _, p1 = _fused(gs, d)
        _, p2 = _fused(gs, d)
    finally:
        gs.set_deterministic(old)
    for k in util.PARAMS:
        assert torch.equal(p1[k].grad, p2[k].grad), k
        util.check_grad(p1[k].grad.cpu().numpy(), d["grad_" + k], k, cal=d["grad32_" + k])


@pytest.mark.parametrize("name", ["g1_generic", "g2_ragged", "g6_huge"])
def test_backward_from_the_saved_sh_jacobian_is_the_backward_from_the_coefficients(gs, name):
    """A forward pass that will be differentiated leaves d colour / d logit and d logit / d position (48 B per Gaussian) in
    project_state (GSPLAT_PROJECT_SAVE_SH_JACOBIAN), and the backward takes them instead of reading the 192 B of SH coefficients
    again (GSPLAT_BACKWARD_SH_JACOBIAN).  Same gradients as the coefficient path up to fp32 rounding, both within the golden
    tolerances; deterministic mode so that the two runs differ by nothing else."""
    ops = _ops()
    d = util.load(name)
    old = gs.set_deterministic(True)
    try:
        assert ops._sh_jacobian
        img_j, pj = _fused(gs, d)
        ops._sh_jacobian = False
        img_c, pc = _fused(gs, d)
    finally:
        ops._sh_jacobian = True
        gs.set_deterministic(old)
    assert float((img_j - img_c).abs().max()) <= 1e-6            # (two instantiations of the colour code: fp contraction may differ)
    for k in util.PARAMS:
        a, b = pj[k].grad, pc[k].grad
        scale = float(b.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= 2e-5 * scale, (k, float((a - b).abs().max()), scale)
        util.check_grad(a.cpu().numpy(), d["grad_" + k], k, cal=d["grad32_" + k])


def test_counters_arrive_by_copy_or_by_mapped_store(gs):
    """gsplat_project hands the counters to the host either with a copy operation (any host memory, flags = 0: what the
    INTEGRATION.md stub does) or by storing them itself into device-mapped pinned memory (GSPLAT_PROJECT_COUNTS_MAPPED); with
    or without the SH colour inside the projection kernel, totalled by the projection kernel or (COUNTS_LATE) by the first
    binning kernel, the counters are the same, and the counter block is left zeroed."""
    import ctypes as C
    import importlib
    abi = importlib.import_module("3d-gaussian-splatting-for-novel-view-synthesis_amd._abi")
    lib = abi.lib()
    d = util.load("g1_generic")
    p = util.tensors(d, F32, device=DEV)
    n = len(d["pos"])
    view = abi.make_view(*util.cam_args(d))
    g = abi.Gaussians(n, p["pos"].data_ptr(), p["opacity_raw"].data_ptr(), None, None, p["scale_raw"].data_ptr(), p["q_raw"].data_ptr(),
                      p["f_dc"].data_ptr(), p["f_rest"].data_ptr())
    c2w = torch.tensor(d["c2w"], device=DEV)
    state = torch.empty(lib.gsplat_project_state_bytes(n, C.byref(view)), dtype=torch.uint8, device=DEV)
    block = torch.zeros(lib.gsplat_project_scratch_bytes(n), dtype=torch.uint8, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    got = []
    M, F, L, J = (abi.GSPLAT_PROJECT_COUNTS_MAPPED, abi.GSPLAT_PROJECT_COLOUR_FUSED, abi.GSPLAT_PROJECT_COUNTS_LATE,
                  abi.GSPLAT_PROJECT_SAVE_SH_JACOBIAN)
    # (COUNTS_LATE: the first binning kernel, queued by the same call, totals the counters instead of the projection's last wave)
    for flags in (0, M, M | F, F, L, M | L, M | F | L, M | F | L | J, F | J):
        host = torch.full((C.sizeof(abi.Counts),), 255, dtype=torch.uint8).pin_memory()
        abi.check(lib.gsplat_project(C.byref(g), C.c_void_p(c2w.data_ptr()), C.byref(view), C.c_void_p(state.data_ptr()),
                                     C.c_void_p(block.data_ptr()), block.numel(), C.c_void_p(host.data_ptr()), None, flags, st), "gsplat_project")
        torch.cuda.synchronize()
        c = abi.Counts.from_buffer_copy(host.numpy().tobytes())
        got.append((c.n_survivors, c.n_visible, c.n_pairs, c.n_binned, c.max_tiles_per_gaussian))
        assert int(block.max()) == 0, "the counter block must be left zeroed"
    assert len(set(got)) == 1 and got[0][1] == len(d["im_ids"]) and got[0][2] == len(d["im_pair_gauss"]), got


def test_deterministic_backward_with_too_small_buffers_is_memory_safe(gs):
    """Deterministic mode + a deferred frame that outgrows its buffers: rows whose slot lies beyond the capacity are dropped, not
    written out of bounds; verify() reports the overflow and the repeat is right (and bitwise repeatable)."""
    ops = _ops()
    d = util.load("g1_generic")
    _fused(gs, d, grad=False)
    key = _cap_key(ops, d)
    real = ops._ws.capacity[key]
    old = gs.set_deterministic(True)
    try:
        ops._ws.capacity[key] = 200
        with ops.deferred_checks() as chk:
            _fused(gs, d)
        torch.cuda.synchronize()
        with pytest.raises(ops.PairCapacityExceeded):
            chk.verify()
        (_, p1), (_, p2) = gs.run_deferred(lambda: (_fused(gs, d), _fused(gs, d)))
        for k in util.PARAMS:
            assert torch.equal(p1[k].grad, p2[k].grad), k
            util.check_grad(p1[k].grad.cpu().numpy(), d["grad_" + k], k, cal=d["grad32_" + k])
    finally:
        gs.set_deterministic(old)
        ops._ws.capacity[key] = max(real, ops._ws.capacity[key])


def test_the_ctypes_stub_of_integration_md_renders_the_golden(gs):
    """INTEGRATION.md shows the binding a maintainer of the reference would write against include/gsplat_mi355x.h: run that very
    code (only the library path is filled in) on the un-fused golden."""
    import os
    import re
    import importlib
    abi = importlib.import_module("3d-gaussian-splatting-for-novel-view-synthesis_amd._abi")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    code = next(b for b in blocks if "def render(" in b and "C.CDLL" in b)
    code = code.replace('C.CDLL("libgsplat_mi355x.so")', f'C.CDLL({abi.LIB_PATH!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    d = util.load("g11_unfused")
    t = {k: torch.tensor(d[k], dtype=F32, device=DEV) for k in ("pos", "opacity_raw")}
    col = torch.tensor(d["color_in"], dtype=F32, device=DEV)
    sig = torch.tensor(d["sigma_in"], dtype=F32, device=DEV)
    img = ns["render"](t["pos"], col, t["opacity_raw"], sig, torch.tensor(d["c2w"], device=DEV), *util.cam_args(d))
    torch.cuda.synchronize()
    util.check_image(img.cpu().numpy(), d["image"])
    again = gs.render(t["pos"], col, t["opacity_raw"], sig, torch.tensor(d["c2w"], device=DEV), *util.cam_args(d))
    assert torch.equal(img, again)


def test_composite_entries_queue_the_same_frame_as_the_separate_calls(gs):
    """gsplat_forward_deferred / gsplat_backward (ONE library call per direction on ONE arena) against the separate calls a deferred
    frame used to make: the same kernels on the same inputs -- bit-identical image, and in deterministic mode bit-identical
    gradients; a second backward pass through the same frame (retain_graph) is right too."""
    ops = _ops()
    d = util.load("g1_generic")
    _fused(gs, d, grad=False)                                    # leaves a pair capacity
    old = gs.set_deterministic(True)
    try:
        res = {}
        for composite in (True, False):
            ops._composite = composite
            before = dict(ops.composite_calls)
            with ops.deferred_checks() as chk:
                img, p = _fused(gs, d)
                img_ng, _ = _fused(gs, d, grad=False)
            chk.verify()
            took = (ops.composite_calls["forward"] - before["forward"], ops.composite_calls["backward"] - before["backward"])
            assert took == ((2, 1) if composite else (0, 0))
            res[composite] = (img.detach().clone(), img_ng.clone(), {k: p[k].grad.clone() for k in util.PARAMS})
        assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1]) and torch.equal(res[True][0], res[True][1])
        for k in util.PARAMS:
            assert torch.equal(res[True][2][k], res[False][2][k]), k
            util.check_grad(res[True][2][k].cpu().numpy(), d["grad_" + k], k, cal=d["grad32_" + k])
        # two backward passes through one composite frame: the second must not add to the first one's sums
        ops._composite = True
        with ops.deferred_checks() as chk:
            p = util.tensors(d, F32, "cuda", grad=True)
            img = gs.render_gaussians(p["pos"], p["f_dc"], p["f_rest"], p["opacity_raw"], p["scale_raw"], p["q_raw"],
                                      torch.tensor(d["c2w"], dtype=F32, device="cuda"), *util.cam_args(d), **d["kwargs"])
            w = torch.tensor(d["wrand"], dtype=F32, device="cuda")
            (img * w).sum().backward(retain_graph=True)
            first = {k: p[k].grad.clone() for k in util.PARAMS}
            for k in util.PARAMS:
                p[k].grad = None
            (img * w).sum().backward()
        chk.verify()
        for k in util.PARAMS:
            assert torch.equal(first[k], p[k].grad), k
    finally:
        ops._composite = True
        gs.set_deterministic(old)


def test_a_deferred_block_left_without_verify_never_hands_out_an_unread_counter_block(gs):
    """DeferredChecks robustness: (1) a block left by an exception reads its frames' counters then and there and raises nothing of
    its own; (2) a block that is simply never verified keeps its pinned counter blocks until the ring comes round, and then they are
    read -- by their owner -- BEFORE another frame gets them: the late verify() still reports that block's own frames."""
    ops = _ops()
    d = util.load("g1_generic")
    _fused(gs, d, grad=False)
    with pytest.raises(ZeroDivisionError):
        with ops.deferred_checks() as chk:
            _fused(gs, d, grad=False)
            1 / 0
    assert not chk.pending and len(chk.counts) == 1 and chk.counts[0].n_visible == len(d["im_ids"])
    off = util.load("g10_offscreen")
    _ = _cap_key(ops, off)
    with ops.deferred_checks() as abandoned:                     # never verified (yet)
        _fused(gs, d, grad=False)
        _fused(gs, off, grad=False) if ops._ws.pair_capacity(_cap_key(ops, off)) else None
    n_pending = len(abandoned.pending)
    assert n_pending >= 1
    for _ in range(3):                                           # more frames than the ring has slots: it comes round
        with ops.deferred_checks() as chk:
            for _ in range(ops.PINNED_SLOTS // 2 - 8):
                _fused(gs, d, grad=False)
        assert all(c.n_visible == len(d["im_ids"]) for c in chk.verify())
    assert not abandoned.pending and len(abandoned.counts) == n_pending          # read when the ring reached them, not overwritten
    assert abandoned.counts[0].n_visible == len(d["im_ids"]) and abandoned.counts[0].n_pairs == len(d["im_pair_gauss"])


def test_pair_capacity_is_kept_per_image_size_and_gaussian_count(gs):
    """A large scene must not make every later frame of a small one pay for its pair capacity (buffers, grids, the deterministic
    mode's clear): capacities live per (device, image size, power-of-two bucket of N); reset_pair_capacity() forgets them."""
    ops = _ops()
    small, big = util.load("g1_generic"), util.load("g3_occlusion")
    ks, kb = _cap_key(ops, small), _cap_key(ops, big)
    _fused(gs, small, grad=False)
    cap_small = ops._ws.pair_capacity(ks)
    _fused(gs, big, grad=False)
    assert ks != kb and cap_small > 0 and ops._ws.pair_capacity(kb) > 0
    assert ops._ws.pair_capacity(ks) == cap_small                 # the other scene's pairs left it alone
    with ops.deferred_checks() as chk:
        img, _ = _fused(gs, small, grad=False)
    chk.verify()
    util.check_image(img.cpu().numpy(), small["image"], cal=small["image_f32"])
    saved = dict(ops._ws.capacity)
    try:
        ops.reset_pair_capacity()
        before = dict(ops.forward_modes)
        with ops.deferred_checks() as chk:                       # nothing known any more: this frame waits like an ordinary one
            _fused(gs, small, grad=False)
        chk.verify()
        assert ops.forward_modes["waited"] == before["waited"] + 1
    finally:
        ops._ws.capacity.update(saved)
