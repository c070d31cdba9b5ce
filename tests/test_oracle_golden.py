"""Pins the CPU oracle (oracle/torch_port.py) to fixtures produced by the real reference.

tests/golden/*.npz come from oracle/gen_golden.py, which imports the unmodified reference in the
build container.  In float64 the restatement must agree with the reference to rounding error:
image, the six parameter gradients and every per-stage intermediate.
"""
import numpy as np
import pytest
import torch

from oracle import torch_port as tp
from tests import util

F64 = torch.float64


def _run(d, dtype=F64, grad=True, stages=None):
    p = util.tensors(d, dtype, grad=grad)
    c2w = torch.tensor(d["c2w"], dtype=dtype)
    img = tp.render_fused(p["pos"], p["f_dc"], p["f_rest"], p["opacity_raw"], p["scale_raw"], p["q_raw"], c2w,
                          *util.cam_args(d), stages=stages, **d["kwargs"])
    if grad:
        (img * torch.tensor(d["wrand"], dtype=dtype)).sum().backward()
    return img.detach().numpy(), p


@pytest.mark.parametrize("name", util.RENDER_CASES)
def test_image_grads_f64(name):
    d = util.load(name)
    stages = {}
    img, p = _run(d, stages=stages)
    assert np.abs(img - d["image"]).max() < 1e-12
    for k in util.PARAMS:
        ref = d["grad_" + k]
        assert np.abs(p[k].grad.numpy() - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), k
    # per-stage intermediates
    assert np.array_equal(stages["ids"].numpy(), d["im_ids"])
    assert np.allclose(stages["u"].detach().numpy(), d["im_u"], rtol=0, atol=1e-10)
    assert np.allclose(stages["v"].detach().numpy(), d["im_v"], rtol=0, atol=1e-10)
    assert np.allclose(stages["cov2d"].detach().numpy(), d["im_cov2d"], rtol=1e-10, atol=1e-12)
    assert np.allclose(stages["evals"].detach().numpy(), d["im_evals"], rtol=1e-10, atol=1e-14)
    con = d["im_conic"]
    con = np.stack([con[:, 0, 0], con[:, 0, 1], con[:, 1, 1]], 1)
    # relative to each matrix's scale (det = ad - bc cancels for needle Gaussians)
    assert (np.abs(stages["conic"].detach().numpy() - con) <= 1e-7 * np.abs(con).max(1, keepdims=True)).all()
    assert np.array_equal(stages["tile_rect"].numpy(), d["im_tile_rect"])
    assert np.array_equal(stages["pair_gauss"].numpy(), d["im_pair_gauss"])
    assert np.array_equal(stages["tile_ids"].numpy(), d["im_tile_ids"])
    assert np.array_equal(stages["tile_start"].numpy(), d["im_tile_start"])
    assert np.array_equal(stages["tile_end"].numpy(), d["im_tile_end"])


@pytest.mark.parametrize("name", ["g1_generic", "g4_thresholds", "g7_tiny"])
def test_image_f32_matches_reference_f32(name):
    """Same op order as the reference, so the fp32 run reproduces its fp32 image to a few ulp."""
    d = util.load(name)
    img, _ = _run(d, dtype=torch.float32, grad=False)
    assert np.abs(img - d["image_f32"]).max() < 2e-6


@pytest.mark.parametrize("name", util.EMPTY_CASES)
def test_empty_is_zero_image_zero_grads(name):
    d = util.load(name)
    img, p = _run(d)
    assert img.shape == (d["H"], d["W"], 3) and np.abs(img).max() == 0
    for k in util.PARAMS:
        g = p[k].grad
        assert g is None or float(g.abs().max()) == 0.0


def test_offscreen_raises():
    d = util.load("g10_offscreen")
    with pytest.raises(Exception, match=str(d["raises"])):
        _run(d, grad=False)


def test_unfused_boundary():
    d = util.load("g11_unfused")
    t = {k: torch.tensor(d[k], dtype=F64).requires_grad_(True) for k in ("pos", "opacity_raw")}
    col = torch.tensor(d["color_in"], dtype=F64).requires_grad_(True)
    sig = torch.tensor(d["sigma_in"], dtype=F64).requires_grad_(True)
    img = tp.render(t["pos"], col, t["opacity_raw"], sig, torch.tensor(d["c2w"], dtype=F64), *util.cam_args(d))
    (img * torch.tensor(d["wrand"])).sum().backward()
    assert np.abs(img.detach().numpy() - d["image"]).max() < 1e-12
    for k, g in (("pos", t["pos"].grad), ("color", col.grad), ("opacity_raw", t["opacity_raw"].grad),
                 ("sigma", sig.grad)):
        ref = d["grad_" + k]
        assert np.abs(g.numpy() - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), k


def test_pieces():
    d = dict(np.load(util.GOLDEN + "/pieces.npz"))
    sr = torch.tensor(d["scale_raw"], dtype=F64, requires_grad=True)
    qr = torch.tensor(d["q_raw"], dtype=F64, requires_grad=True)
    sig = tp.covariance_from_params(sr, qr)
    (sig * torch.tensor(d["w_sigma"])).sum().backward()
    assert np.allclose(sig.detach().numpy(), d["sigma"], rtol=1e-12, atol=1e-14)
    assert np.allclose(sr.grad.numpy(), d["grad_scale_raw"], rtol=1e-10, atol=1e-12)
    assert np.allclose(qr.grad.numpy(), d["grad_q_raw"], rtol=1e-10, atol=1e-12)
    fd = torch.tensor(d["f_dc"], dtype=F64, requires_grad=True)
    fr = torch.tensor(d["f_rest"], dtype=F64, requires_grad=True)
    pt = torch.tensor(d["points"], dtype=F64, requires_grad=True)
    c2w = torch.tensor(d["c2w"], dtype=F64)
    col = tp.sh_colour(fd, fr, pt, c2w)
    (col * torch.tensor(d["w_col"])).sum().backward()
    assert np.allclose(col.detach().numpy(), d["color"], rtol=1e-12, atol=1e-14)
    assert np.allclose(fd.grad.numpy(), d["grad_f_dc"], rtol=1e-10, atol=1e-12)
    assert np.allclose(fr.grad.numpy(), d["grad_f_rest"], rtol=1e-10, atol=1e-12)
    assert np.allclose(pt.grad.numpy(), d["grad_points"], rtol=1e-9, atol=1e-11)
    assert np.allclose(tp.rotmat_from_quat(torch.tensor(d["q_raw"], dtype=F64)).numpy(), d["rot"], atol=1e-14)
    assert np.allclose(tp.inv2x2(torch.tensor(d["m2"], dtype=F64)).numpy(), d["inv2x2"], rtol=1e-12)
    assert np.allclose(np.array(tp.scale_intrinsics(540, 960, 1080, 1920, 1100.0, 1090.0, 961.5, 538.25)),
                       d["scale_intrinsics"])
    uv, x, y, z = tp.project_points(torch.tensor(d["points"], dtype=F64), c2w, 500.0, 510.0, 320.0, 240.0)
    assert np.allclose(uv.numpy(), d["proj_uv"], rtol=1e-10, atol=1e-9)
    assert np.allclose(torch.stack([x, y, z], 1).numpy(), d["proj_xyz"], rtol=1e-12, atol=1e-12)
    assert np.allclose(np.array(sorted(tp.SH_K)), np.array(sorted(d["harmonics_vals"])))


def test_config1_full_digest():
    """Config 1 (10k Gaussians, 256x256, f_rest = 0) at full size against the reference's digests."""
    from oracle import scenes
    d = dict(np.load(util.GOLDEN + "/g13_config1_full.npz"))
    s = scenes.synthetic_scene(1)
    dig = np.array([float(np.abs(s[k]).astype(np.float64).sum()) for k in util.PARAMS])
    if not np.allclose(dig, d["input_digest"], rtol=1e-12):
        pytest.skip("torch RNG stream differs from the one the fixture was generated with")
    p = util.tensors(s, F64, grad=True)
    stages = {}
    img = tp.render_fused(p["pos"], p["f_dc"], p["f_rest"], p["opacity_raw"], p["scale_raw"], p["q_raw"],
                          torch.tensor(s["c2w"], dtype=F64), s["H"], s["W"], s["fx"], s["fy"], s["cx"], s["cy"],
                          stages=stages)
    wr = np.random.default_rng(1).uniform(0, 1, (s["H"], s["W"], 3)).astype(np.float32).astype(np.float64)
    (img * torch.tensor(wr)).sum().backward()
    im = img.detach().numpy()
    assert len(stages["ids"]) == int(d["V"]) and len(stages["pair_gauss"]) == int(d["P"])
    assert np.abs(im.reshape(128, 2, 128, 2, 3).mean(axis=(1, 3)) - d["image_blockmean"]).max() < 1e-6
    assert np.abs(im[::8] - d["image_rows8"]).max() < 1e-6
    for k in util.PARAMS:
        g = p[k].grad.numpy()
        assert abs(np.linalg.norm(g) - float(d["gnorm_" + k])) <= 1e-8 * float(d["gnorm_" + k])
        assert np.allclose(g[:1024], d["grad_" + k + "_head"], rtol=1e-4, atol=1e-6 * np.abs(g).max())


def test_loss_vs_reference():
    d = dict(np.load(util.GOLDEN + "/loss.npz"))
    for tag in "abc":
        p = torch.tensor(d["pred_" + tag], dtype=F64, requires_grad=True)
        t = torch.tensor(d["target_" + tag], dtype=F64)
        total, l1, sl = tp.compute_loss(p, t, 0.8, 0.2)
        total.backward()
        assert np.allclose([float(l1), float(sl), float(total)], d["vals_" + tag], rtol=1e-12, atol=1e-14)
        assert np.abs(p.grad.numpy() - d["grad_" + tag]).max() < 1e-14
        p2 = torch.tensor(d["pred_" + tag], dtype=F64, requires_grad=True)
        t2, _, _ = tp.compute_loss(p2, t, 0.3, 1.7)
        t2.backward()
        assert abs(float(t2) - float(d["total2_" + tag])) < 1e-13
        assert np.abs(p2.grad.numpy() - d["grad2_" + tag]).max() < 1e-13
