"""Seeded scene builders shared by the golden generator, the tests and bench.py.

TEST INFRASTRUCTURE (lives under oracle/): nothing in the product package imports this
file.  bench.py has its own copy of `synthetic_scene` so that the product benchmark does
not depend on oracle/.

`synthetic_scene` is the benchmark scene of SURVEY.md §8(d); the `case_*` builders are the
small parity cases G1..G11 of SURVEY.md §8(c).  Every builder returns a dict of float32
numpy arrays (exactly representable, so the fp32 and fp64 reference runs see the same
inputs) plus the camera / kwargs scalars.
"""
import numpy as np
import torch

# config id -> (N, H, W, fx, mu_s)   SURVEY.md §8(d)
CONFIGS = {
    1: (10_000, 256, 256, 300.0, -3.0),
    2: (100_000, 800, 800, 800.0, -4.5),
    3: (1_000_000, 1080, 1920, 1100.0, -5.0),
    4: (3_000_000, 1080, 1920, 1100.0, -5.4),
    5: (10_000_000, 2160, 3840, 2200.0, -5.8),
    6: (100_000, 800, 800, 800.0, -2.0),       # not a BASELINE.json config: config 2 with the footprints of a trained scene (bench.py)
}


def synthetic_scene(config, n_override=None):
    """seed 0, draws in the fixed order pos, scale, quat, opacity, f_dc, f_rest."""
    N, H, W, fx, mu_s = CONFIGS[config]
    if n_override is not None:
        N = n_override
    g = torch.Generator().manual_seed(0)
    pos = torch.randn(N, 3, generator=g)
    pos[:, 2] += 5.0
    scale_raw = torch.randn(N, 3, generator=g) * 0.3 + mu_s
    q_raw = torch.randn(N, 4, generator=g)
    opacity_raw = torch.randn(N, generator=g)
    f_dc = torch.randn(N, 3, generator=g)
    f_rest = torch.randn(N, 45, generator=g) * 0.1
    if config == 1:
        f_rest = torch.zeros(N, 45)
    return dict(pos=pos.numpy(), scale_raw=scale_raw.numpy(), q_raw=q_raw.numpy(),
                opacity_raw=opacity_raw.numpy(), f_dc=f_dc.numpy(), f_rest=f_rest.numpy(),
                c2w=np.eye(4, dtype=np.float32), H=H, W=W, fx=fx, fy=fx, cx=W / 2.0, cy=H / 2.0,
                kwargs={})


def orbit_c2w(k, n_views=8, centre=(0.0, 0.0, 5.0)):
    """Config-4 views: the identity camera rotated about the scene centre by k*360/n degrees (y axis)."""
    a = 2.0 * np.pi * k / n_views
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    c = np.asarray(centre, dtype=np.float64)
    t = c - R @ c  # camera centre: origin rotated about `centre`
    m = np.eye(4)
    m[:3, :3] = R
    m[:3, 3] = t
    return m.astype(np.float32)


def _rot(ax, ay, az):
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _camera(rng, tilt=0.15):
    """A non-identity c2w: small rotation + translation, looking roughly down +z."""
    R = _rot(*(rng.uniform(-tilt, tilt, 3)))
    m = np.eye(4)
    m[:3, :3] = R
    m[:3, 3] = rng.uniform(-0.3, 0.3, 3)
    return m.astype(np.float32)


def _base(rng, N, H, W, fx, fy, cx, cy, *, depth=(2.5, 7.0), mu_s=-2.3, sd_s=0.45, op_mu=0.0, op_sd=1.5,
          rest_amp=0.3, spread=1.15, c2w=None):
    """Random Gaussians placed inside the frustum of the given camera (positions in world space)."""
    if c2w is None:
        c2w = _camera(rng)
    z = rng.uniform(depth[0], depth[1], N)
    u = rng.uniform(-0.5 * (spread - 1) * W, W * (1 + 0.5 * (spread - 1)), N)
    v = rng.uniform(-0.5 * (spread - 1) * H, H * (1 + 0.5 * (spread - 1)), N)
    pc = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], 1)
    pw = pc @ c2w[:3, :3].astype(np.float64).T + c2w[:3, 3].astype(np.float64)
    return dict(
        pos=pw.astype(np.float32),
        scale_raw=(rng.normal(mu_s, sd_s, (N, 3))).astype(np.float32),
        q_raw=rng.normal(0, 1, (N, 4)).astype(np.float32),
        opacity_raw=rng.normal(op_mu, op_sd, N).astype(np.float32),
        f_dc=rng.normal(0, 1, (N, 3)).astype(np.float32),
        f_rest=(rng.normal(0, 1, (N, 45)) * rest_amp).astype(np.float32),
        c2w=c2w, H=H, W=W, fx=float(fx), fy=float(fy), cx=float(cx), cy=float(cy), kwargs={})


def case_g1():  # generic: rotated+translated camera, fx != fy, off-centre principal point
    rng = np.random.default_rng(101)
    return _base(rng, 600, 64, 96, 85.0, 78.0, 50.5, 29.25)


def case_g2():  # H, W not multiples of 16
    rng = np.random.default_rng(102)
    return _base(rng, 450, 50, 70, 60.0, 64.0, 33.0, 26.0)


def case_g3():  # heavy occlusion: opaque layers so that T_i <= 5e-5 is reached
    rng = np.random.default_rng(103)
    return _base(rng, 700, 48, 48, 50.0, 50.0, 24.0, 24.0, mu_s=-1.2, sd_s=0.25, op_mu=5.0, op_sd=1.0,
                 depth=(2.0, 6.0))


def case_g4():  # alphas near the 1/128 cutoff, the 1/256 prefilter and the 0.99 cap
    rng = np.random.default_rng(104)
    s = _base(rng, 600, 64, 64, 70.0, 70.0, 32.0, 32.0, mu_s=-1.9)
    o = np.empty(600, dtype=np.float32)
    o[0::3] = rng.normal(-4.85, 0.35, 200)    # sigmoid ~ 0.0078 (cutoff) / 0.0039 (prefilter)
    o[1::3] = rng.normal(7.5, 1.5, 200)       # sigmoid -> 0.999 clamp, alpha cap 0.99 at the centre
    o[2::3] = rng.normal(0.0, 1.0, 200)
    s["opacity_raw"] = o
    return s


def case_g5():  # centres in the guard band and outside it, partly off-screen AABBs
    rng = np.random.default_rng(105)
    return _base(rng, 600, 64, 80, 75.0, 75.0, 40.0, 32.0, spread=2.1, mu_s=-1.8)


def case_g6():  # huge Gaussians: lambda > 1e4 (eigen clamp active, radius cap 250)
    rng = np.random.default_rng(106)
    s = _base(rng, 240, 64, 96, 90.0, 90.0, 48.0, 32.0, mu_s=-2.0, op_mu=-1.0)
    big = rng.choice(240, 60, replace=False)
    s["scale_raw"][big] = rng.normal(1.3, 0.5, (60, 3)).astype(np.float32)
    s["opacity_raw"][big] = rng.normal(-3.0, 0.5, 60).astype(np.float32)
    return s


def case_g7():  # tiny Gaussians: lambda < 1e-6 clamp, det clamp, min_conis
    rng = np.random.default_rng(107)
    s = _base(rng, 500, 48, 64, 60.0, 60.0, 32.0, 24.0, mu_s=-2.0)
    tiny = rng.choice(500, 200, replace=False)
    s["scale_raw"][tiny] = rng.normal(-11.5, 1.5, (200, 3)).astype(np.float32)
    flat = rng.choice(500, 100, replace=False)      # needles: one axis tiny, others normal
    s["scale_raw"][flat, 0] = rng.normal(-13.0, 1.0, 100).astype(np.float32)
    return s


def case_g8():  # f_rest = 0 (the "SH degree 0" plumbing case of config 1)
    rng = np.random.default_rng(108)
    s = _base(rng, 500, 64, 64, 70.0, 70.0, 32.0, 32.0)
    s["f_rest"][:] = 0
    return s


def case_g9a():  # every Gaussian fails the opacity prefilter -> zero image, zero grads
    rng = np.random.default_rng(109)
    s = _base(rng, 64, 32, 32, 30.0, 30.0, 16.0, 16.0)
    s["opacity_raw"][:] = -9.0
    return s


def case_g9b():  # every Gaussian behind the camera -> zero image, zero grads
    rng = np.random.default_rng(110)
    s = _base(rng, 64, 32, 32, 30.0, 30.0, 16.0, 16.0, c2w=np.eye(4, dtype=np.float32))
    s["pos"][:, 2] = -np.abs(s["pos"][:, 2])
    return s


def case_g10():  # survivors, none on screen -> Exception("All projected points are off-screen")
    rng = np.random.default_rng(111)
    N, H, W, f = 48, 32, 32, 30.0
    c2w = np.eye(4, dtype=np.float32)
    z = rng.uniform(3.0, 5.0, N)
    u = rng.uniform(-25.0, -18.0, N)           # inside the 32 px guard band, left of the image
    v = rng.uniform(4.0, 28.0, N)
    pc = np.stack([(u - 16.0) / f * z, (v - 16.0) / f * z, z], 1)
    s = _base(rng, N, H, W, f, f, 16.0, 16.0, c2w=c2w)
    s["pos"] = pc.astype(np.float32)
    s["scale_raw"][:] = -4.0                     # radius ~ 1 px -> umax < 0
    return s


def case_g12():  # non-default kwargs
    rng = np.random.default_rng(112)
    s = _base(rng, 500, 64, 96, 85.0, 85.0, 48.0, 32.0)
    big = rng.choice(500, 50, replace=False)     # lambda > 1e3 -> conic diagonal < min_conis = 1e-3
    s["scale_raw"][big] = rng.normal(0.75, 0.2, (50, 3)).astype(np.float32)
    s["opacity_raw"][big] = rng.normal(-2.5, 0.5, 50).astype(np.float32)
    s["kwargs"] = dict(near=2.8, far=6.0, pix_guard=8, min_conis=1e-3, chi_square_clip=4.0, alpha_max=0.9,
                       alpha_cutoff=1 / 64.)
    return s


def _case_g14(T):  # another tile size: same scene for every T (the image must not depend on it; the pair count does)
    rng = np.random.default_rng(114)
    s = _base(rng, 500, 72, 104, 80.0, 84.0, 50.0, 37.0)
    s["kwargs"] = dict(T=T)
    return s


def case_g14_T8():
    return _case_g14(8)


def case_g14_T32():
    return _case_g14(32)


CASES = {
    "g1_generic": case_g1, "g2_ragged": case_g2, "g3_occlusion": case_g3, "g4_thresholds": case_g4,
    "g5_guardband": case_g5, "g6_huge": case_g6, "g7_tiny": case_g7, "g8_deg0": case_g8,
    "g9a_empty_opacity": case_g9a, "g9b_empty_behind": case_g9b, "g10_offscreen": case_g10,
    "g12_kwargs": case_g12, "g14_T8": case_g14_T8, "g14_T32": case_g14_T32,
}
