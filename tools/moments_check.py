#!/usr/bin/env python3
"""For one Gaussian of one seed of test_random_scenes_vs_oracle: the nine per-Gaussian sums the raster backward leaves in grad2d
(moments of dL/dalpha * g about the centre, d opacity, d colour) against the same sums formed from the float64 oracle's cotangents
(diagnostic; needs a GPU).      python tools/moments_check.py <seed> <gaussian>"""
import ctypes as C
import importlib
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from oracle import scenes
from oracle import torch_port as tp
from tests import util

seed, gi = int(sys.argv[1]), int(sys.argv[2])
PKG = "3d-gaussian-splatting-for-novel-view-synthesis_amd"
gs = importlib.import_module(PKG)
ops = importlib.import_module(PKG + ".ops")
abi = importlib.import_module(PKG + "._abi")
rng = np.random.default_rng(1000 + seed)
H, W = int(rng.integers(9, 150)), int(rng.integers(9, 200))
n = int(rng.integers(1, 1800))
f = float(rng.uniform(40, 160))
cam = (H, W, f, f * float(rng.uniform(0.9, 1.1)), W / 2 + float(rng.uniform(-5, 5)), H / 2 + float(rng.uniform(-5, 5)))
c2w = torch.tensor(scenes._camera(rng, tilt=0.3))
s = scenes._base(rng, n, H, W, cam[2], cam[3], cam[4], cam[5], mu_s=float(rng.uniform(-3.2, -1.2)), sd_s=float(rng.uniform(0.2, 1.0)),
                 op_mu=float(rng.uniform(-2, 3)), op_sd=1.5, spread=1.3, c2w=c2w.numpy())
t = {k: torch.tensor(s[k]) for k in util.PARAMS}
zc = tp.to_camera(t["pos"].double(), c2w.double())[2]
zs, order = torch.sort(zc)
keep = torch.ones(n, dtype=torch.bool)
keep[order[1:][(zs[1:] - zs[:-1]) < 2e-5]] = False
t = {k: v[keep].contiguous() for k, v in t.items()}
w = torch.tensor(rng.uniform(0, 1, (H, W, 3)).astype(np.float32))
names = ("pos", "f_dc", "f_rest", "opacity_raw", "scale_raw", "q_raw")
# ---- float64 oracle: cotangents of (u, v, cov2d, opacity, colour) of Gaussian gi
p = {k: v.double().requires_grad_(True) for k, v in t.items()}
st = {}
img = tp.render_fused(*[p[k] for k in names], c2w.double(), *cam, stages=st)
loss = (img * w.double()).sum()
g_u, g_v, g_cov, g_op, g_col = torch.autograd.grad(loss, [st["u"], st["v"], st["cov2d"], st["opacity"], st["color"]])
j = int(np.nonzero(st["ids"].numpy() == gi)[0][0])
M = st["cov2d"][j].detach()
K = torch.linalg.inv(M)
Gs = 0.5 * (g_cov[j] + g_cov[j].t())
GK = -(M @ Gs @ M)                                       # d L / d conic (symmetric 2 x 2), exact chain of the inverse
o = float(st["opacity"][j])
ref = dict(g_u=float(g_u[j]), g_v=float(g_v[j]), g_A11=float(GK[0, 0]), g_A12=float(2 * GK[0, 1]), g_A22=float(GK[1, 1]), g_op=float(g_op[j]))
# ---- HIP: capture grad2d of the separate-calls path
hip = C.CDLL("libamdhip64.so")
captured = {}
lib = abi.lib()
real = lib.gsplat_project_backward


def spy(g, c2w_, view, ps, grad2d, gg, flags, stream):
    torch.cuda.synchronize()
    nn = len(t["pos"])
    buf = np.zeros((nn, 16), np.float32)
    hip.hipMemcpy(buf.ctypes.data_as(C.c_void_p), grad2d, C.c_size_t(buf.nbytes), 2)
    captured["grad2d"] = buf
    return real(g, c2w_, view, ps, grad2d, gg, flags, stream)


ops._composite = False
lib.gsplat_project_backward = spy
q = {k: v.to("cuda:0").requires_grad_(True) for k, v in t.items()}
im = gs.render_gaussians(*[q[k] for k in names], c2w.to("cuda:0"), *cam)
(im * w.to("cuda:0")).sum().backward()
torch.cuda.synchronize()
lib.gsplat_project_backward = real
r = captured["grad2d"][gi].astype(np.float64)
Mx, My, Mxx, Mxy, Myy, M0 = r[:6]
A11, A12, A22 = float(K[0, 0]), float(K[0, 1]), float(K[1, 1])
hipv = dict(g_u=o * (A11 * Mx + A12 * My), g_v=o * (A12 * Mx + A22 * My), g_A11=-0.5 * o * Mxx, g_A12=-o * Mxy, g_A22=-0.5 * o * Myy, g_op=M0)
print(f"seed {seed} Gaussian {gi}: opacity {o:.6f} conic {A11:.6g} {A12:.6g} {A22:.6g} eigenvalues of cov2d {st['evals'][j].detach().numpy()}")
for k in ref:
    print(f"  {k:6s} float64 oracle {ref[k]: .9e}   HIP moments {hipv[k]: .9e}   rel diff {abs(hipv[k] - ref[k]) / (abs(ref[k]) + 1e-300):.2e}")
# what the difference does after the (exact, float64) chain to the covariance: -K G K in the eigenbasis of cov2d
ev, V = np.linalg.eigh(M.numpy())
for name, d in (("oracle", ref), ("HIP", hipv)):
    G = np.array([[d["g_A11"], d["g_A12"] / 2], [d["g_A12"] / 2, d["g_A22"]]])
    Gc = -(K.numpy() @ G @ K.numpy())
    Ge = V.T @ Gc @ V
    print(f"  d L / d cov2d in its eigenbasis ({name}): thin-thin {Ge[0, 0]: .6e}  thin-long {Ge[0, 1]: .6e}  long-long {Ge[1, 1]: .6e}")
