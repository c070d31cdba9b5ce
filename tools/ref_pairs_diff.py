#!/usr/bin/env python3
"""Which Gaussians make the kernels' count of the reference's (tile, Gaussian) pairs P differ from the reference's own fp32 count?
(diagnostic; needs a GPU and the DIAGNOSTICS build).  Per Gaussian: the tile rectangle the projection kernel derives (F10: u +-
ceil(2.5 sqrt(lambda_max)), floor, clamp, // T) against the one the oracle derives in float32 -- the reference's own arithmetic:
three chained bmm for Sigma, torch.linalg.eigh (LAPACK) for lambda -- and in float64.

    python tools/ref_pairs_diff.py [config]
"""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "3d-gaussian-splatting-for-novel-view-synthesis_amd"
os.environ.setdefault("GSPLAT_MI355X_LIB", os.path.join(ROOT, PKG, "csrc", "libgsplat_mi355x_diag.so"))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import bench
from oracle import torch_port as tp

gs = importlib.import_module(PKG)
abi = importlib.import_module(PKG + "._abi")
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
params, cam = bench.synthetic_scene(cfg)
N = params["pos"].shape[0]
H, W = cam["H"], cam["W"]
dev = torch.device("cuda:0")
lib = abi.lib()
lib.gsplat_debug_set_ref_rect.argtypes = [C.c_void_p, C.c_void_p]
rect = torch.zeros(N, 2, dtype=torch.int32, device=dev)
tiles = torch.zeros(N, dtype=torch.int32, device=dev)
lib.gsplat_debug_set_ref_rect(C.c_void_p(rect.data_ptr()), C.c_void_p(tiles.data_ptr()))
with torch.no_grad():
    gs.render_gaussians(*[params[k].to(dev) for k in bench.NAMES], torch.eye(4, device=dev), H, W, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
torch.cuda.synchronize()
lib.gsplat_debug_set_ref_rect(None, None)
V_dev, P_dev = gs.render_stats()[1:]
dev_tiles = tiles.cpu().numpy().astype(np.int64)
r = rect.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
dev_rect = np.stack([r[:, 0] & 0xFFFF, r[:, 0] >> 16, r[:, 1] & 0xFFFF, r[:, 1] >> 16], 1)          # tx0, ty0, tx1, ty1


def oracle(dtype):
    st = {}
    q = {k: params[k].to(dtype) for k in bench.NAMES}
    torch.set_num_threads(bench.host_cores())
    tp.render_fused(*[q[k] for k in bench.NAMES], torch.eye(4, dtype=dtype), H, W, cam["fx"], cam["fy"], cam["cx"], cam["cy"], stages=st,
                    stop_after_binning=True)
    t = np.zeros(N, np.int64)
    rc = np.zeros((N, 4), np.int64)
    ids = st["ids"].numpy()
    tr = st["tile_rect"].numpy()
    t[ids] = (tr[:, 2] - tr[:, 0] + 1) * (tr[:, 3] - tr[:, 1] + 1)
    rc[ids] = tr
    lam = np.zeros(N)
    lam[ids] = st["evals"][:, 1].double().numpy()
    uv = np.zeros((N, 2))
    uv[ids, 0], uv[ids, 1] = st["u"].double().numpy(), st["v"].double().numpy()
    return t, rc, lam, uv


t32, r32, lam32, uv32 = oracle(torch.float32)
t64, r64, lam64, uv64 = oracle(torch.float64)
print(f"config {cfg}: V device {V_dev}; P device {P_dev}, oracle float32 {t32.sum()} (V {int((t32 > 0).sum())}), oracle float64 {t64.sum()} (V {int((t64 > 0).sum())})")
for name, t, rc in (("float32 oracle", t32, r32), ("float64 oracle", t64, r64)):
    d = np.nonzero(dev_tiles != t)[0]
    print(f"device vs {name}: {len(d)} Gaussians differ, sum of differences {int((dev_tiles - t)[d].sum())}, sum |.| {int(np.abs(dev_tiles - t)[d].sum())}")
d = np.nonzero(dev_tiles != t32)[0]
ulp = lambda x: np.spacing(np.float32(x)).astype(np.float64)
for i in d[:40]:
    x32, x64 = 2.5 * np.sqrt(lam32[i]), 2.5 * np.sqrt(lam64[i])
    print(f"  #{i}: device rect {dev_rect[i].tolist()} tiles {dev_tiles[i]} | f32 oracle rect {r32[i].tolist()} tiles {t32[i]} | f64 rect {r64[i].tolist()} tiles {t64[i]}\n"
          f"        2.5 sqrt(lambda): f32 oracle {x32:.9f} f64 {x64:.9f} (distance to an integer {abs(x64 - round(x64)):.2e} = {abs(x64 - round(x64)) / ulp(x64):.1f} ulp)   "
          f"u {uv64[i, 0]:.6f} v {uv64[i, 1]:.6f} (frac {uv64[i, 0] % 1:.2e} {uv64[i, 1] % 1:.2e})")
# how many Gaussians sit that close to a flip at all?  (what ANY fp32 evaluation may get differently)
x = 2.5 * np.sqrt(np.clip(lam64[t64 > 0], 1e-12, 1e4))
dist = np.abs(x - np.round(x)) / ulp(x)
fu = np.minimum(uv64[t64 > 0] % 1, 1 - uv64[t64 > 0] % 1) / ulp(np.abs(uv64[t64 > 0]) + 1)
print(f"visible Gaussians with 2.5 sqrt(lambda) within 4 / 16 / 64 fp32 ulp of an integer: {(dist < 4).sum()} / {(dist < 16).sum()} / {(dist < 64).sum()};"
      f" with u or v within 4 ulp of an integer: {(fu.min(1) < 4).sum()}")
