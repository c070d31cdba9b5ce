#!/usr/bin/env python3
"""Per-wave statistics of the raster kernels on the benchmark scene (diagnostic; needs a GPU).  Uses the DIAGNOSTICS build of the
library (the product library has no statistics hook):
    make -C 3d-gaussian-splatting-for-novel-view-synthesis_amd/csrc libgsplat_mi355x_diag.so
    python tools/raster_stats.py [config]"""
import ctypes as C
import importlib
import os
import sys

os.environ.setdefault("GSPLAT_MI355X_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                        "3d-gaussian-splatting-for-novel-view-synthesis_amd", "csrc", "libgsplat_mi355x_diag.so"))

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench

PKG = "3d-gaussian-splatting-for-novel-view-synthesis_amd"
gs = importlib.import_module(PKG)
abi = importlib.import_module(PKG + "._abi")
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
params, cam = bench.synthetic_scene(cfg)
dev = torch.device("cuda:0")
p = {k: v.to(dev).requires_grad_(True) for k, v in params.items()}
H, W = cam["H"], cam["W"]
nreg = ((H + 7) // 8) * ((W + 15) // 16)            # one record per half-tile list
sf = torch.zeros(nreg, 4, dtype=torch.int32, device=dev)
sb = torch.zeros(nreg, 4, dtype=torch.int32, device=dev)
lib = abi.lib()
lib._FuncPtr  # noqa
lib.gsplat_debug_set_stats.argtypes = [C.c_void_p, C.c_void_p]
args = [p[k] for k in bench.NAMES] + [torch.eye(4, device=dev), H, W, cam["fx"], cam["fy"], cam["cx"], cam["cy"]]
gimg = torch.rand(H, W, 3, device=dev)
for it in range(2):
    if it == 1:
        lib.gsplat_debug_set_stats(C.c_void_p(sf.data_ptr()), C.c_void_p(sb.data_ptr()))
    img = gs.render_gaussians(*args)
    img.backward(gimg)
    torch.cuda.synchronize()
lib.gsplat_debug_set_stats(None, None)
for name, t in (("forward", sf), ("backward", sb)):
    a = t.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    ln, ch, vis, cyc = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    act = ln > 0
    # "visited" = loop iterations (longest sub-tile queue of every chunk, summed): 8 (sub-tile, Gaussian) pairs each when balanced
    print(f"{name}: lists {act.sum()}  list_len mean {ln[act].mean():.0f} max {ln.max()}  chunks mean {ch[act].mean():.1f} max {ch.max()} "
          f" iterations mean {vis[act].mean():.0f} max {vis.max()} total {vis.sum() / 1e6:.2f}M  (entries / iteration {ln.sum() / max(vis.sum(), 1):.2f})")
    print(f"   cycles: mean {cyc[act].mean():.0f} p50 {np.percentile(cyc[act], 50):.0f} p90 {np.percentile(cyc[act], 90):.0f} "
          f"p99 {np.percentile(cyc[act], 99):.0f} max {cyc.max()}  (100 MHz ticks? see below)  sum {cyc.sum() / 1e6:.1f}M")
    i = np.argsort(cyc)[-5:][::-1]
    print("   slowest:", [(int(ln[k]), int(ch[k]), int(vis[k]), int(cyc[k])) for k in i])
    frac = ch[act] * 64 / np.maximum(ln[act], 1)
    print(f"   fraction of list staged before termination: mean {np.minimum(frac, 1).mean():.2f}; cycles per visited: {cyc[act].sum() / max(vis.sum(), 1):.1f}")
