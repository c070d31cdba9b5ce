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
sf = torch.zeros(nreg, 6, dtype=torch.int32, device=dev)
sb = torch.zeros(nreg, 6, dtype=torch.int32, device=dev)
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
    ln, ch, vis, cyc = a[:, 0], a[:, 1] & 0xFFF, a[:, 2], a[:, 3]
    real = a[:, 1] >> 12                                            # wave duration, 10 ns ticks
    act = ln > 0
    # "visited" = loop iterations (longest sub-tile queue of every chunk, summed): 8 (sub-tile, Gaussian) pairs each when balanced
    print(f"{name}: lists {act.sum()}  list_len mean {ln[act].mean():.0f} max {ln.max()}  chunks mean {ch[act].mean():.1f} max {ch.max()} "
          f" iterations mean {vis[act].mean():.0f} max {vis.max()} total {vis.sum() / 1e6:.2f}M  (entries / iteration {ln.sum() / max(vis.sum(), 1):.2f})")
    print(f"   cycles: mean {cyc[act].mean():.0f} p50 {np.percentile(cyc[act], 50):.0f} p90 {np.percentile(cyc[act], 90):.0f} "
          f"p99 {np.percentile(cyc[act], 99):.0f} max {cyc.max()}  (100 MHz ticks? see below)  sum {cyc.sum() / 1e6:.1f}M")
    i = np.argsort(cyc)[-5:][::-1]
    print("   slowest:", [(int(ln[k]), int(ch[k]), int(vis[k]), int(cyc[k])) for k in i])
    # timeline: waves resident over the kernel's duration (begin = low 32 bits of s_memtime at wave start)
    beg = a[:, 4][act]
    # begin = s_memrealtime (100 MHz, chip-wide); a wave's duration in the same unit from the kernel's own clock ratio
    d = ((beg - beg[0] + (1 << 31)) & 0xFFFFFFFF) - (1 << 31)
    b0 = (d - d.min()).astype(np.float64) * 10e-3                  # us
    xcc = (a[:, 5][act] >> 24) & 0xF
    dur = real[act] * 10e-3                                        # us
    clk = cyc[act].sum() / max(real[act].sum(), 1) / 10.0          # s_memtime ticks per ns
    e0 = b0 + dur
    total = e0.max()
    edges = np.linspace(0, total, 21)
    resident = [int(((b0 < edges[k + 1]) & (e0 > edges[k])).sum()) for k in range(20)]
    print(f"   kernel span {total:.1f} us (s_memtime runs at {clk:.2f} GHz); waves resident per 5 % slice of the span: {resident}; XCDs seen {len(np.unique(xcc))}")
    print(f"   last wave to START begins at {b0.max() / total:.2f} of the span; slot-time used {dur.sum() / (total * max(resident)):.2f} of span x peak residency")
    late = np.argsort(e0)[-5:][::-1]
    print("   last to finish (len, iterations, start/span, cycles, launch index):",
          [(int(ln[act][k]), int(vis[act][k]), round(float(b0[k] / total), 2), round(float(dur[k]), 1), int(a[:, 5][act][k] & 0xFFFFFF)) for k in late])
    frac = ch[act] * 64 / np.maximum(ln[act], 1)
    print(f"   fraction of list staged before termination: mean {np.minimum(frac, 1).mean():.2f}; cycles per visited: {cyc[act].sum() / max(vis.sum(), 1):.1f}")
