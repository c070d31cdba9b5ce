#!/usr/bin/env python3
"""Soak test of the counter protocol of the projection kernel (last-wave totals, self-cleaning counter block, mapped host store):
thousands of frames over a few camera poses without waiting; every frame's counters must equal the ones the same pose gave when
rendered alone, and the images must stay bit-identical.    python tools/soak.py [config] [frames]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench
import gsplat_amd as gs

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
params, cam = bench.synthetic_scene(cfg)
dev = torch.device("cuda:0")
p = [params[k].to(dev) for k in bench.NAMES]
cams = [bench.orbit_c2w(k).to(dev) for k in (0, 1, 7)]
cargs = (cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
ref_counts, ref_img = [], []
with torch.no_grad():
    for c in cams:
        ref_img.append(gs.render_gaussians(*p, c, *cargs))
        ref_counts.append(gs.render_stats())
    torch.cuda.synchronize()
    bad = 0
    t0 = time.perf_counter()
    for start in range(0, frames, 120):
        with gs.deferred_checks() as chk:
            imgs = [gs.render_gaussians(*p, cams[k % 3], *cargs) for k in range(start, min(start + 120, frames))]
        counts = chk.verify()[-len(imgs):]
        for k, (c, im) in enumerate(zip(counts, imgs)):
            want = ref_counts[(start + k) % 3]
            if (c.n_survivors, c.n_visible, int(c.n_pairs)) != tuple(want):
                bad += 1
                print("frame", start + k, "counters", (c.n_survivors, c.n_visible, int(c.n_pairs)), "expected", want)
        if not all(torch.equal(im, ref_img[(start + k) % 3]) for k, im in enumerate(imgs[:6])):
            bad += 1
            print("block at", start, ": image differs")
    torch.cuda.synchronize()
    print(f"{frames} frames in {time.perf_counter() - t0:.1f} s, mismatches: {bad}")
sys.exit(1 if bad else 0)
