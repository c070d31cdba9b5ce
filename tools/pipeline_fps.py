#!/usr/bin/env python3
"""Forward-only frames per second on the benchmark scene, frame by frame (render_gaussians) against software-pipelined over
two streams (render_frames) (diagnostic; needs a GPU):  python tools/pipeline_fps.py [config] [frames]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench
import gsplat_amd as gs

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 64
params, cam = bench.synthetic_scene(cfg)
dev = torch.device("cuda:0")
p = [params[k].to(dev) for k in bench.NAMES]
cams = [bench.orbit_c2w(k % 8).to(dev) for k in range(frames)]
cargs = (cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
with torch.no_grad():
    ref = [gs.render_gaussians(*p, c, *cargs) for c in cams[:8]]
    got = gs.render_frames(*p, cams[:8], *cargs)
    torch.cuda.synchronize()
    print("max |pipelined - sequential| over 8 frames:", max(float((a - b).abs().max()) for a, b in zip(ref, got)))
    for name, fn in (("sequential", lambda: [gs.render_gaussians(*p, c, *cargs) for c in cams]),
                     ("pipelined, waiting per frame (on_frame)", lambda: gs.render_frames(*p, cams, *cargs, on_frame=lambda k, im: None)),
                     ("pipelined, no waiting (deferred checks)", lambda: gs.render_frames(*p, cams, *cargs))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name}: {dt / frames * 1e3:.3f} ms per frame = {frames / dt:.0f} fps")
