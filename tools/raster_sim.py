#!/usr/bin/env python3
"""Design arithmetic for the raster kernels (CPU, numpy, no GPU): replays the chunk / sub-tile-queue traversal of
raster_forward_kernel / raster_backward_kernel on a sample of the lists of a synthetic scene and counts loop
iterations (= longest sub-tile queue of every chunk, summed) for variants of the queue construction:

    base        exact ellipse / sub-tile test, chunks of 64, queue cap 24 (what the backward runs today)
    dead        + entries dropped from the queues of sub-tiles whose 16 pixels are all dead (T <= 5e-5) when the chunk is staged
    opac        + ellipse shrunk to {q <= min(chi, 2 ln(o / alpha_cutoff))}: beyond it alpha < alpha_cutoff, i.e. alpha = 0
    both        dead + opac
    4x2         16 queues of 4 x 2 sub-tiles (exact test), same chunking
    free        lower bound for queues decoupled across chunks: max over the queues of their TOTAL length

    python tools/raster_sim.py [config] [lists sampled]
"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tools")
from subtile_stats import project, scene          # noqa: E402

CHI, ACUT, AMAX, CHUNK, QCAP = 6.25, 1.0 / 128.0, 0.99, 64, 24


def qmin_rect(u, v, A, B, C, x0, x1, y0, y1):
    """minimum of q over the rectangle [x0, x1] x [y0, y1] (gs_math.h ellipse_touches_rect), vectorised over Gaussians"""
    dx0, dx1, dy0, dy1 = x0 - u, x1 - u, y0 - v, y1 - v
    X = np.clip(0.0, dx0, dx1)
    t = np.clip(-B / C * X, dy0, dy1)
    qx = A * X * X + (2 * B * X + C * t) * t
    Y = np.clip(0.0, dy0, dy1)
    s = np.clip(-B / A * Y, dx0, dx1)
    qy = C * Y * Y + (2 * B * Y + A * s) * s
    return np.minimum(qx, qy)


def simulate(config=3, n_lists=400, seed=0):
    pos, scale, q, op, H, W, fx = scene(config)
    u, v, ex, ey, A11, A12, A22, o = project(pos, scale, q, op, H, W, fx)
    # depth of the survivors (same filter as project())
    X, Y, Z = pos.T
    cx, cy = W / 2, H / 2
    oo = np.clip(1 / (1 + np.exp(-op)), 0, 0.999)
    keep = (oo >= 1 / 256) & (Z > 0.01) & (Z < 100) & (fx * X > Z * (-32 - cx)) & (fx * X < Z * (W + 32 - cx)) & \
           (fx * Y > Z * (-32 - cy)) & (fx * Y < Z * (H + 32 - cy))
    z = Z[keep]
    lists_x, lists_y = (W + 15) // 16, (H + 7) // 8
    rng = np.random.default_rng(seed)
    sample = rng.choice(lists_x * lists_y, size=min(n_lists, lists_x * lists_y), replace=False)
    x0b, x1b, y0b, y1b = u - ex, u + ex, v - ey, v + ey
    qlim_opac = np.minimum(CHI, 2.0 * np.log(np.maximum(o / ACUT, 1e-30)))
    tot = {k: 0 for k in ("base", "dead", "opac", "both", "4x2", "4x2both", "free", "freeboth")}
    pairs = {k: 0 for k in tot}
    entries = chunks = 0
    chunks_base = 0
    inside_base = 0
    chunks_pool = {}
    px = np.arange(16)[None, :].repeat(8, 0).reshape(-1).astype(np.float64)
    py = np.arange(8)[:, None].repeat(16, 1).reshape(-1).astype(np.float64)
    sub44 = (py.astype(int) // 4) * 4 + px.astype(int) // 4            # sub-tile of every pixel, 4 x 4
    sub42 = (py.astype(int) // 2) * 4 + px.astype(int) // 4            # 4 x 2
    for l in sample:
        lx, ly = l % lists_x, l // lists_x
        ox, oy = lx * 16.0, ly * 8.0
        m = (x1b >= ox) & (x0b <= ox + 15) & (y1b >= oy) & (y0b <= oy + 7)
        idx = np.nonzero(m)[0]
        if len(idx) == 0:
            continue
        # exact list test (what is binned)
        qm = qmin_rect(u[idx], v[idx], A11[idx], A12[idx], A22[idx], ox, ox + 15, oy, oy + 7)
        idx = idx[qm <= CHI * 1.001 + 1e-4]
        idx = idx[np.argsort(z[idx], kind="stable")]
        n = len(idx)
        if n == 0:
            continue
        uu, vv, A, B, C, op_ = u[idx], v[idx], A11[idx], A12[idx], A22[idx], o[idx]
        # exact sub-tile masks, both shapes, with and without the opacity bound
        def masks(sw, sh, lim):
            gx, gy = 16 // sw, 8 // sh
            cols = []
            for r in range(gy):
                for c in range(gx):
                    qq = qmin_rect(uu, vv, A, B, C, ox + sw * c, ox + sw * c + sw - 1, oy + sh * r, oy + sh * r + sh - 1)
                    cols.append(qq <= lim * 1.001 + 1e-4)
            return np.stack(cols, 1)                                  # [n, queues]
        m44, m44o = masks(4, 4, np.full(n, CHI)), masks(4, 4, qlim_opac[idx])
        m42, m42o = masks(4, 2, np.full(n, CHI)), masks(4, 2, qlim_opac[idx])
        # per-pixel alpha of every entry (for T): [n, 128]
        du, dv = (ox + px)[None, :] - uu[:, None], (oy + py)[None, :] - vv[:, None]
        qq = A[:, None] * du * du + 2 * B[:, None] * du * dv + C[:, None] * dv * dv
        al = np.minimum(op_[:, None] * np.exp(-0.5 * np.minimum(qq, CHI)), AMAX) * (qq <= CHI)
        al = np.where(al >= ACUT, al, 0.0)

        def run(mask, subof, nq, drop_dead, cap=QCAP, chunked=True, pool=None):
            T = np.ones(128)
            it = 0
            pr = 0
            base = 0
            qtot = np.zeros(nq)
            while base < n:
                alive_px = T > 5e-5
                if not alive_px.any():
                    break
                live_q = np.array([alive_px[subof == s].any() for s in range(nq)])
                take = min(CHUNK, n - base)
                mm = mask[base:base + take].copy()
                if drop_dead:
                    mm &= live_q[None, :]
                # queue cap: longest prefix of the chunk whose queues all fit
                cs = mm.cumsum(0)
                over = np.nonzero((cs > cap).any(1))[0] if pool is None else np.nonzero(cs.sum(1) > pool)[0]
                if len(over):
                    take = max(int(over[0]), 1)
                    mm = mm[:take]
                lens = mm.sum(0)
                inside[0] += int((qq[base:base + take] <= CHI).sum())           # lane evaluations that land inside the ellipse
                it += int(lens.max())
                pr += int(lens.sum())
                qtot += lens
                # transmittance after the chunk (entries of dropped queues touch only dead pixels: no change)
                a = al[base:base + take]
                for k in range(take):
                    ak = np.where(T > 5e-5, a[k], 0.0)
                    T = T - ak * T
                base += take
                nch[0] += 1
            return it, pr, int(qtot.max())
        nch = [0]
        inside = [0]
        i0, p0, f0 = run(m44, sub44, 8, False)
        chunks_base += nch[0]
        inside_base += inside[0]
        for pl in (128, 160, 192, 10 ** 9):
            nch = [0]
            ip, pp, _ = run(m44, sub44, 8, False, pool=pl)
            key = f"pool{pl if pl < 10 ** 9 else 'inf'}"
            tot[key] = tot.get(key, 0) + ip
            pairs[key] = pairs.get(key, 0) + pp
            chunks_pool[key] = chunks_pool.get(key, 0) + nch[0]
        nch = [0]
        i1, p1, _ = run(m44, sub44, 8, True)
        i2, p2, _ = run(m44o, sub44, 8, False)
        i3, p3, f3 = run(m44o, sub44, 8, True)
        i4, p4, _ = run(m42, sub42, 16, False, cap=QCAP)
        i5, p5, _ = run(m42o, sub42, 16, True, cap=QCAP)
        for k, (i, p) in zip(("base", "dead", "opac", "both", "4x2", "4x2both"), ((i0, p0), (i1, p1), (i2, p2), (i3, p3), (i4, p4), (i5, p5))):
            tot[k] += i
            pairs[k] += p
        tot["free"] += f0
        pairs["free"] += p0
        tot["freeboth"] += f3
        pairs["freeboth"] += p3
        entries += n
    nl = len(sample)
    scale_up = lists_x * lists_y / nl
    print(f"config {config}: {nl} lists sampled of {lists_x * lists_y}; entries {entries} (-> P_b ~ {entries * scale_up / 1e6:.2f} M)")
    b = tot["base"]
    print(f"  chunks: base (queue cap {QCAP}) {chunks_base * scale_up / 1e3:.1f} K; " + "; ".join(f"{k} {v * scale_up / 1e3:.1f} K" for k, v in chunks_pool.items()))
    useful = inside_base / max(tot["base"] * 128, 1)
    print(f"  base: (pixel, Gaussian) evaluations inside the ellipse {inside_base * scale_up / 1e6:.1f} M of {tot['base'] * 128 * scale_up / 1e6:.1f} M lane "
          f"evaluations = useful-lane fraction {useful:.3f}")
    if len(sys.argv) > 3:
        import json
        json.dump({"config": config, "lists_sampled": nl, "iterations": tot["base"] * scale_up, "chunks": chunks_base * scale_up,
                   "subtile_pairs": pairs["base"] * scale_up, "inside_evaluations": inside_base * scale_up, "useful_lane_fraction": useful,
                   "queue_imbalance": tot["base"] * 8 / max(pairs["base"], 1),
                   "variants_iterations_vs_base": {k: tot[k] / b for k in tot}}, open(sys.argv[3], "w"), indent=1)
    for k in tot:
        nq = 16 if k.startswith("4x2") else 8
        print(f"  {k:9s} iterations {tot[k]:9d} ({tot[k] * scale_up / 1e6:6.3f} M scaled)  = {tot[k] / b:5.3f} x base   "
              f"(sub-tile, Gaussian) pairs {pairs[k] * scale_up / 1e6:6.2f} M  balance {tot[k] * nq / max(pairs[k], 1):4.2f}")


if __name__ == "__main__":
    simulate(int(sys.argv[1]) if len(sys.argv) > 1 else 3, int(sys.argv[2]) if len(sys.argv) > 2 else 400)
