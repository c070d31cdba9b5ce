#!/usr/bin/env python3
"""Design arithmetic for the rasterizer (CPU, numpy): how many (region, Gaussian) evaluations a scene needs for different
region shapes, and how uneven the sub-tile queues of one 16x8 list are.  No GPU, no oracle: plain projection math on the
synthetic scenes of SURVEY.md 8d (own restatement, float64).

    python tools/subtile_stats.py [config]
"""
import sys

import numpy as np


def scene(config):
    import torch
    cfg = {1: (10_000, 256, 256, 300.0, -3.0), 2: (100_000, 800, 800, 800.0, -4.5), 3: (1_000_000, 1080, 1920, 1100.0, -5.0),
           4: (3_000_000, 1080, 1920, 1100.0, -5.4), 5: (10_000_000, 2160, 3840, 2200.0, -5.8)}[config]
    N, H, W, fx, mu = cfg
    g = torch.Generator().manual_seed(0)
    pos = torch.randn(N, 3, generator=g)
    pos[:, 2] += 5.0
    scale = torch.randn(N, 3, generator=g) * 0.3 + mu
    q = torch.randn(N, 4, generator=g)
    op = torch.randn(N, generator=g)
    return pos.double().numpy(), scale.double().numpy(), q.double().numpy(), op.double().numpy(), H, W, fx


def project(pos, scale, q, op, H, W, fx):
    s = np.maximum(np.exp(scale), 1e-6)
    qn = q / (np.linalg.norm(q, axis=1, keepdims=True) + 1e-9)
    x, y, z, w = qn.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                  2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    RS = R * s[:, None, :]
    S = RS @ RS.transpose(0, 2, 1)
    o = np.clip(1 / (1 + np.exp(-op)), 0, 0.999)
    X, Y, Z = pos.T
    cx, cy = W / 2, H / 2
    keep = (o >= 1 / 256) & (Z > 0.01) & (Z < 100) & (fx * X > Z * (-32 - cx)) & (fx * X < Z * (W + 32 - cx)) & \
           (fx * Y > Z * (-32 - cy)) & (fx * Y < Z * (H + 32 - cy))
    X, Y, Z, S, o = X[keep], Y[keep], Z[keep], S[keep], o[keep]
    u, v = fx * X / Z + cx, fx * Y / Z + cy
    J = np.zeros((len(X), 2, 3))
    J[:, 0, 0] = fx / Z; J[:, 0, 2] = -fx * X / Z ** 2; J[:, 1, 1] = fx / Z; J[:, 1, 2] = -fx * Y / Z ** 2
    C = J @ S @ J.transpose(0, 2, 1)
    a, b, d = C[:, 0, 0], 0.5 * (C[:, 0, 1] + C[:, 1, 0]), C[:, 1, 1]
    det = np.maximum(a * d - b * b, 1e-12)
    A11, A12, A22 = np.maximum(d / det, 1e-6), -b / det, np.maximum(a / det, 1e-6)
    D = A11 * A22 - A12 ** 2
    ex, ey = np.sqrt(6.25 * A22 / D) + 0.01, np.sqrt(6.25 * A11 / D) + 0.01
    return u, v, ex, ey, A11, A12, A22, o


def count_regions(u, v, ex, ey, W, H, rw, rh):
    """bounding-box count of rw x rh pixel regions touched (pixel centres, clipped to the image), and evaluations."""
    x0 = np.clip(np.ceil(u - ex), 0, W - 1); x1 = np.clip(np.floor(u + ex), 0, W - 1)
    y0 = np.clip(np.ceil(v - ey), 0, H - 1); y1 = np.clip(np.floor(v + ey), 0, H - 1)
    ok = (np.ceil(u - ex) <= W - 1) & (np.floor(u + ex) >= 0) & (np.ceil(v - ey) <= H - 1) & (np.floor(v + ey) >= 0) & \
         (np.ceil(u - ex) <= np.floor(u + ex)) & (np.ceil(v - ey) <= np.floor(v + ey))
    nx = (x1 // rw - x0 // rw + 1) * ok
    ny = (y1 // rh - y0 // rh + 1) * ok
    return float((nx * ny).sum())


def main():
    config = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    pos, scale, q, op, H, W, fx = scene(config)
    u, v, ex, ey, A11, A12, A22, o = project(pos, scale, q, op, H, W, fx)
    V = len(u)
    area = np.pi * 6.25 / np.sqrt(A11 * A22 - A12 ** 2)
    print(f"config {config}: survivors {V}, mean ellipse area {area.mean():.1f} px, mean ex {ex.mean():.2f} ey {ey.mean():.2f}")
    base = None
    for rw, rh in ((16, 16), (16, 8), (8, 8), (8, 4), (4, 4), (4, 2), (2, 2)):
        n = count_regions(u, v, ex, ey, W, H, rw, rh)
        ev = n * rw * rh
        if (rw, rh) == (16, 8):
            base = ev
        print(f"  regions {rw:2d}x{rh:<2d}: pairs {n / 1e6:7.2f} M  ({n / V:5.2f} per Gaussian)   pixel evaluations {ev / 1e6:8.1f} M"
              + (f"  = {base / ev:4.2f}x fewer than 16x8" if base else ""))
    # imbalance of the 8 (4x4) or 16 (4x2) sub-tile queues inside 16x8 lists, chunks of 64 / 128 candidates (bounding-box test,
    # arrival order = random, which is what depth order is for this scene)
    rng = np.random.default_rng(0)
    x0 = np.ceil(u - ex); x1 = np.floor(u + ex); y0 = np.ceil(v - ey); y1 = np.floor(v + ey)
    lists_x, lists_y = (W + 15) // 16, (H + 7) // 8
    sample = rng.choice(lists_x * lists_y, size=300, replace=False)
    for sw, sh in ((4, 4), (4, 2)):
        gx, gy = 16 // sw, 8 // sh
        for chunk in (64, 128, 10 ** 9):
            tot_max = tot_mean = 0.0
            for l in sample:
                lx, ly = l % lists_x, l // lists_x
                px0, py0 = lx * 16, ly * 8
                m = (x1 >= px0) & (x0 <= px0 + 15) & (y1 >= py0) & (y0 <= py0 + 7)
                idx = np.nonzero(m)[0]
                rng.shuffle(idx)
                if len(idx) == 0:
                    continue
                cols = [(x1[idx] >= px0 + sw * c) & (x0[idx] <= px0 + sw * c + sw - 1) for c in range(gx)]
                rows = [(y1[idx] >= py0 + sh * r) & (y0[idx] <= py0 + sh * r + sh - 1) for r in range(gy)]
                hit = np.stack([rows[r] & cols[c] for r in range(gy) for c in range(gx)], 1)       # [n, groups]
                for s in range(0, len(idx), chunk):
                    c = hit[s:s + chunk].sum(0)
                    tot_max += c.max()
                    tot_mean += c.mean()
            print(f"  sub-tiles {sw}x{sh} ({gx * gy} queues), chunk {chunk if chunk < 10 ** 9 else 'whole list'}: "
                  f"iterations / balanced iterations = {tot_max / tot_mean:.3f}")


if __name__ == "__main__":
    main()


def exact_vs_bbox(config=3, sample=100_000, sw=4, sh=4):
    """How many of the bounding-box sub-tiles does the ellipse {q <= chi} really touch (exact convex minimisation per rectangle)?"""
    pos, scale, q, op, H, W, fx = scene(config)
    u, v, ex, ey, A11, A12, A22, o = project(pos, scale, q, op, H, W, fx)
    idx = np.random.default_rng(1).choice(len(u), size=min(sample, len(u)), replace=False)
    u, v, ex, ey, A11, A12, A22 = (a[idx] for a in (u, v, ex, ey, A11, A12, A22))
    x0 = np.clip(np.ceil(u - ex), 0, W - 1); x1 = np.clip(np.floor(u + ex), 0, W - 1)
    y0 = np.clip(np.ceil(v - ey), 0, H - 1); y1 = np.clip(np.floor(v + ey), 0, H - 1)
    ok = (np.ceil(u - ex) <= W - 1) & (np.floor(u + ex) >= 0) & (np.ceil(v - ey) <= H - 1) & (np.floor(v + ey) >= 0)
    bb = ex_ = 0
    for i in np.nonzero(ok)[0]:
        for ty in range(int(y0[i]) // sh, int(y1[i]) // sh + 1):
            for tx in range(int(x0[i]) // sw, int(x1[i]) // sw + 1):
                bb += 1
                rx0, rx1, ry0, ry1 = tx * sw - u[i], tx * sw + sw - 1 - u[i], ty * sh - v[i], ty * sh + sh - 1 - v[i]
                if rx0 <= 0 <= rx1 and ry0 <= 0 <= ry1:
                    ex_ += 1
                    continue
                best = 1e30
                for X in (rx0, rx1):
                    t = min(max(-A12[i] / A22[i] * X, ry0), ry1)
                    best = min(best, A11[i] * X * X + (2 * A12[i] * X + A22[i] * t) * t)
                for Y in (ry0, ry1):
                    t = min(max(-A12[i] / A11[i] * Y, rx0), rx1)
                    best = min(best, A22[i] * Y * Y + (2 * A12[i] * Y + A11[i] * t) * t)
                ex_ += best <= 6.25
    print(f"  sub-tiles {sw}x{sh}: bounding-box pairs {bb}, exact pairs {ex_}: exact / box = {ex_ / bb:.3f}")


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "exact":
    exact_vs_bbox(int(sys.argv[1]), 30_000, 4, 4)
    exact_vs_bbox(int(sys.argv[1]), 30_000, 4, 2)


def disc_vs_bbox(config=3, sample=20000, sw=4, sh=4):
    """Cheap conservative refinement of the bounding-box test: {q <= chi} lies inside the disc of radius sqrt(chi / lambda_min(conic))
    around the centre, so a sub-tile whose nearest point is farther than that cannot be touched."""
    pos, scale, q, op, H, W, fx = scene(config)
    u, v, ex, ey, A11, A12, A22, o = project(pos, scale, q, op, H, W, fx)
    idx = np.random.default_rng(1).choice(len(u), size=min(sample, len(u)), replace=False)
    u, v, ex, ey, A11, A12, A22 = (a[idx] for a in (u, v, ex, ey, A11, A12, A22))
    lam = 0.5 * (A11 + A22) - np.sqrt(0.25 * (A11 - A22) ** 2 + A12 ** 2)
    r2 = 6.25 / lam * 1.002 + 0.05
    x0 = np.clip(np.ceil(u - ex), 0, W - 1); x1 = np.clip(np.floor(u + ex), 0, W - 1)
    y0 = np.clip(np.ceil(v - ey), 0, H - 1); y1 = np.clip(np.floor(v + ey), 0, H - 1)
    bb = dd = 0
    for i in range(len(u)):
        for ty in range(int(y0[i]) // sh, int(y1[i]) // sh + 1):
            for tx in range(int(x0[i]) // sw, int(x1[i]) // sw + 1):
                bb += 1
                dx = max(tx * sw - u[i], 0.0, u[i] - (tx * sw + sw - 1))
                dy = max(ty * sh - v[i], 0.0, v[i] - (ty * sh + sh - 1))
                dd += dx * dx + dy * dy <= r2[i]
    print(f"  sub-tiles {sw}x{sh}: bounding-box pairs {bb}, box + disc pairs {dd}: {dd / bb:.3f}")
