"""Shared helpers for the parity tests: golden loading and the stated fp32 tolerances."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PARAMS = ["pos", "scale_raw", "q_raw", "opacity_raw", "f_dc", "f_rest"]
RENDER_CASES = ["g1_generic", "g2_ragged", "g3_occlusion", "g4_thresholds", "g5_guardband", "g6_huge", "g7_tiny",
                "g8_deg0", "g12_kwargs", "g14_T8", "g14_T32"]
EMPTY_CASES = ["g9a_empty_opacity", "g9b_empty_behind"]

# Stated fp32 tolerance of the HIP path against float64 results of the reference / the oracle = SURVEY.md 8c:
#   image |delta| <= 1e-5 on >= 99.99 % of the values and <= 5e-3 on the rest; every gradient tensor rel-L2 <= 1e-3, max-abs / max <= 2e-3.
# The render has hard thresholds (q <= chi_square_clip, alpha >= alpha_cutoff, T > 5e-5): ANY fp32 evaluation -- the reference's
# own included -- flips a few of them against float64, and one chi flip is worth up to opacity * exp(-chi / 2) = 4.4e-2.  So a
# bound of SURVEY 8c may only be exceeded next to a MEASURED fp32-vs-fp64 disagreement of the reference's arithmetic on the
# same inputs (the reference's fp32 outputs stored in the goldens, or the oracle run in float32), by at most K_CAL times.
IMG_TOL_BULK = 1e-5      # |delta| allowed on >= IMG_BULK_FRAC of the values
IMG_BULK_FRAC = 0.9999
IMG_TOL_REST = 5e-3      # |delta| allowed on the rest ...
IMG_TOL_FLIP = 4.4e-2    # ... except threshold flips (0.999 * exp(-6.25 / 2) = 0.0439), counted against the calibration
GRAD_TOL_L2 = 1e-3       # ||g - g_ref|| / ||g_ref|| per gradient tensor
GRAD_TOL_MAX = 2e-3      # max|g - g_ref| / max|g_ref|
K_CAL = 3.0


def load(name):
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    H, W, fx, fy, cx, cy = d["cam"]
    d["H"], d["W"], d["fx"], d["fy"], d["cx"], d["cy"] = int(H), int(W), float(fx), float(fy), float(cx), float(cy)
    d["kwargs"] = {str(k): float(v) for k, v in zip(d["kw_names"], d["kw_vals"])}
    for k in ("pix_guard", "T"):
        if k in d["kwargs"]:
            d["kwargs"][k] = int(d["kwargs"][k])
    return d


def tensors(d, dtype, device="cpu", grad=False, names=PARAMS):
    return {k: torch.tensor(d[k], dtype=dtype, device=device).requires_grad_(grad) for k in names}


def cam_args(d):
    return (d["H"], d["W"], d["fx"], d["fy"], d["cx"], d["cy"])


def image_errors(img, ref):
    d = np.abs(np.asarray(img, dtype=np.float64) - np.asarray(ref, dtype=np.float64))
    return {"bad": float((d > IMG_TOL_BULK).mean()), "big": int((d > IMG_TOL_REST).sum()), "max": float(d.max()), "mean": float(d.mean()),
            "n": int(d.size)}


def grad_errors(g, ref):
    g, ref = np.asarray(g, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    nr, mr = np.linalg.norm(ref), np.abs(ref).max() if ref.size else 0.0
    if nr == 0:
        return {"l2": 0.0 if np.abs(g).max(initial=0.0) == 0 else np.inf, "mx": 0.0 if np.abs(g).max(initial=0.0) == 0 else np.inf}
    return {"l2": float(np.linalg.norm(g - ref) / nr), "mx": float(np.abs(g - ref).max() / mr)}


def check_image(img, ref, cal=None, what="image", flip=IMG_TOL_FLIP, frac=None):
    """img (HIP, fp32) against ref (float64).  cal = an fp32 evaluation of the reference's arithmetic on the same inputs (or its
    image_errors() dict measured elsewhere on the same scene): with it the flip fraction may reach K_CAL x the calibration's; without
    it SURVEY 8c's bounds apply as they stand.  `frac` overrides the bulk fraction only together with a printed reason at the call."""
    img = np.asarray(img, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert img.shape == ref.shape, (img.shape, ref.shape)
    assert np.isfinite(img).all(), f"{what}: non-finite values"
    e = image_errors(img, ref)
    c = cal if isinstance(cal, dict) or cal is None else image_errors(cal, ref)
    # (+ 3 values = ONE pixel when a calibration is given: a single flipped pixel where the fp32 reference happens to have none is
    # a Poisson event of this small a count, not a defect -- small images have fewer than 10^4 pixels)
    allowed_bad = max(1.0 - (IMG_BULK_FRAC if frac is None else frac), K_CAL * c["bad"] if c else 0.0) + (3.0 / img.size if c else 0.0)
    # values beyond 5e-3 are threshold flips: as many as K_CAL x the calibration's (+ one pixel: they are rare events), none without one
    n_cal_big = c["big"] * img.size / c["n"] if c else 0          # (a calibration measured on a window of the scene is scaled to the frame)
    allowed_big = int(np.ceil(K_CAL * n_cal_big)) + (3 if c else 0)       # (+3 = the three channels of ONE pixel, as above)
    print(f"{what}: beyond {IMG_TOL_BULK}: {e['bad']:.2e} (allowed {allowed_bad:.2e}, fp32 reference {c['bad'] if c else float('nan'):.2e}); "
          f"beyond {IMG_TOL_REST}: {e['big']} (allowed {allowed_big}); max {e['max']:.2e}, mean {e['mean']:.2e}")
    assert e["bad"] <= allowed_bad, f"{what}: {e['bad']:.3e} of the values beyond {IMG_TOL_BULK} (allowed {allowed_bad:.3e})"
    assert e["big"] <= allowed_big, f"{what}: {e['big']} values beyond {IMG_TOL_REST} (allowed {allowed_big})"
    assert e["max"] <= (flip if c else IMG_TOL_REST) * (1 + 1e-3), f"{what}: max |delta| {e['max']:.3e}"


def check_grad(g, ref, name, cal=None, l2=GRAD_TOL_L2, mx=GRAD_TOL_MAX, band=None):
    """One gradient tensor against float64.  cal = the same gradient from an fp32 evaluation of the reference's arithmetic (or its
    grad_errors() dict): the bounds become max(K_CAL x the calibration's error).  band (same shape, >= 0; only for a
    case whose cause is on record): what moving a hard threshold within fp32's reach does to every entry in float64 -- the
    error is measured outside that band (an entry inside it is as right as ANY fp32 evaluation of the thresholds can be)."""
    g = np.asarray(g, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    if band is not None:
        d = g - ref
        g = ref + np.sign(d) * np.maximum(np.abs(d) - K_CAL * np.asarray(band, dtype=np.float64), 0.0)
    assert g.shape == ref.shape, (name, g.shape, ref.shape)
    assert np.isfinite(g).all(), f"grad {name}: non-finite values"
    if np.linalg.norm(ref) == 0:
        assert np.abs(g).max(initial=0.0) == 0, f"grad {name}: expected exact zeros"
        return
    e = grad_errors(g, ref)
    c = cal if isinstance(cal, dict) or cal is None else grad_errors(cal, ref)
    a2, am = max(l2, K_CAL * c["l2"] if c else 0.0), max(mx, K_CAL * c["mx"] if c else 0.0)
    print(f"grad {name}: rel-L2 {e['l2']:.2e} (allowed {a2:.2e}, fp32 reference {c['l2'] if c else float('nan'):.2e}); "
          f"max/max {e['mx']:.2e} (allowed {am:.2e}, fp32 reference {c['mx'] if c else float('nan'):.2e})")
    assert e["l2"] <= a2, f"grad {name}: rel-L2 {e['l2']:.3e} > {a2:.3e}"
    assert e["mx"] <= am, f"grad {name}: max-abs/max {e['mx']:.3e} > {am:.3e}"
