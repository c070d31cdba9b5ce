"""ctypes loader of the plain-C oracle (oracle/gs_oracle.c).  TEST INFRASTRUCTURE: imported only by tests/."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libgs_oracle.so")


class View(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("near_z", C.c_double), ("far_z", C.c_double), ("pix_guard", C.c_double), ("T", C.c_int32),
                ("min_conis", C.c_double), ("chi", C.c_double), ("alpha_max", C.c_double), ("alpha_cutoff", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None and os.environ.get("GS_ORACLE_LIB"):   # `make check-asan`: the sanitizer build
        _lib = C.CDLL(os.environ["GS_ORACLE_LIB"])
        _lib.ora_render.restype = C.c_int
    if _lib is None:
        src = os.path.join(HERE, "gs_oracle.c")
        if not os.path.exists(SO) or os.path.getmtime(src) > os.path.getmtime(SO):
            subprocess.check_call(["make", "-C", HERE])
        _lib = C.CDLL(SO)
        _lib.ora_render.restype = C.c_int
    return _lib


def render(s, H, W, fx, fy, cx, cy, grad_image=None, near=0.01, far=100.0, pix_guard=32, T=16, min_conis=1e-6, chi_square_clip=6.25,
           alpha_max=0.99, alpha_cutoff=1 / 128.):
    """s: dict of arrays (pos, f_dc, f_rest, opacity_raw, scale_raw, q_raw, c2w).  Returns (status, image, grads, (V, P))."""
    a = {k: np.ascontiguousarray(s[k], np.float64) for k in ("pos", "f_dc", "f_rest", "opacity_raw", "scale_raw", "q_raw", "c2w")}
    n = len(a["pos"])
    v = View(int(H), int(W), fx, fy, cx, cy, near, far, float(pix_guard), int(T), min_conis, chi_square_clip, alpha_max, alpha_cutoff)
    img = np.zeros((int(H), int(W), 3))
    counts = np.zeros(2, np.int64)
    p = lambda x: x.ctypes.data_as(C.c_void_p) if x is not None else None
    g = None
    gi = None
    if grad_image is not None:
        gi = np.ascontiguousarray(grad_image, np.float64)
        g = {k: np.zeros_like(a[k]) for k in ("pos", "f_dc", "f_rest", "opacity_raw", "scale_raw", "q_raw")}
    st = lib().ora_render(C.c_int64(n), p(a["pos"]), p(a["f_dc"]), p(a["f_rest"]), p(a["opacity_raw"]), p(a["scale_raw"]), p(a["q_raw"]),
                          p(a["c2w"]), C.byref(v), p(img), p(gi), *(p(g[k]) if g else None for k in
                                                                  ("pos", "f_dc", "f_rest", "opacity_raw", "scale_raw", "q_raw")), p(counts))
    return st, img, g, (int(counts[0]), int(counts[1]))
