"""ctypes mirror of include/gsplat_mi355x.h and the loader of the HIP library.

The product path has NO CPU fallback: if libgsplat_mi355x.so is missing or does not export every
symbol the header declares, `lib()` raises and every op fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSPLAT_MI355X_LIB: explicit path of another build of the same library (tools/ use it for the diagnostics build)
LIB_PATH = os.environ.get("GSPLAT_MI355X_LIB") or os.path.join(_HERE, "csrc", "libgsplat_mi355x.so")

GSPLAT_OK = 0
GSPLAT_SCENE_OK = 0
GSPLAT_SCENE_ALL_CULLED = 10
GSPLAT_SCENE_ALL_OFFSCREEN = 11
ABI_VERSION = 8
GSPLAT_PROJECT_COLOUR_FUSED = 1
GSPLAT_PROJECT_COUNTS_MAPPED = 2
GSPLAT_PROJECT_SAVE_SH_JACOBIAN = 4
GSPLAT_PROJECT_COUNTS_LATE = 8
GSPLAT_BACKWARD_SH_JACOBIAN = 1
GSPLAT_FRAME_BACKWARD = 1
GSPLAT_FRAME_NO_SH_JACOBIAN = 2
GSPLAT_BACKWARD_PHASE_RASTER = 2
GSPLAT_BACKWARD_PHASE_PROJECT = 4
GSPLAT_BACKWARD_GRAD2D_DIRTY = 8
GSPLAT_BACKWARD_ACCUMULATE = 16

_F = C.POINTER(C.c_float)


class View(C.Structure):
    """gsplat_view: intrinsics + the 8 keyword arguments of render() (reference render.py:62-64)."""
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("near_z", C.c_float), ("far_z", C.c_float), ("pix_guard", C.c_float),
                ("tile", C.c_int32), ("min_conis", C.c_float), ("chi_square_clip", C.c_float),
                ("alpha_max", C.c_float), ("alpha_cutoff", C.c_float)]


class Gaussians(C.Structure):
    _fields_ = [("n", C.c_int64), ("pos", C.c_void_p), ("opacity_raw", C.c_void_p), ("color", C.c_void_p),
                ("sigma", C.c_void_p), ("scale_raw", C.c_void_p), ("q_raw", C.c_void_p), ("f_dc", C.c_void_p),
                ("f_rest", C.c_void_p)]


class GaussianGrads(C.Structure):
    _fields_ = [("pos", C.c_void_p), ("opacity_raw", C.c_void_p), ("color", C.c_void_p), ("sigma", C.c_void_p),
                ("scale_raw", C.c_void_p), ("q_raw", C.c_void_p), ("f_dc", C.c_void_p), ("f_rest", C.c_void_p)]


class AdamGroup(C.Structure):
    _fields_ = [("n", C.c_int64), ("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("lr", C.c_float), ("step", C.c_int32), ("grad_scale", C.c_void_p)]


class Counts(C.Structure):
    _fields_ = [("n_survivors", C.c_int32), ("n_visible", C.c_int32), ("n_pairs", C.c_int64),
                ("max_tiles_per_gaussian", C.c_int32), ("reserved", C.c_int32), ("n_binned", C.c_int64)]


_VP, _I64, _INT = C.c_void_p, C.c_int64, C.c_int
_PV, _PG, _PGG, _PC = C.POINTER(View), C.POINTER(Gaussians), C.POINTER(GaussianGrads), C.POINTER(Counts)

# name -> (restype, argtypes); must list every function include/gsplat_mi355x.h declares
SIGNATURES = {
    "gsplat_abi_version": (_INT, []),
    "gsplat_last_error": (C.c_char_p, []),
    "gsplat_classify_counts": (_INT, [_PC]),
    "gsplat_project_state_bytes": (_I64, [_I64, _PV]),
    "gsplat_project_scratch_bytes": (_I64, [_I64]),
    "gsplat_bin_state_bytes": (_I64, [_I64, _PV]),
    "gsplat_bin_scratch_bytes": (_I64, [_I64, _PV]),
    "gsplat_project": (_INT, [_PG, _VP, _PV, _VP, _VP, _I64, _VP, _VP, C.c_int32, _VP]),
    "gsplat_bin": (_INT, [_I64, _I64, _PV, _VP, _VP, _VP, _I64, _VP]),
    "gsplat_rasterize_forward": (_INT, [_I64, _I64, _PV, _VP, _VP, _VP, _VP, _VP, _VP]),
    "gsplat_rasterize_backward_scratch_bytes": (_I64, [_I64, _I64]),
    "gsplat_rasterize_backward": (_INT, [_I64, _I64, _PV, _VP, _VP, _VP, _VP, _VP, C.c_int32, _VP, _I64, _VP]),
    "gsplat_project_backward": (_INT, [_PG, _VP, _PV, _VP, _VP, _PGG, C.c_int32, _VP]),
    "gsplat_logit_grad": (_INT, [_I64, _PV, _VP, _VP, _VP, _VP]),
    "gsplat_frame_bytes": (_I64, [_I64, _I64, _PV, C.c_int32]),
    "gsplat_forward_deferred": (_INT, [_PG, _VP, _PV, _VP, _I64, _I64, _VP, _I64, _VP, _I64, _VP, _VP, _VP, C.c_int32, _VP]),
    "gsplat_backward": (_INT, [_PG, _VP, _PV, _VP, _I64, _I64, _VP, _PGG, _VP, _VP, _I64, C.c_int32, _VP]),
    "gsplat_sh_accumulate": (_INT, [_I64, C.c_int32, _VP, _VP, _VP, C.c_float, _VP, _VP, _VP]),
    "gsplat_build_sigma": (_INT, [_I64, _VP, _VP, _VP, _VP]),
    "gsplat_build_sigma_backward": (_INT, [_I64, _VP, _VP, _VP, _VP, _VP, _VP]),
    "gsplat_evaluate_sh": (_INT, [_I64, _VP, _VP, _VP, _VP, _VP, _VP]),
    "gsplat_evaluate_sh_backward": (_INT, [_I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "gsplat_loss_scratch_bytes": (_I64, [_I64, C.c_int32, C.c_int32, C.c_int32]),
    "gsplat_loss": (_INT, [_VP, _VP, _I64, C.c_int32, C.c_int32, C.c_float, C.c_float, _VP, _VP, _VP, _VP]),
    "gsplat_loss_forward": (_INT, [_VP, _VP, _I64, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, _VP, _VP, _VP, C.c_int32, _VP]),
    "gsplat_loss_backward": (_INT, [_VP, _VP, _I64, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, _VP, _VP, _VP, _VP]),
    "gsplat_clip_scratch_bytes": (_I64, []),
    "gsplat_clip_grad_norm": (_INT, [_I64, _VP, C.c_float, _VP, _VP, _VP]),
    "gsplat_adam_step": (_INT, [_I64, _VP, _VP, _VP, _VP, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, _VP, _VP]),
    "gsplat_adam_step_multi": (_INT, [C.c_int32, C.POINTER(AdamGroup), C.c_float, C.c_float, C.c_float, _VP]),
    "gsplat_backward_adam_rest": (_INT, [_PG, _VP, _PV, _VP, _I64, _I64, _VP, _PGG, _VP, _I64, C.c_int32, C.POINTER(AdamGroup), C.c_float, C.c_float,
                                         C.c_float, _VP]),
}

_lib = None


class GsplatLibraryError(RuntimeError):
    pass


def lib():
    """Load libgsplat_mi355x.so (once).  Raises GsplatLibraryError if it is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GsplatLibraryError(
            f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (or `make -C "
            f"{os.path.dirname(LIB_PATH)}`).  There is no CPU fallback for the render path.")
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise GsplatLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    ver = handle.gsplat_abi_version()
    if ver != ABI_VERSION:
        raise GsplatLibraryError(f"ABI version mismatch: library {ver}, Python host {ABI_VERSION}")
    _lib = handle
    return _lib


def check(status, what):
    if status != GSPLAT_OK:
        msg = lib().gsplat_last_error()
        raise RuntimeError(f"{what} failed with status {status}: {msg.decode() if msg else ''}")


def make_view(H, W, fx, fy, cx, cy, near=0.01, far=100.0, pix_guard=32, T=16, min_conis=1e-6, chi_square_clip=6.25,
              alpha_max=0.99, alpha_cutoff=1 / 128.):
    return View(int(H), int(W), float(fx), float(fy), float(cx), float(cy), float(near), float(far), float(pix_guard),
                int(T), float(min_conis), float(chi_square_clip), float(alpha_max), float(alpha_cutoff))
