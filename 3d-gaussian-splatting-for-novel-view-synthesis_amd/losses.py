"""Training loss of the reference (gaussian_splatting/losses.py) on the MI355X: L1 + SSIM, value and gradient from two
fused HIP kernels (csrc/gsplat_loss.hip) -- SURVEY.md §8(f), "next" row 1.

Same names, arguments and return values as the reference:
    l1_loss(pred, target)                                              losses.py:27
    ssim_loss(pred, target, window_size=11, size_average=True)         losses.py:44
    compute_loss(pred, target, lambda_l1=0.8, lambda_ssim=0.2)         losses.py:158  -> (total, {'l1', 'ssim', 'total'})
pred / target: [H, W, 3] or [B, H, W, 3].  Differentiable w.r.t. `pred` (the reference never needs d/d target).
"""
import ctypes as C

import torch

from . import _abi
from .ops import _f32, _p, _stage, _stream_ptr


class _LossFn(torch.autograd.Function):
    """(total, values): total = scale * (l1w * l1 + sw * (1 - ssim)), differentiable w.r.t. pred; values = scale * (l1, 1 - ssim,
    total), detached.  The forward leaves the SSIM term's partial-derivative maps in scratch; the gradient is written by the backward,
    with the upstream scalar multiplied in by the kernel (gsplat_loss_forward / gsplat_loss_backward)."""

    @staticmethod
    def forward(ctx, pred, target, l1w, sw, scale=1.0):
        lib = _abi.lib()
        shape = tuple(pred.shape)
        if len(shape) not in (3, 4) or shape[-1] != 3 or tuple(target.shape) != shape:
            raise ValueError(f"pred/target must both be [H, W, 3] or [B, H, W, 3], got {shape} and {tuple(target.shape)}")
        b = shape[0] if len(shape) == 4 else 1
        h, w = shape[-3], shape[-2]
        x, y = _f32(pred, shape, "pred"), _f32(target, shape, "target")
        dev = x.device
        need = ctx.needs_input_grad[0]
        with torch.cuda.device(dev):
            values = torch.empty(3, dtype=torch.float32, device=dev)
            total = torch.empty((), dtype=torch.float32, device=dev)
            scratch = torch.empty(lib.gsplat_loss_scratch_bytes(b, h, w, 1 if need else 0), dtype=torch.uint8, device=dev)
            with _stage("loss"):
                _abi.check(lib.gsplat_loss_forward(_p(x), _p(y), b, h, w, float(l1w), float(sw), float(scale), _p(values), _p(total),
                                                   _p(scratch), 1 if need else 0, _stream_ptr(dev)), "gsplat_loss_forward")
        ctx.saved = (x, y, scratch, b, h, w, float(l1w), float(sw), float(scale)) if need else None
        ctx.dtype = pred.dtype
        ctx.mark_non_differentiable(values)
        return total, values

    @staticmethod
    def backward(ctx, g_total, _g_values):
        lib = _abi.lib()
        x, y, scratch, b, h, w, l1w, sw, scale = ctx.saved
        dev = x.device
        up = g_total.detach().to(device=dev, dtype=torch.float32).contiguous()
        with torch.cuda.device(dev):
            grad = torch.empty_like(x)
            with _stage("loss"):
                _abi.check(lib.gsplat_loss_backward(_p(x), _p(y), b, h, w, l1w, sw, scale, _p(up), _p(grad), _p(scratch), _stream_ptr(dev)),
                           "gsplat_loss_backward")
        return (grad if ctx.dtype == torch.float32 else grad.to(ctx.dtype)), None, None, None, None


def compute_loss(pred, target, lambda_l1=0.8, lambda_ssim=0.2):
    """lambda_l1 * L1 + lambda_ssim * (1 - SSIM); returns (total_loss, dict of floats) like the reference."""
    total, v = _LossFn.apply(pred, target, lambda_l1, lambda_ssim)
    l1, ssim, tot = v.tolist()                       # ONE host read instead of the reference's three .item() calls
    return (total if pred.dtype == torch.float32 else total.to(pred.dtype)), {'l1': l1, 'ssim': ssim, 'total': tot}


def l1_loss(pred, target):
    return _LossFn.apply(pred, target, 1.0, 0.0)[0].to(pred.dtype)


def ssim_loss(pred, target, window_size=11, size_average=True):
    if window_size != 11 or not size_average:
        raise NotImplementedError("the fused kernel implements the reference defaults: window_size=11, size_average=True")
    return _LossFn.apply(pred, target, 0.0, 1.0)[0].to(pred.dtype)


def compute_loss_device(pred, target, lambda_l1=0.8, lambda_ssim=0.2, scale=1.0):
    """compute_loss without the host read: returns (total_loss, values) with values = device tensor scale * [l1, 1 - ssim, total]
    (the training step keeps the loss on the GPU and reads it only when the caller logs it).  `scale` (1 / batch size in the
    training loop) is applied by the kernels: the value comes out scaled and so does the gradient."""
    total, v = _LossFn.apply(pred, target, lambda_l1, lambda_ssim, scale)
    return (total if pred.dtype == torch.float32 else total.to(pred.dtype)), v
