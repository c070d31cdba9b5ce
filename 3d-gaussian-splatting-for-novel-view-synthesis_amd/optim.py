"""Optimiser step of the reference's training loop on the MI355X ("next" row 2).

Mirrors the call sites of scripts/train.py: `optim.Adam([...six groups...], lr=lr, eps=1e-15)` (:394-401), the position
learning-rate schedule (:446-457), `clip_grad_norm_(model.pos, max_norm=1.0)` (:536) and `optimizer.step()` (:538).
`GaussianAdam` takes torch-style param groups; `step()` runs ONE fused HIP kernel for all of them (csrc/gsplat_optim.hip); the
clip coefficient is computed and applied on the device, so a training step has no host synchronisation here.
"""
import contextlib
import ctypes as C

import torch

from . import _abi, ops
from .ops import _p, _stage, _stream_ptr


def position_lr(iteration, position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01,
                position_lr_max_steps=30000):
    """Exponential decay with the initial x0.01 delay phase (scripts/train.py:446-457)."""
    if iteration < position_lr_max_steps:
        lr = position_lr_init * (position_lr_final / position_lr_init) ** (iteration / position_lr_max_steps)
    else:
        lr = position_lr_final
    if iteration < position_lr_delay_mult * position_lr_max_steps:
        lr *= 0.01
    return lr


def reference_param_groups(model, position_lr_init=0.00016, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005,
                           rotation_lr=0.001):
    """The six groups of scripts/train.py:394-401; `model` has pos, opacity_raw, f_dc, f_rest, scale_raw, q_raw."""
    return [{'params': [model.pos], 'lr': position_lr_init, 'name': 'pos'},
            {'params': [model.opacity_raw], 'lr': opacity_lr, 'name': 'opacity'},
            {'params': [model.f_dc], 'lr': feature_lr, 'name': 'f_dc'},
            {'params': [model.f_rest], 'lr': feature_lr / 20.0, 'name': 'f_rest'},
            {'params': [model.scale_raw], 'lr': scaling_lr, 'name': 'scale'},
            {'params': [model.q_raw], 'lr': rotation_lr, 'name': 'rotation'}]


class _RestUpdate:
    """What ops._backward_impl needs to fold the Adam step of one parameter (f_rest) into the backward pass."""

    def __init__(self, opt, param, lr):
        self.opt, self.param, self.lr, self.applied = opt, param, lr, False

    def matches(self, f_rest32, src_ptr):
        p = self.param
        return (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and src_ptr == p.data_ptr()
                and f_rest32.data_ptr() == p.data_ptr() and p.data_ptr() % 16 == 0)

    def begin(self):
        st = self.opt._state(self.param)
        st['step'] += 1
        self.applied = True
        group = _abi.AdamGroup(self.param.numel(), _p(self.param).value, None, _p(st['exp_avg']).value, _p(st['exp_avg_sq']).value,
                               float(self.lr), int(st['step']), None)
        return group, float(self.opt.betas[0]), float(self.opt.betas[1]), float(self.opt.eps)

    def rollback(self):
        """The frame turned out to have outgrown its buffers (or to be off screen): the kernel stepped nothing, so nothing counts."""
        if self.applied:
            self.opt._state(self.param)['step'] -= 1
            self.applied = False


class GaussianAdam:
    """torch.optim.Adam (amsgrad off, no weight decay) restricted to what the reference uses, fused per tensor.

    `clip` = (parameter tensor, max_norm): clip_grad_norm_ on that tensor before the step, fused into its update."""

    def __init__(self, param_groups, lr=0.01, betas=(0.9, 0.999), eps=1e-15):
        self.param_groups = []
        for g in param_groups:
            g = dict(g)
            g.setdefault('lr', lr)
            g['params'] = list(g['params'])
            self.param_groups.append(g)
        self.betas, self.eps = betas, eps
        self.state = {}
        self._clip_buf = None

    def zero_grad(self, set_to_none=True):
        for g in self.param_groups:
            for p in g['params']:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()

    def _state(self, p):
        st = self.state.get(p)
        if st is None:
            st = self.state[p] = {'step': 0, 'exp_avg': torch.zeros_like(p, memory_format=torch.contiguous_format),
                                  'exp_avg_sq': torch.zeros_like(p, memory_format=torch.contiguous_format)}
        return st

    @contextlib.contextmanager
    def fused_rest_update(self, param):
        """For an iteration of ONE view rendered inside ops.deferred_checks(): the backward pass applies this optimiser's step of
        `param` (the model's f_rest: 81 % of all parameters) as the gradient is formed, param.grad stays None and step() skips it.
        Yields the hook: call hook.rollback() when the frame's checks fail afterwards (ops.PairCapacityExceeded, off-screen) -- the
        kernel itself stepped nothing in that case."""
        lr = next(g['lr'] for g in self.param_groups if any(p is param for p in g['params']))
        hook = _RestUpdate(self, param, lr)
        ops.set_rest_update(hook)
        try:
            yield hook
        finally:
            ops.set_rest_update(None)

    @torch.no_grad()
    def clip_grad_norm_(self, param, max_norm=1.0):
        """Device-side clip_grad_norm_ for one tensor; returns the [coef, norm] device tensor (no host read)."""
        lib = _abi.lib()
        g = param.grad
        if g is None:
            return None
        if not (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous()):
            raise RuntimeError("GaussianAdam needs contiguous fp32 GPU gradients (there is no CPU fallback)")
        dev = g.device
        with torch.cuda.device(dev):
            if self._clip_buf is None or self._clip_buf[0].device != dev:
                self._clip_buf = (torch.empty(2, dtype=torch.float32, device=dev),
                                  torch.empty(lib.gsplat_clip_scratch_bytes(), dtype=torch.uint8, device=dev))
            out, scratch = self._clip_buf
            _abi.check(lib.gsplat_clip_grad_norm(g.numel(), _p(g), float(max_norm), _p(out), _p(scratch), _stream_ptr(dev)),
                       "gsplat_clip_grad_norm")
        self._pending_clip = (param, out)
        return out

    @torch.no_grad()
    def step(self):
        """One launch per device for all parameter groups (gsplat_adam_step_multi, eight tensors per launch)."""
        lib = _abi.lib()
        pending = getattr(self, "_pending_clip", None)
        per_device = {}
        for g in self.param_groups:
            for p in g['params']:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()):
                    raise RuntimeError("GaussianAdam needs contiguous fp32 GPU parameters (there is no CPU fallback)")
                st = self._state(p)
                st['step'] += 1
                scale = pending[1] if pending is not None and pending[0] is p else None
                per_device.setdefault(p.device, []).append(
                    _abi.AdamGroup(p.numel(), _p(p).value, _p(p.grad).value, _p(st['exp_avg']).value, _p(st['exp_avg_sq']).value,
                                   float(g['lr']), int(st['step']), _p(scale).value if scale is not None else None))
        with _stage("adam"):
            for dev, groups in per_device.items():
                with torch.cuda.device(dev):
                    for k in range(0, len(groups), 8):
                        chunk = groups[k:k + 8]
                        arr = (_abi.AdamGroup * len(chunk))(*chunk)
                        _abi.check(lib.gsplat_adam_step_multi(len(chunk), arr, float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                                              _stream_ptr(dev)), "gsplat_adam_step_multi")
        self._pending_clip = None
