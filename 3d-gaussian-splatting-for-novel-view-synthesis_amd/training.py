"""One training iteration of the reference's loop on the MI355X ("next" row 2: optimiser / training step).

Follows scripts/train.py:446-569 statement by statement, on the fused pieces of this package:

    position learning rate (:446-457)          optim.position_lr
    zero_grad (:460)                           GaussianAdam.zero_grad
    per view of the batch (:471-527)           ops.render_gaussians (covariance build + SH folded in; the reference's
                                               build_sigma + evaluate_sh + render) -> losses.compute_loss_device,
                                               loss / batch size, accumulated
    backward (:530)                            per view, gradients accumulate in .grad (same sum, frees each view's buffers)
    [data parallel]                            dp.allreduce_gradients: sum over ranks (not in the reference)
    clip_grad_norm_(pos, 1.0) + step (:536-538) GaussianAdam.clip_grad_norm_ / .step (device-side coefficient, no host sync)
    densify / prune every `densification_interval` (:544-561)   model.GaussianModel.densify_and_prune + a fresh optimiser
                                               with the CURRENT position learning rate, as there
    opacity reset every `opacity_reset_interval` (:564-569)     GaussianModel.reset_opacity

With data parallelism every rank holds the full model, renders its share of the batch's views and divides by the GLOBAL
number of views; the split noise of densification comes from a generator seeded identically on all ranks, so the
replicas stay bit-identical without a broadcast.
"""
import contextlib
from dataclasses import dataclass

import torch
import torch.distributed as dist

from . import dp, losses, ops, optim


@dataclass
class TrainConfig:
    """Hyper-parameters of scripts/train.py:222-251 (same names, same defaults)."""
    iterations: int = 30000
    lr: float = 0.01
    position_lr_init: float = 0.00016
    position_lr_final: float = 0.0000016
    position_lr_delay_mult: float = 0.01
    position_lr_max_steps: int = 30000
    feature_lr: float = 0.0025
    opacity_lr: float = 0.05
    scaling_lr: float = 0.005
    rotation_lr: float = 0.001
    lambda_l1: float = 0.8
    lambda_ssim: float = 0.2
    densify_until_iter: int = 15000
    densification_interval: int = 100
    opacity_reset_interval: int = 3000
    prune_opacity_threshold: float = 0.01
    max_grad: float = 0.01
    scale_threshold: float = 0.01
    checkpoint_interval: int = 1000
    densify_seed: int = 0
    # not a reference hyper-parameter: the views of one iteration alternate between this many HIP streams (1 or 2), so that the
    # latency-bound front of view k + 1 (projection, binning) runs beside the VALU-bound rasterisation of view k: ~10 % per view
    # at config 3.  Gradients accumulate in view order either way.
    view_streams: int = 2
    # not a reference hyper-parameter either: an iteration of ONE view on one process applies the Adam step of f_rest (81 % of the
    # parameters) inside the projection backward -- its gradient is neither written nor read back by the optimiser (0.07 ms of 1.0
    # at config 3); the same arithmetic as the optimiser's kernel, bit for bit.
    fold_rest_step: bool = True
    # an iteration of SEVERAL views on one process: the projection backward adds each view's gradients to one buffer itself
    # (ops.accumulate_grads) instead of autograd's accumulation pass per view -- the same sums in the same order.
    sum_views_in_kernel: bool = True


_side_streams = {}           # per device: the two streams the views of an iteration alternate between (TrainConfig.view_streams)


class Trainer:
    """Owns the optimiser of a GaussianModel and runs iterations.  `group`: torch.distributed group for data parallelism
    by camera view (None = default group if initialised, else single process)."""

    def __init__(self, model, config=None, group=None):
        self.model = model
        self.cfg = config or TrainConfig()
        self.group = group
        self.optimizer = self._new_optimizer(self.cfg.position_lr_init)
        self._gen = None

    def _new_optimizer(self, pos_lr):
        c = self.cfg
        groups = optim.reference_param_groups(self.model, position_lr_init=pos_lr, feature_lr=c.feature_lr,
                                              opacity_lr=c.opacity_lr, scaling_lr=c.scaling_lr, rotation_lr=c.rotation_lr)
        return optim.GaussianAdam(groups, lr=c.lr, eps=1e-15)

    def _view_streams(self, dev):
        key = (dev.type, dev.index)
        got = _side_streams.get(key)
        if got is None:
            got = _side_streams[key] = tuple(torch.cuda.Stream(dev) for _ in range(2))
            # the parameters' AccumulateGrad nodes live on the caller's stream and the views' backward passes on the side streams: that
            # is the point (the engine orders the two); torch would warn about it in every iteration
            quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
            if quiet is not None:
                quiet(False)
        return got

    def _world(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group)
        return 1

    def _densify_generator(self, iteration):
        # same stream of split noise on every rank: CPU generator keyed by (seed, iteration)
        g = torch.Generator()
        g.manual_seed(self.cfg.densify_seed * 1000003 + iteration)
        return g

    def step(self, iteration, views, global_views=None, views_per_rank=None):
        """One iteration on this rank's `views` (list of dicts with image [H,W,3], c2w [4,4], H, W, fx, fy, cx, cy — the
        sample dict of data.GaussianDataset).  Data parallel: the ranks may hold different numbers of views; how many each
        holds is agreed COLLECTIVELY (it selects the exchange's sequence of collectives): pass `views_per_rank` (one count
        per rank, the same list on every rank) or leave it None and the counts are exchanged with one tiny all-gather.
        `global_views` (views of the whole batch over all ranks) is only checked against that.  Returns {'loss', 'l1',
        'ssim'} as device scalars (this rank's share, already divided by the global batch), 'gaussians', 'lr_pos',
        'densified'."""
        c, m = self.cfg, self.model
        world = self._world()
        dev = m.pos.device
        even, n_global = dp.agree_on_views(len(views), self.group, views_per_rank, device=dev)
        if global_views is not None and int(global_views) != n_global:
            raise ValueError(f"global_views={global_views}, but the ranks hold {n_global} views in all")
        pos_lr = optim.position_lr(iteration, c.position_lr_init, c.position_lr_final, c.position_lr_delay_mult,
                                   c.position_lr_max_steps)
        self.optimizer.param_groups[0]['lr'] = pos_lr
        for attempt in range(4):
            self.optimizer.zero_grad()
            acc = torch.zeros(3, dtype=torch.float32, device=dev)
            # data parallel: SH gradients travel in factored form (dp.FactoredExchange, 2.6x fewer bytes over xGMI at 8 views)
            exchange = dp.FactoredExchange(m.get_params(), world_views=1, group=self.group, equal_views=even) if world > 1 else None
            # no host synchronisation per view: the renders size their buffers from earlier frames, the per-frame checks
            # (off-screen exception, buffer capacity) are made ONCE, after the last backward is queued
            pass_error = None
            fold = c.fold_rest_step and world == 1 and len(views) == 1
            rest_hook = None
            try:
                sum_in_kernel = c.sum_views_in_kernel and world == 1 and len(views) > 1      # the views' gradients summed by the projection backward itself
                with ops.deferred_checks() as checks, (exchange if exchange is not None else contextlib.nullcontext()), \
                        (ops.accumulate_grads(m.get_params()) if sum_in_kernel else contextlib.nullcontext()) as grad_acc, \
                        (self.optimizer.fused_rest_update(m.f_rest) if fold else contextlib.nullcontext()) as rest_hook:
                    side = self._view_streams(dev) if (c.view_streams > 1 and len(views) > 1 and world == 1) else ()     # (one process: the
                    # exchange's collectives of a data-parallel pass stay on the caller's stream)
                    main = torch.cuda.current_stream(dev) if side else None
                    for st in side:
                        st.wait_stream(main)                                           # parameters, zeroed gradients
                    per_view = []
                    for k, v in enumerate(views):                                      # (the gradient sink is always removed again)
                        with (torch.cuda.stream(side[k % len(side)]) if side else contextlib.nullcontext()):
                            image_gt = torch.as_tensor(v['image']).to(dev)
                            c2w = torch.as_tensor(v['c2w'], dtype=torch.float32).to(dev)
                            rendered = ops.render_gaussians(m.pos, m.f_dc, m.f_rest, m.opacity_raw, m.scale_raw, m.q_raw, c2w,
                                                            int(v['H']), int(v['W']), float(v['fx']), float(v['fy']), float(v['cx']), float(v['cy']))
                            loss, vals = losses.compute_loss_device(rendered, image_gt, c.lambda_l1, c.lambda_ssim, scale=1.0 / n_global)
                            loss.backward()                            # (loss / batch size: the division is inside the loss kernels)
                            per_view.append(vals)
                    for st in side:
                        main.wait_stream(st)
                    if grad_acc is not None:
                        grad_acc.assign()
                    for vals in per_view:                                              # (on the caller's stream, in view order)
                        if side:
                            vals.record_stream(main)
                        acc += vals
            except Exception as e:                # single process: nothing to agree on, the exception leaves as it is
                if world == 1:
                    raise
                pass_error = e
            # ---- what happened on this rank, as ONE code; then ONE agreement over the ranks.  Nothing may leave this function
            # between the first collective of the pass and the agreement: a rank that raised here while its peers sat in the
            # exchange would leave them waiting for ever.
            status, err = dp.STATUS_OK, None
            try:
                checks.verify()
                if exchange is not None and exchange.n_added != len(views):
                    raise RuntimeError(f"{exchange.n_added} of this rank's {len(views)} views went through the factored exchange: a render "
                                       "of this model took the ordinary backward (are f_dc / f_rest the model's own tensors?)")
            except ops.PairCapacityExceeded:
                status = dp.STATUS_REDO           # a view outgrew the buffers: this pass's gradients are invalid (capacity now raised)
            except Exception as e:                # the reference's off-screen Exception (render.py:235-236), or anything else
                status, err = (dp.STATUS_OFFSCREEN if str(e) == ops.OFFSCREEN_MSG else dp.STATUS_ERROR), e
            if status != dp.STATUS_OK and rest_hook is not None:
                rest_hook.rollback()              # (the kernel stepped nothing for a frame that overflowed or is off screen: nothing counts)
            if pass_error is not None:            # an exception inside the render loop itself (a frame that waited for its counters, a device error)
                err = pass_error
                status = dp.STATUS_OFFSCREEN if str(err) == ops.OFFSCREEN_MSG else dp.STATUS_ERROR
            if world > 1:                         # every rank repeats the pass, or none does; every rank raises, or none does
                if exchange is not None:
                    exchange.pad_views(len(views))          # (a pass that stopped early: keep the sequence of collectives aligned)
                status = dp.agree_status(status, self.group, device=dev)
            if exchange is not None:
                if status != dp.STATUS_OK:
                    exchange.abandon()
                else:
                    exchange.finish()             # the loss was already divided by the global batch: world_views = 1
            if status >= dp.STATUS_OFFSCREEN:
                if err is not None:
                    raise err
                if status == dp.STATUS_OFFSCREEN:
                    raise Exception(ops.OFFSCREEN_MSG + " (on another rank of the data-parallel group)")
                raise RuntimeError("another rank of the data-parallel group failed in this training step")
            if status == dp.STATUS_OK:
                break
        else:
            raise RuntimeError("the pair buffers overflowed four times in a row")
        names = dp.PARAM_NAMES
        folded = rest_hook is not None and rest_hook.applied      # f_rest was stepped inside the backward pass: no gradient, no second step
        for k in names:
            p = getattr(m, k)
            if p.grad is None and not (folded and p is m.f_rest):
                p.grad = torch.zeros_like(p)
        self.optimizer.clip_grad_norm_(m.pos, max_norm=1.0)
        self.optimizer.step()
        densified = False
        if iteration < c.densify_until_iter and iteration % c.densification_interval == 0:
            grads = {'pos': m.pos.grad, 'opacity_raw': m.opacity_raw.grad}
            m.densify_and_prune(grads, opacity_threshold=c.prune_opacity_threshold, max_grad=c.max_grad,
                                scale_threshold=c.scale_threshold, generator=self._densify_generator(iteration))
            self.optimizer = self._new_optimizer(pos_lr)
            densified = True
        if iteration % c.opacity_reset_interval == 0:
            m.reset_opacity()
        return {'loss': acc[2], 'l1': acc[0], 'ssim': acc[1], 'gaussians': m.get_num_gaussians(), 'lr_pos': pos_lr,
                'densified': densified}
