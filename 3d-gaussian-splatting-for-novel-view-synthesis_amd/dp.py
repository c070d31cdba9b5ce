"""Data parallelism by camera view: one process per GPU, full parameter replica per rank, no data-path collective
in forward/backward, one all-reduce(sum) of the six parameter gradients per step.

The reference has no distributed code at all (single device, views rendered sequentially, scripts/train.py:471-527,
loss / batch_size at :514-519); this module is the MI355X-native replacement for that loop's batching:
`torch.distributed` with backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.

Bucketing: the gradient message is 236 B per Gaussian (59 floats), of which f_rest is 180 B (76 %).  xGMI is
point-to-point (7 links per GPU), so few large messages are best: f_rest goes as its own bucket (already contiguous,
no copy), the other five tensors (56 B per Gaussian) are flattened into one bucket.  Both are launched async and
waited together, so RCCL can pipeline them.
"""
import torch
import torch.distributed as dist

PARAM_NAMES = ("pos", "opacity_raw", "f_dc", "f_rest", "scale_raw", "q_raw")


def shard_views(n_views, rank, world_size):
    """Round-robin assignment of camera views to ranks (independent units, no exchange)."""
    return list(range(rank, n_views, world_size))


def _big_and_small(grads):
    """Split a list of gradient tensors into the largest (sent in place) and the rest (flattened together)."""
    order = sorted(range(len(grads)), key=lambda i: grads[i].numel(), reverse=True)
    return order[0], order[1:]


def _common_base(grads):
    """If all gradient tensors are disjoint contiguous views into ONE storage that they (nearly) cover -- what the render
    backward produces (ops._flat_like) -- return a flat tensor spanning them, else None.  Detected through the storage,
    because autograd detaches the tensors it puts into .grad."""
    g0 = grads[0]
    try:
        sp = g0.untyped_storage().data_ptr()
        if any((not g.is_contiguous()) or g.dtype != g0.dtype or g.device != g0.device or
               g.untyped_storage().data_ptr() != sp for g in grads):
            return None
        spans = sorted((g.storage_offset(), g.storage_offset() + g.numel()) for g in grads)
    except (RuntimeError, AttributeError):
        return None
    if any(a_end > b_start for (_, a_end), (b_start, _) in zip(spans, spans[1:])):
        return None                                           # overlapping views
    lo, hi = spans[0][0], spans[-1][1]
    if hi - lo > sum(g.numel() for g in grads) + 64 * len(grads):
        return None                                           # too much foreign data in between
    return torch.empty(0, dtype=g0.dtype, device=g0.device).set_(g0.untyped_storage(), lo, (hi - lo,))


def allreduce_gradients(grads, world_views, group=None):
    """Sum gradient tensors over ranks and scale by 1 / world_views (== loss / batch_size in the reference).

    grads: list of tensors (same shapes on every rank), modified in place.  Returns the list.  When the tensors are the
    views of one flat buffer produced by the render backward, a single in-place all-reduce of that buffer is issued.
    """
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        if world_views != 1:
            for g in grads:
                g.mul_(1.0 / world_views)
        return grads
    base = _common_base(grads)
    if base is not None:
        dist.all_reduce(base, op=dist.ReduceOp.SUM, group=group)
        if world_views != 1:
            base.mul_(1.0 / world_views)
        return grads
    big, small = _big_and_small(grads)
    works = [dist.all_reduce(grads[big], op=dist.ReduceOp.SUM, group=group, async_op=True)]
    flat = None
    if small:
        flat = torch.cat([grads[i].reshape(-1) for i in small])
        works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()
    scale = 1.0 / world_views
    grads[big].mul_(scale)
    if flat is not None:
        flat.mul_(scale)
        off = 0
        for i in small:
            n = grads[i].numel()
            grads[i].copy_(flat[off:off + n].view_as(grads[i]))
            off += n
    return grads


def data_parallel_step(render_fn, params, views, targets_grad_fn, world_views, group=None):
    """One DP step on this rank: render this rank's views, back-propagate, all-reduce the parameter gradients.

    render_fn(params, view) -> image; targets_grad_fn(image, view) -> scalar loss.  Gradients end up in p.grad of
    every tensor in `params` (dict name -> leaf tensor), identical on every rank after the call.
    """
    for p in params.values():
        p.grad = None
    total = None
    for v in views:
        loss = targets_grad_fn(render_fn(params, v), v)
        total = loss if total is None else total + loss
    if total is not None:
        total.backward()
    grads = []
    for p in params.values():
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        grads.append(p.grad)
    allreduce_gradients(grads, world_views, group)
    return total
