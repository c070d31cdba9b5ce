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


def agree_on_views(n_local, group=None, views_per_rank=None, device=None):
    """(equal, n_global): do all ranks render the same number of views this step, and how many views is that in all?

    Every rank must reach the same answer, because it selects the sequence of collectives of the exchange (FactoredExchange
    gathers view by view when the counts are equal and once, with a size exchange, when they are not): a rank deciding
    from its own count alone can disagree with the others and the step hangs.  So the counts are either given by the
    caller (`views_per_rank`: one entry per rank, the same list on every rank) or exchanged here with one tiny all-gather."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return True, int(n_local)
    world = dist.get_world_size(group)
    if views_per_rank is not None:
        counts = [int(c) for c in views_per_rank]
        if len(counts) != world:
            raise ValueError(f"views_per_rank has {len(counts)} entries for {world} ranks")
        if counts[dist.get_rank(group)] != int(n_local):
            raise ValueError(f"views_per_rank says {counts[dist.get_rank(group)]} views for this rank, got {n_local}")
    else:
        on_gpu = dist.get_backend(group) != "gloo"
        mine = torch.tensor([int(n_local)], dtype=torch.int64, device=device if on_gpu and device is not None else "cpu")
        out = torch.empty(world, dtype=torch.int64, device=mine.device)
        dist.all_gather_into_tensor(out, mine, group=group)
        counts = out.tolist()
    return all(c == counts[0] for c in counts), int(sum(counts))


def any_rank(flag, group=None, device=None):
    """Logical OR of a host-side flag over the ranks (one tiny all-reduce; the caller has just synchronised anyway)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return int(bool(flag))
    on_gpu = dist.get_backend(group) != "gloo"
    t = torch.tensor([int(bool(flag))], dtype=torch.int32, device=device if on_gpu and device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


STATUS_OK, STATUS_REDO, STATUS_OFFSCREEN, STATUS_ERROR = 0, 1, 2, 3


def agree_status(code, group=None, device=None):
    """The worst status over the ranks (ONE tiny all-reduce, MAX): 0 = every rank's pass is good, 1 = some rank's pair buffers
    overflowed (all repeat the pass), 2 = some rank's view had survivors but nothing on screen (all raise the reference's
    Exception), 3 = some rank failed otherwise (all raise).  Every rank must call it once per pass, whatever happened to it
    locally: a rank that leaves the step by an exception before this point would leave the others waiting in a collective."""
    code = int(code)
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return code
    on_gpu = dist.get_backend(group) != "gloo"
    t = torch.tensor([code], dtype=torch.int32, device=device if on_gpu and device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


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


def _all_gather_cat(t, group=None, equal=True):
    """Concatenation over ranks along dim 0 (RCCL all-gather; gloo with GPU tensors is staged through the host, for
    rehearsals only).  equal=False: the ranks may hold different numbers of rows (one extra small collective for the
    sizes, rows padded to the longest)."""
    world = dist.get_world_size(group)
    if not equal:
        mine = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
        sizes = _all_gather_cat(mine, group).tolist()
        longest = max(sizes)
        if any(sz != longest for sz in sizes):
            pad = torch.zeros((longest,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            pad[:t.shape[0]] = t
            full = _all_gather_cat(pad, group).view((world, longest) + tuple(t.shape[1:]))
            return torch.cat([full[r, :sizes[r]] for r in range(world)], 0)
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        parts = [torch.empty(t.shape, dtype=t.dtype) for _ in range(world)]
        dist.all_gather(parts, t.cpu(), group=group)
        return torch.cat(parts, 0).to(t.device)
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    return out


class FactoredExchange:
    """Gradient exchange that moves 2.6x fewer bytes than the all-reduce of all six tensors (8 views): per view the
    SH-coefficient gradient (192 of the 236 bytes per Gaussian) is the outer product of the 3 colour-logit gradients with
    the 16 SH basis values of the view direction, which every rank can evaluate itself.  So the ranks all-reduce only the
    gradients of pos / opacity_raw / scale_raw / q_raw (44 B per Gaussian), all-gather the logit gradients of their views
    (12 B per Gaussian and view) and camera positions, and rebuild sum_v g_v (x) Y(d_v) locally (gsplat_sh_accumulate).
    Same fp32 arithmetic as the plain path up to the order of the sum over views.

        ex = FactoredExchange(params, world_views)        # params: dict name -> leaf tensor
        with ex:                                            # render backward hands logit gradients to `ex`
            for view in my_views: loss(render_gaussians(...)).backward()
        ex.finish()                                         # collectives + rebuild; every p.grad is final and identical on all ranks

    By default every rank must render the same number of views (equal_views=False lifts that for one extra small
    collective per step).  Works unchanged in a single process (no collective)."""

    SMALL = ("pos", "opacity_raw", "scale_raw", "q_raw")

    def __init__(self, params, world_views, group=None, accumulate=None, force_collectives=False, equal_views=True):
        self.params, self.world_views, self.group = params, world_views, group
        self.equal_views = equal_views
        self.logits, self.eyes, self._early = [], [], []
        self._early = []                         # (gathered logits, gathered eyes, pending collectives) per local view
        self._accumulate = accumulate
        self._force = force_collectives          # tests: issue the collectives even in a one-rank group
        self.n_added = 0                         # views handed in by the render backward (Trainer.step checks it against its views)

    def owns(self, inputs, src_ptrs=None):
        """Is this render differentiating the parameters this exchange was built for (same f_dc / f_rest storage)?
        src_ptrs: the addresses of the caller's own f_dc / f_rest tensors, BEFORE the host's dtype / layout conversion (a model
        whose SH tensors are not fp32-contiguous is converted by the render; comparing the converted copies would silently
        send its frames down the ordinary path, and the replicas would diverge)."""
        try:
            mine = (self.params["f_dc"].data_ptr(), self.params["f_rest"].data_ptr())
            if src_ptrs is not None:
                return tuple(src_ptrs) == mine
            return (inputs["f_dc"].data_ptr(), inputs["f_rest"].data_ptr()) == mine
        except (KeyError, AttributeError):
            return False

    def _distributed(self):
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self._force)

    def add(self, grad_logit, eye):
        """Called by the render backward with one view's logit gradients BEFORE it launches the projection backward: with
        equal view counts the all-gather of this view starts here (async) and overlaps that kernel."""
        eye = eye.detach().to(torch.float32).contiguous()
        self.n_added += 1
        if self._distributed() and self.equal_views and not (dist.get_backend(self.group) == "gloo" and grad_logit.is_cuda):
            world = dist.get_world_size(self.group)
            n = grad_logit.shape[0]             # outputs in the concatenated form (every backend takes it), viewed per rank below
            out = torch.empty((world * n, 3), dtype=grad_logit.dtype, device=grad_logit.device)
            eyes = torch.empty(world * 3, dtype=torch.float32, device=eye.device)
            works = [dist.all_gather_into_tensor(out, grad_logit.contiguous(), group=self.group, async_op=True),
                     dist.all_gather_into_tensor(eyes, eye.reshape(3), group=self.group, async_op=True)]
            self._early.append((out.view(world, n, 3), eyes.view(world, 3), works + [grad_logit, eye]))   # inputs kept alive
        else:
            self.logits.append(grad_logit)
            self.eyes.append(eye)

    def __enter__(self):
        from . import ops
        ops.set_sh_gradient_sink(self)
        return self

    def __exit__(self, *exc):
        from . import ops
        ops.set_sh_gradient_sink(None)
        return False

    def pad_views(self, expected):
        """A rank whose pass stopped early (an exception in one of its renders) has issued fewer per-view collectives than its
        peers: issue the missing ones on zeros, so that the NEXT collective of every rank -- the status agreement -- lines up.
        (Only the view-by-view gathers of add() are collectives; with unequal view counts nothing is exchanged before finish().)"""
        n = self.params["pos"].shape[0]
        dev = self.params["pos"].device
        while self.n_added < int(expected):
            self.add(torch.zeros((n, 3), dtype=torch.float32, device=dev), torch.zeros(3, dtype=torch.float32, device=dev))

    def abandon(self):
        """Drop what was collected (the pass is being repeated): collectives already in flight are waited for, nothing is kept."""
        for _, _, works in self._early:
            for w in works[:2]:
                w.wait()
        self.logits, self.eyes, self._early = [], [], []
        self.n_added = 0

    def finish(self):
        p = self.params
        small = []
        for k in self.SMALL:
            if p[k].grad is None:
                p[k].grad = torch.zeros_like(p[k])
            small.append(p[k].grad)
        distributed = self._distributed()
        work = base = None
        n = p["pos"].shape[0]
        if self.logits:
            logits, eyes = torch.stack(self.logits), torch.stack(self.eyes)
        else:
            logits = torch.zeros((0, n, 3), dtype=torch.float32, device=p["pos"].device)
            eyes = torch.zeros((0, 3), dtype=torch.float32, device=p["pos"].device)
        if distributed:
            # the four small gradients are views of one flat buffer when they come from the render backward: one async
            # all-reduce, overlapped with the all-gather of the logit gradients
            base = _common_base(small)
            if base is not None:
                work = dist.all_reduce(base, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            if self._early:                      # gathered view by view since add(): just wait
                for _, _, works in self._early:
                    for w in works[:2]:
                        w.wait()
                logits = torch.cat([o for o, _, _ in self._early] + ([logits] if logits.shape[0] else []), 0)
                eyes = torch.cat([e for _, e, _ in self._early] + ([eyes] if eyes.shape[0] else []), 0)
            else:
                logits = _all_gather_cat(logits, self.group, self.equal_views)
                eyes = _all_gather_cat(eyes, self.group, self.equal_views)
            if work is None:
                allreduce_gradients(small, self.world_views, self.group)
        elif self.world_views != 1:
            for g in small:
                g.mul_(1.0 / self.world_views)
        acc = self._accumulate
        if acc is None:
            from . import ops
            acc = ops.sh_accumulate
        # the SH rebuild only needs the gathered logit gradients: it runs while the all-reduce of the small gradients is
        # still in flight on the collective's stream
        g_dc, g_rest = acc(p["pos"].detach(), eyes, logits, 1.0 / self.world_views)
        for k, g in (("f_dc", g_dc), ("f_rest", g_rest)):
            g = g.to(p[k].dtype)
            p[k].grad = g if p[k].grad is None else p[k].grad + g
        if distributed and work is not None:
            work.wait()
            if self.world_views != 1:
                base.mul_(1.0 / self.world_views)
        self.logits, self.eyes, self._early = [], [], []
        self.n_added = 0
