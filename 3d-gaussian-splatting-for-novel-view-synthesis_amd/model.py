"""Gaussian parameter container with the reference's adaptive density control ("next" row 2).

    GaussianModel(initial_params, device)                     scripts/train.py:48-86
      .densify_and_prune(grads, opacity_threshold=0.01, max_grad=0.01, scale_threshold=0.01, max_screen_size=20)
                                                              scripts/train.py:89-141
      ._prune_points / ._split_points / ._clone_points        scripts/train.py:143-195
      .reset_opacity(threshold=0.01, bump=0.01)               scripts/train.py:564-569 (inline in the loop there)
      .save_checkpoint / .load_checkpoint                     scripts/train.py:197-219 (-> harness.py)

Semantics kept (checked against the reference's own methods, tests/golden/densify.npz):
  * prune first (`sigmoid(opacity_raw) < opacity_threshold`), gradients are re-indexed with the same mask;
  * split keeps the parent and appends ONE child per selected Gaussian: position + randn * exp(scale_raw) * 0.1, scale_raw - 0.5,
    everything else copied; clone appends an exact copy; children go to the end, in mask order; split children before clones;
  * `max_screen_size` is accepted and unused, as in the reference;
  * the optimiser state is not carried over (the reference builds a fresh Adam after every densification, :554-561).

One documented divergence: in the reference the clone mask is computed before the split and applied after it
(:136-141), so when a split and a clone are both due it indexes [N+S]-row tensors with an [N]-row mask and raises
IndexError.  Here the clone mask addresses the N pre-split rows it was computed for (what the reference evidently means,
and what it does whenever it does not raise); the fixture records the reference's exception for that case.

Host-side tensor bookkeeping (boolean masks + concatenation, run every `densification_interval` iterations): plain
torch ops on whatever device the parameters live on; no kernel of its own.  `generator` makes the split noise
reproducible and identical on every data-parallel rank.
"""
import torch

from . import harness

PARAM_KEYS = harness.PARAM_KEYS


class GaussianModel:
    def __init__(self, initial_params, device='cuda'):
        self.device = device
        for k in PARAM_KEYS:
            setattr(self, k, torch.nn.Parameter(initial_params[k].to(device)))

    def get_params(self):
        return {k: getattr(self, k) for k in PARAM_KEYS}

    def get_num_gaussians(self):
        return self.pos.shape[0]

    # ---- adaptive density control -------------------------------------------------------------------
    def densify_and_prune(self, grads, opacity_threshold=0.01, max_grad=0.01, scale_threshold=0.01, max_screen_size=20,
                          generator=None):
        opacity = torch.sigmoid(self.opacity_raw)
        prune_mask = opacity < opacity_threshold
        self._prune_points(prune_mask)
        if grads is not None:
            for key in grads:
                if grads[key] is not None:
                    grads[key] = grads[key][~prune_mask]
        if grads is not None and grads.get('pos') is not None:
            grad_norm = grads['pos'].norm(dim=-1)
            max_scale = torch.exp(self.scale_raw).max(dim=-1)[0]
            hot = grad_norm > max_grad
            split_mask = (max_scale > scale_threshold) & hot
            clone_mask = (max_scale <= scale_threshold) & hot
            n_before = self.pos.shape[0]
            self._split_points(split_mask, generator=generator)
            grown = self.pos.shape[0] - n_before
            if grown:       # the clone mask belongs to the pre-split rows (see the module docstring)
                clone_mask = torch.cat([clone_mask, clone_mask.new_zeros(grown)])
            self._clone_points(clone_mask)

    def _replace(self, new):
        for k in PARAM_KEYS:
            setattr(self, k, torch.nn.Parameter(new[k]))

    def _prune_points(self, mask):
        if not mask.any():
            return
        keep = ~mask
        self._replace({k: getattr(self, k)[keep] for k in PARAM_KEYS})

    def _split_points(self, mask, generator=None):
        if not mask.any():
            return
        sel = {k: getattr(self, k)[mask].clone() for k in PARAM_KEYS}
        if generator is None:
            noise = torch.randn_like(sel['pos'])
        else:
            noise = torch.randn(sel['pos'].shape, generator=generator, device=generator.device,
                                dtype=sel['pos'].dtype).to(sel['pos'].device)
        sel['pos'] = sel['pos'] + noise * torch.exp(self.scale_raw[mask]) * 0.1
        sel['scale_raw'] = sel['scale_raw'] - 0.5
        self._replace({k: torch.cat([getattr(self, k), sel[k]], dim=0) for k in PARAM_KEYS})

    def _clone_points(self, mask):
        if not mask.any():
            return
        self._replace({k: torch.cat([getattr(self, k), getattr(self, k)[mask]], dim=0) for k in PARAM_KEYS})

    @torch.no_grad()
    def reset_opacity(self, threshold=0.01, bump=0.01):
        """Opacity reset of the training loop (scripts/train.py:564-569): Gaussians below `threshold` get
        logit(clamp(opacity + bump, 0, 1)).  Returns the number of Gaussians touched (a host read, as `mask.any()` is there)."""
        opacity = torch.sigmoid(self.opacity_raw)
        mask = opacity < threshold
        n = int(mask.sum())
        if n:
            self.opacity_raw.data[mask] = torch.logit(torch.clamp(opacity[mask] + bump, 0, 1))
        return n

    # ---- checkpoints --------------------------------------------------------------------------------
    def save_checkpoint(self, path, iteration):
        harness.save_checkpoint(path, self, iteration)

    def load_checkpoint(self, path):
        params, iteration = harness.load_checkpoint(path, device=self.device)
        self._replace({k: params[k] for k in PARAM_KEYS})
        return iteration
