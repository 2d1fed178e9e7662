"""Optimizers that pace each Gaussian by how visible it was in the view (reference optim/visibility_aware.py:13-124).

`step(indexes, visibility, basis=None)`: `visibility` is the rasterizer's per-splat blend weight sum for the rows in
`indexes`.  A running visibility per Gaussian (a power mean with exponent 4 against the history, `vis_beta`) turns
it into the fractional step weight visibility / running, and the gradient of a row is rescaled by
grad_scale / (visibility + vis_smooth) so that barely visible splats are not starved.
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import _native as nv
from .fractional import ADAM, LAPROP, FractionalOpt


def power_mean_update(history: torch.Tensor, value: torch.Tensor, keep: float, exponent: float = 4.0) -> torch.Tensor:
    """((1 - keep) value^p + keep history^p)^(1/p): with p = 4 a recent high visibility dominates the mean"""
    vp, hp = value ** exponent, history ** exponent
    return (vp + (hp - vp) * keep) ** (1.0 / exponent)


class VisibilityOptimizer(FractionalOpt):
    def __init__(self, param_groups: list, lr=0.001, betas=(0.9, 0.999), eps=1e-16, vis_beta=0.9,
                 vis_smooth: float = 0.01, bias_correction=True, grad_scale: float = 1.0):
        assert 0.0 <= vis_beta < 1.0, f"Invalid visibility beta: {vis_beta}"
        super().__init__(param_groups, lr=lr, betas=betas, eps=eps, bias_correction=bias_correction)
        self.vis_beta, self.vis_smooth, self.grad_scale = vis_beta, vis_smooth, grad_scale

    @torch.no_grad()
    def step(self, indexes: torch.Tensor, visibility: torch.Tensor, basis: Optional[torch.Tensor] = None):
        assert visibility.shape == indexes.shape, f"shape mismatch {visibility.shape} != {indexes.shape}"
        first = self._rows()[0]
        running, total_weight = first.shared("running_vis"), first.shared("total_weight")
        indexes, visibility = indexes.contiguous(), visibility.to(torch.float32).contiguous()
        nv.require_device(visibility, running, total_weight, what="optimizer step")
        nv.require_device(indexes, dtype=torch.int64, what="optimizer step indexes")
        weight, row_scale = torch.empty_like(visibility), torch.empty_like(visibility)
        # one launch for the whole pacing arithmetic (power_mean_update above states what it computes)
        nv.check(nv.lib().gs_optim_visibility_weights(
            indexes.shape[0], nv.ptr(indexes), nv.ptr(visibility), nv.ptr(running), nv.ptr(total_weight),
            float(self.vis_beta), float(self.grad_scale), float(self.vis_smooth), nv.ptr(weight), nv.ptr(row_scale),
            nv.stream()), "gs_optim_visibility_weights")
        self._take_step(indexes, weight, basis, row_scale=row_scale, counted=True)


class VisibilityAwareAdam(VisibilityOptimizer):
    algorithm = ADAM

    def __init__(self, param_groups, lr=0.001, betas=(0.9, 0.999), eps=1e-16, vis_beta=0.5, vis_smooth: float = 0.01,
                 bias_correction=True):
        super().__init__(param_groups, lr=lr, betas=betas, eps=eps, vis_beta=vis_beta, vis_smooth=vis_smooth,
                         bias_correction=bias_correction)


class VisibilityAwareLaProp(VisibilityOptimizer):
    algorithm = LAPROP

    def __init__(self, param_groups, lr=0.001, betas=(0.9, 0.999), eps=1e-16, vis_beta=0.5, vis_smooth: float = 0.01,
                 bias_correction=True):
        super().__init__(param_groups, lr=lr, betas=betas, eps=eps, vis_beta=vis_beta, vis_smooth=vis_smooth,
                         bias_correction=bias_correction)
