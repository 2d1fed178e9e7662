"""Kernel factories with the call signature of the reference's per-algorithm modules (optim/fractional_adam.py:7-85,
optim/fractional_laprop.py:7-85): `scalar_kernel(betas, eps, bias_correction)` / `vector_kernel(betas, eps, dims,
bias_correction)` return a callable

    kernel(lr_step, indexes, weight, m, v, total_weight, grad, lr)

that updates the moments of the rows in `indexes` in place and writes their step, multiplied by lr but not yet by the
row's saturation 1 - exp(-2 w) (the caller's, reference optim/fractional.py:57-63), into `lr_step` (len(indexes), D).  Here the callable is one launch of gs_optim_step;
the optimizers of fractional.py do not go through it (they let the kernel apply the update to the parameter as well),
it exists for code written against those modules.
"""
from __future__ import annotations

from functools import lru_cache

import torch

from .. import _native as nv


def _factory(algorithm: int, per_row_moment: bool, betas, eps: float, dims, bias_correction: bool):
    beta1, beta2 = (float(b) for b in betas)

    def kernel(lr_step: torch.Tensor, indexes: torch.Tensor, weight: torch.Tensor, m: torch.Tensor, v: torch.Tensor,
               total_weight: torch.Tensor, grad: torch.Tensor, lr: float) -> None:
        count, width = lr_step.shape
        assert dims is None or width == dims, f"kernel built for {dims} columns, lr_step has {width}"
        assert indexes.shape[0] == count == weight.shape[0], "one index and one weight per row of lr_step"
        assert m.shape == grad.shape and m.shape[1] == width, f"moments {tuple(m.shape)} vs grad {tuple(grad.shape)}"
        assert v.shape == ((m.shape[0],) if per_row_moment else m.shape), f"second moment has shape {tuple(v.shape)}"
        for t in (lr_step, indexes, weight, m, v, total_weight, grad):
            assert t.is_contiguous(), "optimizer kernels take contiguous tensors"
        nv.require_device(lr_step, weight, m, v, total_weight, grad, what="optimizer kernel")
        nv.require_device(indexes, dtype=torch.int64, what="optimizer kernel indexes")
        nv.check(nv.lib().gs_optim_step(algorithm, int(per_row_moment), count, width, nv.ptr(indexes), nv.ptr(weight),
                                        nv.ptr(m), nv.ptr(v), nv.ptr(total_weight), nv.ptr(grad), float(lr), beta1,
                                        beta2, float(eps), int(bool(bias_correction)), nv.ptr(lr_step), None, None,
                                        None, None, nv.stream()), "gs_optim_step")

    return kernel


@lru_cache(maxsize=None)
def make(algorithm: int, per_row_moment: bool, betas=(0.9, 0.999), eps: float = 1e-16, dims=None,
         bias_correction: bool = True):
    return _factory(algorithm, per_row_moment, tuple(betas), eps, dims, bias_correction)
