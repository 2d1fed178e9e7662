"""Row-sparse optimizers driven by the renderer's visible set.

A training view touches only the Gaussians in `points_in_view`; these optimizers update exactly those rows, and
let a row take a *fraction* of a step (weight w in place of 1: moments decay by beta**w, the update is scaled by
1 - exp(-2 w)).  Semantics follow the reference (optim/fractional.py:17-222, optim/fractional_adam.py,
optim/fractional_laprop.py, optim/util.py): parameter groups hold ONE (N, ...) tensor each and are typed

    "scalar"        second moment per element
    "vector"        one second moment per row (norm of the row's gradient)
    "local_vector"  as "vector", in a per-row frame: `basis` (rows, D, D) maps local -> parameter coordinates

with optional `mask_lr` (per column) and `point_lr` (per row) multipliers.  The moment update -- and for the
first two group types the parameter update itself -- is one launch of gs_optim_step per group.
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import _native as nv

ADAM, LAPROP = 0, 1
_GROUP_TYPES = ("scalar", "vector", "local_vector")


def saturate(x: torch.Tensor) -> torch.Tensor:
    """fraction of a full step a row with weight x takes: 1 - exp(-2 x)"""
    return -torch.expm1(-2.0 * x)


class _Rows:
    """One parameter group seen as an (N, D) matrix, with its optimizer state."""

    def __init__(self, group: dict, state: dict):
        tensors = group["params"]
        assert len(tensors) == 1, f"expected 1 tensor in group {group['name']}, got {len(tensors)}"
        (tensor,) = tensors
        self.tensor, self.state, self.options = tensor, state[tensor], group
        self.name, self.kind = group["name"], group["type"]
        if self.kind not in _GROUP_TYPES:
            raise ValueError(f"unknown group type {self.kind}")
        self.num_points = tensor.shape[0]
        self.param = tensor.view(self.num_points, -1)
        self.grad = None if tensor.grad is None else tensor.grad.view(self.num_points, -1)

    @property
    def per_row_moment(self) -> bool:
        return self.kind != "scalar"

    def moments(self):
        """(first moment (N, D), second moment (N, D) or (N)), created on first use.  Keys as in the reference's
        checkpoints (optim/util.py:5-18): first moment under 'v', second moment under 'm'."""
        st = self.state
        if "v" not in st:
            st["v"] = torch.zeros_like(self.param)
            st["m"] = self.param.new_zeros(self.num_points) if self.per_row_moment else torch.zeros_like(self.param)
        # a loaded state dict may hold non-contiguous moments: make the STATE contiguous once (the kernel updates the
        # buffers it is handed in place; a temporary copy would be updated and dropped, freezing the moments)
        for key in ("v", "m"):
            if not st[key].is_contiguous():
                st[key] = st[key].contiguous()
        return st["v"], st["m"]

    def shared(self, key: str) -> torch.Tensor:
        """an (N) float32 counter kept in this group's state (the first group carries the optimizer-wide ones)"""
        if key not in self.state:
            self.state[key] = torch.zeros(self.num_points, dtype=torch.float32, device=self.param.device)
        return self.state[key]


def _launch(rows: _Rows, algorithm: int, indexes, weight, total_weight, grad, row_scale, in_place: bool):
    """gs_optim_step for one group; returns lr_step (rows, D) unless the update was applied in place"""
    m, v = rows.moments()
    opt = rows.options
    grad, indexes, weight = grad.contiguous(), indexes.contiguous(), weight.contiguous()
    scale = None if row_scale is None else row_scale.contiguous()
    # a state dict from elsewhere may hold anything under these keys: the kernel indexes m[idx * D + j] and
    # v[idx] / v[idx * D + j] without further checks
    second = (rows.num_points,) if rows.per_row_moment else tuple(rows.param.shape)
    if tuple(m.shape) != tuple(rows.param.shape):
        raise ValueError(f"{rows.name}: first moment (state['v']) has shape {tuple(m.shape)}, expected "
                         f"{tuple(rows.param.shape)}")
    if tuple(v.shape) != second:
        raise ValueError(f"{rows.name}: second moment (state['m']) has shape {tuple(v.shape)}, expected {second}")
    nv.require_device(grad, weight, m, v, total_weight, scale, what="optimizer step")
    nv.require_device(indexes, dtype=torch.int64, what="optimizer step indexes")
    count, width = indexes.shape[0], rows.param.shape[1]
    mask = per_point = None
    if in_place:
        if opt["mask_lr"] is not None:
            mask = opt["mask_lr"].reshape(-1).to(torch.float32).contiguous()
            assert mask.shape[0] == width, f"mask_lr has {mask.shape[0]} entries for {width} columns"
        if opt["point_lr"] is not None:
            per_point = opt["point_lr"].to(torch.float32).contiguous()
        nv.require_device(rows.param, mask, per_point, what="optimizer step")
    lr_step = None if in_place else rows.param.new_zeros(count, width)
    beta1, beta2 = opt["betas"]
    nv.check(nv.lib().gs_optim_step(algorithm, int(rows.per_row_moment), count, width, nv.ptr(indexes), nv.ptr(weight),
                                    nv.ptr(m), nv.ptr(v), nv.ptr(total_weight), nv.ptr(grad), float(opt["lr"]),
                                    float(beta1), float(beta2), float(opt["eps"]), int(opt["bias_correction"]),
                                    nv.ptr(lr_step), nv.ptr(scale), nv.ptr(rows.param) if in_place else None,
                                    nv.ptr(mask), nv.ptr(per_point), nv.stream()), "gs_optim_step")
    return lr_step


def update_rows(rows: _Rows, algorithm: int, indexes: torch.Tensor, weight: torch.Tensor, total_weight: torch.Tensor,
                basis: Optional[torch.Tensor] = None, row_scale: Optional[torch.Tensor] = None) -> None:
    """One fractional step of the visible rows of one group (reference optim/fractional.py:107-147 + :57-63)."""
    grad = rows.grad
    if rows.kind == "local_vector":
        assert basis is not None, "basis is required for local_vector optimizer"
        # gradient into the local frame, step back out of it; the kernel sees a scratch copy of the visible rows so
        # that the caller's .grad stays as autograd left it
        local = grad.clone()
        visible = grad[indexes] if row_scale is None else grad[indexes] * row_scale.unsqueeze(1)
        local[indexes] = torch.einsum("bij,bj->bi", torch.linalg.inv(basis), visible)
        step = _launch(rows, algorithm, indexes, weight, total_weight, local, None, in_place=False)
        step = torch.einsum("bij,bj->bi", basis, step)
        if rows.options["mask_lr"] is not None:
            step = step * rows.options["mask_lr"].reshape(1, -1)
        if rows.options["point_lr"] is not None:
            step = step * rows.options["point_lr"][indexes].unsqueeze(1)
        rows.param[indexes] -= step * saturate(weight).unsqueeze(1)
    elif rows.param.is_contiguous():
        _launch(rows, algorithm, indexes, weight, total_weight, grad, row_scale, in_place=True)
    else:  # a parameter that is a strided view: let torch do the scatter
        step = _launch(rows, algorithm, indexes, weight, total_weight, grad, row_scale, in_place=False)
        if rows.options["mask_lr"] is not None:
            step = step * rows.options["mask_lr"].reshape(1, -1)
        if rows.options["point_lr"] is not None:
            step = step * rows.options["point_lr"][indexes].unsqueeze(1)
        rows.param[indexes] -= step * saturate(weight).unsqueeze(1)


class FractionalOpt(torch.optim.Optimizer):
    """step(indexes, weight, basis=None): rows `indexes` take a step of fraction `weight` each"""
    algorithm = ADAM

    def __init__(self, param_groups: list, lr=0.001, betas=(0.9, 0.999), eps=1e-16, bias_correction=True, **extra):
        assert lr > 0, f"Invalid learning rate: {lr}"
        assert eps > 0, f"Invalid epsilon: {eps}"
        for i, beta in enumerate(betas, start=1):
            assert 0.0 <= beta < 1.0, f"Invalid beta{i}: {beta}"
        defaults = dict(lr=lr, betas=betas, eps=eps, mask_lr=None, point_lr=None, type="scalar",
                        bias_correction=bias_correction)
        defaults.update(extra)
        super().__init__(param_groups, defaults)

    def _rows(self):
        views = [_Rows(group, self.state) for group in self.param_groups]
        count = views[0].num_points
        for view in views:
            assert view.num_points == count, f"param shape {view.num_points} != {count}"
        return views

    @torch.no_grad()
    def _take_step(self, indexes, weight, basis=None, row_scale=None, counted: bool = False):
        """`counted`: total_weight already includes this step's weights"""
        assert weight.shape == indexes.shape, f"shape mismatch {weight.shape} != {indexes.shape}"
        views = self._rows()
        total_weight = views[0].shared("total_weight")
        if not counted:
            total_weight[indexes] += weight
        for view in views:
            if view.grad is not None:
                update_rows(view, self.algorithm, indexes, weight, total_weight, basis, row_scale)

    def step(self, indexes: torch.Tensor, weight: torch.Tensor, basis: Optional[torch.Tensor] = None):
        self._take_step(indexes, weight, basis)


class FractionalAdam(FractionalOpt):
    algorithm = ADAM


class FractionalLaProp(FractionalOpt):
    algorithm = LAPROP


class _WholeSteps(FractionalOpt):
    """step(indexes, basis=None): every listed row takes a full step (weight 1)"""

    def step(self, indexes: torch.Tensor, basis: Optional[torch.Tensor] = None):
        self._take_step(indexes, torch.ones(indexes.shape[0], device=indexes.device, dtype=torch.float32), basis)


class SparseAdam(_WholeSteps):
    algorithm = ADAM


class SparseLaProp(_WholeSteps):
    algorithm = LAPROP
