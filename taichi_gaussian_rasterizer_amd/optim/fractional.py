"""FractionalOpt and its Adam / LaProp variants (reference optim/fractional.py:17-222, optim/util.py)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from .. import _native as nv

ADAM, LAPROP = 0, 1


@dataclass
class Group:
    name: str
    type: str
    param: torch.Tensor
    grad: Optional[torch.Tensor]
    state: dict
    lr: float
    betas: Tuple[float, float]
    eps: float
    bias_correction: bool
    mask_lr: Optional[torch.Tensor]
    point_lr: Optional[torch.Tensor]

    @property
    def num_points(self):
        return self.param.shape[0]


def make_group(group, state) -> Group:
    n = len(group["params"])
    assert n == 1, f"expected 1 tensor in group {group['name']}, got {n}"
    params = group["params"][0]
    return Group(name=group["name"], type=group["type"], param=params.view(params.shape[0], -1),
                 grad=params.grad.view(params.shape[0], -1) if params.grad is not None else None, state=state[params],
                 lr=group["lr"], betas=group["betas"], eps=group["eps"], bias_correction=group["bias_correction"],
                 mask_lr=group["mask_lr"], point_lr=group["point_lr"])


def get_vector_state(state: dict, param: torch.Tensor):
    # reference optim/util.py:5-10 (first moment per element, second moment per row)
    if 'v' not in state:
        state['v'] = torch.zeros_like(param.view(param.shape[0], -1))
        state['m'] = torch.zeros((param.shape[0],), dtype=param.dtype, device=param.device)
    return state['v'], state['m']


def get_scalar_state(state: dict, param: torch.Tensor):
    if 'v' not in state:
        state['v'] = torch.zeros_like(param.view(param.shape[0], -1))
        state['m'] = torch.zeros_like(param.view(param.shape[0], -1))
    return state['v'], state['m']


def get_total_weight(state: dict, n: int, device: torch.device):
    if 'total_weight' not in state:
        state['total_weight'] = torch.zeros(n, device=device, dtype=torch.float32)
    return state['total_weight']


def weighted_step(group: Group, visible_weight: torch.Tensor, visible_indexes: torch.Tensor,
                  total_weight: torch.Tensor, kind: int, basis: Optional[torch.Tensor] = None,
                  row_scale: Optional[torch.Tensor] = None, apply: bool = False):
    """reference optim/fractional.py:107-147.  With apply=True the kernel also performs the caller's
    `param[indexes] -= lr_step * saturate(weight)` (with mask_lr / point_lr) for scalar and vector groups and
    None is returned; local_vector groups need the basis round trip and return lr_step as the reference does.
    row_scale (one factor per visible row) multiplies the gradient inside the kernel."""
    if group.type in ["vector", "local_vector"]:
        m, v = get_vector_state(group.state, group.param)
        vector = 1
    elif group.type == "scalar":
        m, v = get_scalar_state(group.state, group.param)
        vector = 0
    else:
        raise ValueError(f"unknown group type {group.type}")

    if group.type == "local_vector":
        assert basis is not None, "basis is required for local_vector optimizer"
        inv_basis = torch.linalg.inv(basis)
        group.grad[visible_indexes] = torch.einsum('bij,bj->bi', inv_basis, group.grad[visible_indexes])

    grad = group.grad.contiguous()
    nv.require_device(grad, visible_weight, m, v, total_weight, row_scale, what="optimizer step")
    nv.require_device(visible_indexes, dtype=torch.int64, what="optimizer step indexes")
    idx, w = visible_indexes.contiguous(), visible_weight.contiguous()
    rows, dims = idx.shape[0], group.param.shape[1]
    fused = apply and group.type != "local_vector" and group.param.is_contiguous()
    mask_lr = point_lr = None
    if fused:
        if group.mask_lr is not None:
            mask_lr = group.mask_lr.reshape(-1).to(dtype=torch.float32).contiguous()
            assert mask_lr.shape[0] == dims, f"mask_lr has {mask_lr.shape[0]} entries for {dims} columns"
        if group.point_lr is not None:
            point_lr = group.point_lr.to(dtype=torch.float32).contiguous()
        nv.require_device(group.param, mask_lr, point_lr, what="optimizer step")
    lr_step = None if fused else group.param.new_zeros(rows, dims)
    scale = None if row_scale is None else row_scale.contiguous()
    nv.check(nv.lib().gs_optim_step(kind, vector, rows, dims, nv.ptr(idx), nv.ptr(w), nv.ptr(m), nv.ptr(v),
                                    nv.ptr(total_weight), nv.ptr(grad), float(group.lr), float(group.betas[0]),
                                    float(group.betas[1]), float(group.eps), int(group.bias_correction),
                                    nv.ptr(lr_step), nv.ptr(scale), nv.ptr(group.param) if fused else None,
                                    nv.ptr(mask_lr), nv.ptr(point_lr), nv.stream()), "gs_optim_step")
    if fused:
        return None

    if group.type == "local_vector":
        lr_step = torch.einsum('bij,bj->bi', basis, lr_step)
    if group.mask_lr is not None:
        lr_step *= group.mask_lr.view(-1).unsqueeze(0)
    if group.point_lr is not None:  # per row learning rate
        lr_step *= group.point_lr[visible_indexes].unsqueeze(1)
    return lr_step


def saturate(x: torch.Tensor):
    return 1 - 1 / torch.exp(2 * x)


class FractionalOpt(torch.optim.Optimizer):
    def __init__(self, kind: int, param_groups: list, lr=0.001, betas=(0.9, 0.999), eps=1e-16,
                 bias_correction=True):
        assert lr > 0, f"Invalid learning rate: {lr}"
        assert eps > 0, f"Invalid epsilon: {eps}"
        assert 0.0 <= betas[0] < 1.0, f"Invalid beta1: {betas[0]}"
        assert 0.0 <= betas[1] < 1.0, f"Invalid beta2: {betas[1]}"
        defaults = dict(lr=lr, betas=betas, eps=eps, mask_lr=None, point_lr=None, type="scalar",
                        bias_correction=bias_correction)
        self.kind = kind
        super().__init__(param_groups, defaults)

    @torch.no_grad()
    def step(self, indexes: torch.Tensor, weight: torch.Tensor, basis: Optional[torch.Tensor] = None):
        assert weight.shape == indexes.shape, f"shape mismatch {weight.shape} != {indexes.shape}"
        groups = [make_group(group, self.state) for group in self.param_groups]
        n = groups[0].param.shape[0]
        total_weight = get_total_weight(groups[0].state, n, device=weight.device)
        total_weight[indexes] += weight
        for group in groups:
            if group.grad is None:
                continue
            assert group.num_points == n, f"param shape {group.num_points} != {n}"
            lr_step = weighted_step(group, weight, indexes, total_weight, self.kind, basis, apply=True)
            if lr_step is not None:
                group.param[indexes] -= lr_step * saturate(weight).unsqueeze(1)


class FractionalAdam(FractionalOpt):
    def __init__(self, params, lr=0.001, betas=(0.9, 0.999), eps=1e-16, bias_correction=True):
        super().__init__(ADAM, params, lr, betas, eps, bias_correction)


class FractionalLaProp(FractionalOpt):
    def __init__(self, params, lr=0.001, betas=(0.9, 0.999), eps=1e-16, bias_correction=True):
        super().__init__(LAPROP, params, lr, betas, eps, bias_correction)


class SparseAdam(FractionalOpt):
    def __init__(self, params, lr=0.001, betas=(0.9, 0.999), eps=1e-16, bias_correction=True):
        super().__init__(ADAM, params, lr, betas, eps, bias_correction)

    def step(self, indexes: torch.Tensor, basis: Optional[torch.Tensor] = None):
        weight = torch.ones(indexes.shape[0], device=indexes.device, dtype=torch.float32)
        super().step(indexes, weight, basis)


class SparseLaProp(FractionalOpt):
    def __init__(self, params, lr=0.001, betas=(0.9, 0.999), eps=1e-16, bias_correction=True):
        super().__init__(LAPROP, params, lr, betas, eps, bias_correction)

    def step(self, indexes: torch.Tensor, basis: Optional[torch.Tensor] = None):
        weight = torch.ones(indexes.shape[0], device=indexes.device, dtype=torch.float32)
        super().step(indexes, weight, basis)
