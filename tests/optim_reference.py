"""CPU restatement (plain torch) of the reference's sparse optimizer kernels -- test infrastructure only.

Follows optim/fractional_adam.py:20-85, optim/fractional_laprop.py (same kernels, LaProp ordering),
optim/fractional.py:107-147 (weighted_step), :36-63 (FractionalOpt.step) and optim/visibility_aware.py:24-108.
Parity for this row is pinned by a known answer rather than reference outputs (the reference kernels need
taichi): with weight == 1 on every row the scalar Adam step is torch.optim.Adam's with lr * (1 - exp(-2)),
checked in tests/test_optim_cpu.py.
"""
import math

import torch


def lerp(t, a, b):
    return a * t + b * (1.0 - t)


def moment_step(kind, vector, idx, w, m, v, total_weight, grad, lr, betas, eps, bias_correction=True):
    """Updates m, v rows in place; returns lr_step (rows, dims).  kind: 'adam' | 'laprop'."""
    b1, b2 = betas
    tw = total_weight[idx]
    b1w, b2w = (b1 ** w).unsqueeze(1), (b2 ** w).unsqueeze(1)
    g = grad[idx]
    bias1 = (1 - b1 ** tw).unsqueeze(1) if bias_correction else torch.ones_like(b1w)
    bias2 = (1 - b2 ** tw).unsqueeze(1) if bias_correction else torch.ones_like(b1w)
    sq = (g * g).sum(1, keepdim=True) if vector else g * g
    v_old = v[idx].unsqueeze(1) if vector else v[idx]
    v_new = lerp(b2w, v_old, sq)
    if kind == 'adam':
        m_new = lerp(b1w, m[idx], g)
        step = m_new / torch.clamp_min(torch.sqrt(v_new), eps) * (torch.sqrt(bias2) / bias1) * lr
    else:
        m_new = lerp(b1w, m[idx], g / torch.clamp_min(torch.sqrt(v_new / bias2), eps))
        step = m_new * lr / bias1
    m[idx] = m_new
    v[idx] = v_new.squeeze(1) if vector else v_new
    return step


def saturate(x):
    return 1 - 1 / torch.exp(2 * x)


class RefOptimizer:
    """State-holding restatement of FractionalOpt / VisibilityOptimizer over a dict of named groups.
    groups: list of dicts {name, param (N, ...), type, lr, mask_lr, point_lr}."""

    def __init__(self, kind, groups, betas=(0.9, 0.999), eps=1e-16, bias_correction=True, visibility=False,
                 vis_beta=0.5, vis_smooth=0.01, grad_scale=1.0):
        self.kind, self.groups, self.betas, self.eps, self.bias = kind, groups, betas, eps, bias_correction
        self.visibility, self.vis_beta, self.vis_smooth, self.grad_scale = visibility, vis_beta, vis_smooth, grad_scale
        n = groups[0]['param'].shape[0]
        self.total_weight = torch.zeros(n, dtype=groups[0]['param'].dtype)
        self.running_vis = torch.zeros(n, dtype=groups[0]['param'].dtype)
        for g in groups:
            p = g['param'].view(n, -1)
            g['m'] = torch.zeros_like(p)
            g['v'] = torch.zeros(n, dtype=p.dtype) if g['type'] != 'scalar' else torch.zeros_like(p)

    def step(self, grads, indexes, weight, basis=None):
        if self.visibility:
            vis = weight
            t = self.vis_beta
            a, b = vis ** 4, self.running_vis[indexes] ** 4
            updated = (a + (b - a) * t) ** 0.25
            self.running_vis[indexes] = updated
            weight = vis / torch.clamp_min(updated, 1e-12)
        self.total_weight[indexes] += weight
        for g in self.groups:
            n = g['param'].shape[0]
            param = g['param'].view(n, -1)
            grad = grads[g['name']].reshape(n, -1).clone()
            if self.visibility:
                scaled = grad[indexes] * self.grad_scale / (vis.unsqueeze(1) + self.vis_smooth)
                grad = torch.zeros_like(grad)
                grad[indexes] = scaled
            if g['type'] == 'local_vector':
                grad[indexes] = torch.einsum('bij,bj->bi', torch.linalg.inv(basis), grad[indexes])
            step = moment_step(self.kind, g['type'] != 'scalar', indexes, weight, g['m'], g['v'], self.total_weight,
                               grad, g['lr'], self.betas, self.eps, self.bias)
            if g['type'] == 'local_vector':
                step = torch.einsum('bij,bj->bi', basis, step)
            if g.get('mask_lr') is not None:
                step = step * g['mask_lr'].view(-1).unsqueeze(0)
            if g.get('point_lr') is not None:
                step = step * g['point_lr'][indexes].unsqueeze(1)
            param[indexes] -= step * saturate(weight).unsqueeze(1)
