"""Pins the optimizer restatement (tests/optim_reference.py) to a known answer, and checks the product
optimizers refuse to run without a HIP device."""
import math

import pytest
import torch

from optim_reference import RefOptimizer


def test_full_weight_scalar_adam_is_torch_adam():
    torch.manual_seed(0)
    n, d, lr = 50, 7, 0.01
    p0 = torch.randn(n, d, dtype=torch.float64)
    ours = p0.clone()
    theirs = torch.nn.Parameter(p0.clone())
    scale = 1 - math.exp(-2.0)       # saturate(1), optim/fractional.py:31-32
    adam = torch.optim.Adam([theirs], lr=lr * scale, betas=(0.9, 0.999), eps=1e-16)
    ref = RefOptimizer('adam', [dict(name='p', param=ours, type='scalar', lr=lr)])
    idx = torch.arange(n)
    for step in range(20):
        g = torch.randn(n, d, dtype=torch.float64)
        theirs.grad = g.clone()
        adam.step()
        ref.step({'p': g}, idx, torch.ones(n, dtype=torch.float64))
    assert torch.allclose(ours, theirs.detach(), rtol=1e-9, atol=1e-12)


def test_sparse_rows_untouched_and_fraction_monotone():
    torch.manual_seed(1)
    n = 30
    p = torch.randn(n, 3, dtype=torch.float64)
    before = p.clone()
    ref = RefOptimizer('laprop', [dict(name='p', param=p, type='vector', lr=0.1)])
    idx = torch.tensor([3, 5, 11])
    w = torch.tensor([0.1, 0.5, 1.0], dtype=torch.float64)
    g = torch.ones(n, 3, dtype=torch.float64)
    ref.step({'p': g}, idx, w)
    moved = (p - before).abs().sum(1)
    untouched = torch.ones(n, dtype=torch.bool)
    untouched[idx] = False
    assert (moved[untouched] == 0).all()
    assert moved[3] < moved[5] < moved[11]           # larger fractional weight, larger step


def test_product_optimizer_requires_device():
    from taichi_gaussian_rasterizer_amd.optim import SparseAdam
    p = torch.nn.Parameter(torch.randn(8, 3))
    p.grad = torch.randn(8, 3)
    opt = SparseAdam([dict(params=[p], name='p', type='scalar')], lr=0.1)
    with pytest.raises((ValueError, RuntimeError)):
        opt.step(torch.arange(8))
