"""Real spherical-harmonics basis in torch, degree 0..3 (16 functions), in the sign and ordering convention of the
renderer (reference torch_lib/rsh.py, spherical_harmonics.py:38-106).  Plain torch -- differentiable, any device, any
float dtype -- for checks and small experiments; the renderer itself evaluates SH in sh.hip.
"""
from __future__ import annotations

import torch

# normalisation constants by degree
_K0 = 0.282094791773878
_K1 = 0.48860251190292
_K2 = (1.09254843059208, 0.94617469575756, 0.31539156525252, 0.54627421529604)
_K3 = (0.590043589926644, 2.89061144264055, 0.304697199642977, 1.24392110863372, 0.497568443453487, 1.44530572132028)


def rsh_cart(xyz: torch.Tensor, degree: int) -> torch.Tensor:
    """unit directions (..., 3) -> basis values (..., (degree + 1)^2)"""
    assert 0 <= degree <= 3, f"SH degree must be between 0 and 3, got {degree}"
    x, y, z = xyz[..., 0], xyz[..., 1], xyz[..., 2]
    terms = [torch.full_like(x, _K0)]
    if degree >= 1:
        terms += [-_K1 * y, _K1 * z, -_K1 * x]
    if degree >= 2:
        a, b, c, d = _K2
        terms += [a * x * y, -a * y * z, b * z * z - c, -a * x * z, d * x * x - d * y * y]
    if degree >= 3:
        a, b, c, d, e, f = _K3
        x2, y2, z2 = x * x, y * y, z * z
        terms += [-a * y * (3.0 * x2 - y2), b * x * y * z, c * y * (1.5 - 7.5 * z2), d * z * (1.5 * z2 - 0.5) - e * z,
                  c * x * (1.5 - 7.5 * z2), f * z * (x2 - y2), -a * x * (x2 - 3.0 * y2)]
    return torch.stack(terms, dim=-1)


def rsh_cart_0(xyz): return rsh_cart(xyz, 0)
def rsh_cart_1(xyz): return rsh_cart(xyz, 1)
def rsh_cart_2(xyz): return rsh_cart(xyz, 2)
def rsh_cart_3(xyz): return rsh_cart(xyz, 3)
