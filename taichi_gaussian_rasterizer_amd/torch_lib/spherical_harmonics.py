"""View-dependent colour from SH coefficients in plain torch (reference torch_lib/spherical_harmonics.py:32-44):
colour[c] = clamp(sum_d Y_d(direction) * coefficients[c, d] + 0.5, 0, 1), direction from the camera to the point.
A differentiable stand-alone utility; `taichi_gaussian_rasterizer_amd.spherical_harmonics.evaluate_sh_at` is the HIP operator."""
from __future__ import annotations

import math

import torch

from .rsh import rsh_cart


def sh_degree(sh_params: torch.Tensor) -> int:
    degree = math.isqrt(sh_params.shape[-1]) - 1
    assert (degree + 1) ** 2 == sh_params.shape[-1], f"last dimension {sh_params.shape[-1]} is not a square"
    return degree


def evaluate_sh(sh_params: torch.Tensor, directions: torch.Tensor) -> torch.Tensor:
    """coefficients (N, C, D), unit directions (N, 3) -> (N, C), before the +0.5 offset and the clamp"""
    basis = rsh_cart(directions, sh_degree(sh_params))
    return torch.einsum("ncd,nd->nc", sh_params, basis)


def evaluate_sh_at(sh_params: torch.Tensor, points: torch.Tensor, indexes: torch.Tensor,
                   camera_pos: torch.Tensor) -> torch.Tensor:
    toward = torch.nn.functional.normalize(points[indexes] - camera_pos.reshape(1, 3), dim=1)
    return (evaluate_sh(sh_params[indexes], toward) + 0.5).clamp(0.0, 1.0)
