"""2D toy path of the reference (misc/renderer2d.py:17-33, :135-149): Gaussians2D -> packed
7-vector, then rasterize.  Used by the reference's rasterizer tests and by
examples/fit_image_gaussians.py:101-123 (BASELINE config 1)."""
from __future__ import annotations

from numbers import Integral
from typing import Tuple

import torch

from ..data_types import Gaussians2D, RasterConfig


def project_gaussians2d(points: Gaussians2D) -> torch.Tensor:
    """(N,7) packed [mean.xy, axis.xy, sigma.xy, alpha] (pure torch, differentiable)."""
    alpha = torch.sigmoid(points.alpha_logit)
    sigma = points.scaling
    v1 = points.rotation / torch.norm(points.rotation, dim=1, keepdim=True)
    return torch.cat([points.position, v1, sigma, alpha.reshape(-1, 1)], dim=-1)


def render_gaussians(gaussians: Gaussians2D, image_size: Tuple[Integral, Integral],
                     raster_config: RasterConfig = RasterConfig()):
    from ..rasterizer.function import rasterize
    gaussians2d = project_gaussians2d(gaussians)
    return rasterize(gaussians2d=gaussians2d, depth=torch.clamp(gaussians.z_depth, 0, 1),
                     features=gaussians.feature, image_size=image_size, config=raster_config)


def _unit_axes(points: Gaussians2D):
    major = torch.nn.functional.normalize(points.rotation, dim=1)
    minor = torch.stack([-major[:, 1], major[:, 0]], dim=-1)
    return major, minor


def point_basis(points: Gaussians2D, eps: float = 1e-4) -> torch.Tensor:
    """(N,2,2) local frame of every Gaussian: columns = unit axes scaled by sigma (reference
    misc/renderer2d.py:37-43); the `basis` of the local_vector optimizer groups."""
    major, minor = _unit_axes(points)
    sigma = points.scaling.clamp_min(eps)
    return torch.stack([major, minor], dim=2) * sigma.unsqueeze(-2)


def point_rotation(points: Gaussians2D) -> torch.Tensor:
    """(N,2,2) rows = unit major / minor axis (reference misc/renderer2d.py:47-52)."""
    major, minor = _unit_axes(points)
    return torch.stack([major, minor], dim=1)


def point_covariance(points: Gaussians2D) -> torch.Tensor:
    """(N,2,2) covariance B B^T (reference misc/renderer2d.py:54-56)."""
    basis = point_basis(points)
    return basis @ basis.transpose(1, 2)
