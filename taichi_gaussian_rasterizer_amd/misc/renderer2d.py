"""2D toy path of the reference (misc/renderer2d.py:17-33, :135-149): Gaussians2D -> packed
7-vector, then rasterize.  Used by the reference's rasterizer tests and by
examples/fit_image_gaussians.py:101-123 (BASELINE config 1)."""
from __future__ import annotations

import math
from numbers import Integral
from typing import Optional, Tuple

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


# ---- densification helpers of the 2D example (reference misc/renderer2d.py:60-132; plain torch, no kernels)
def split_with_offsets(points: Gaussians2D, offsets: torch.Tensor, depth_noise: float = 1e-2) -> Gaussians2D:
    """Every Gaussian becomes `n` copies displaced by offsets (N, n, 2); depths are jittered so that the copies do
    not tie in the sort (reference misc/renderer2d.py:60-70)."""
    count, n, _ = offsets.shape
    copies = points.apply(lambda t: torch.repeat_interleave(t, repeats=n, dim=0), batch_size=[count * n])
    jitter = torch.randn_like(copies.z_depth) * depth_noise
    return copies.replace(position=copies.position + offsets.reshape(-1, 2),
                          z_depth=torch.clamp_min(copies.z_depth + jitter, 1e-6))


def repeat_sample_gaussians(samples: torch.Tensor, points: Gaussians2D, n: int = 2) -> torch.Tensor:
    """unit-space samples (N, n, 2) -> offsets in pixels through every Gaussian's own frame"""
    basis = point_basis(points).repeat_interleave(repeats=n, dim=0)
    return (basis @ samples.reshape(-1, 2, 1)).reshape(-1, n, 2)


def sample_gaussians(points: Gaussians2D) -> torch.Tensor:
    """one offset per Gaussian drawn from it (reference misc/renderer2d.py:99-101)"""
    unit = torch.randn_like(points.position)
    return (point_basis(points) @ unit.unsqueeze(-1)).squeeze(-1)


def split_gaussians2d(points: Gaussians2D, n: int = 2, scaling: Optional[float] = None,
                      depth_noise: float = 1e-2) -> Gaussians2D:
    """The splitting step of 3DGS densification in 2D: n shrunken copies at positions sampled from the parent
    (reference misc/renderer2d.py:73-96; default shrink 1/sqrt(n))."""
    samples = 0.5 * torch.randn((points.batch_size[0], n, 2), device=points.position.device)
    offsets = repeat_sample_gaussians(samples, points, n)
    shrink = 1.0 / math.sqrt(n) if scaling is None else scaling
    return split_with_offsets(points.replace(log_scaling=points.log_scaling + math.log(shrink)), offsets, depth_noise)


def uniform_split_gaussians2d(points: Gaussians2D, n: int = 2, scaling: Optional[float] = None,
                              depth_noise: float = 1e-2, sep: float = 0.7, random_axis: bool = False,
                              eps: float = 1e-6) -> Gaussians2D:
    """n copies evenly spaced in [-sep, sep] sigma along one axis (the longest, or one drawn in proportion to the
    sigmas), shrunk along that axis only (reference misc/renderer2d.py:109-132)."""
    if random_axis:
        weights = torch.nn.functional.normalize(points.scaling + eps, p=1, dim=1)
        axis = torch.multinomial(weights, num_samples=1).squeeze(1)
    else:
        axis = torch.argmax(points.log_scaling, dim=1)
    along = torch.nn.functional.one_hot(axis, num_classes=2)
    steps = torch.linspace(-sep, sep, n, device=points.position.device)
    offsets = repeat_sample_gaussians(steps.view(1, -1, 1) * along.view(-1, 1, 2), points, n)
    shrink = math.sqrt(n) / n if scaling is None else scaling
    return split_with_offsets(points.set_scaling(points.scaling * (along * shrink + (1 - along))), offsets, depth_noise)
