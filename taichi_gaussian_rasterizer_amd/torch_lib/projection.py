"""Small torch helpers that sit ON the render path in the reference (torch_lib/projection.py):
ndc_depth (:120-123, called from renderer.py:189), its inverse (:126-129) and the point
(un)projection helpers used to build synthetic scenes (:48-60).  The reference's torch
*oracles* for projection / SH are NOT restated here -- parity is checked against golden
vectors generated from the reference itself (tests/golden, oracle/make_golden.py).
"""
from __future__ import annotations

import torch


def ndc_depth(depth: torch.Tensor, near: float, far: float) -> torch.Tensor:
    """ndc from 0 (near) to 1 (far).  Same formula as the reference; the sort-key path inside
    render_gaussians uses the fused HIP version with a fixed f32 operation order."""
    return 1 - (1. / depth - 1. / far) / (1. / near - 1. / far)


def inverse_ndc_depth(ndc_depth: torch.Tensor, near: float, far: float) -> torch.Tensor:
    return 1.0 / ((1.0 - ndc_depth) * (1 / near - 1 / far) + 1 / far)


def make_homog(points):
    shape = list(points.shape)
    shape[-1] = 1
    return torch.cat([points, torch.ones(shape, dtype=points.dtype, device=points.device)], dim=-1)


def transform44(transform, points):
    return (transform.reshape(1, 4, 4) @ points.reshape(-1, 4, 1))[..., 0]


def project_points(transform, xyz):
    homog = transform44(transform, make_homog(xyz))
    depth = homog[..., 2:3]
    return homog[..., 0:2] / depth, depth


def unproject_points(uv, depth, transform):
    points = torch.cat([uv * depth, depth, torch.ones_like(depth)], dim=-1)
    transformed = transform44(torch.inverse(transform), points)
    return transformed[..., 0:3] / transformed[..., 3:4]


def inverse_sigmoid(x: torch.Tensor):
    return torch.log(x / (1 - x))


def quat_to_mat(quat: torch.Tensor) -> torch.Tensor:
    """xyzw quaternion -> rotation matrix (reference torch_lib/transforms.py:4-15)."""
    x, y, z, w = quat[..., 0], quat[..., 1], quat[..., 2], quat[..., 3]
    x2, y2, z2 = x * x, y * y, z * z
    return torch.stack([
        1 - 2 * y2 - 2 * z2, 2 * x * y - 2 * w * z, 2 * x * z + 2 * w * y,
        2 * x * y + 2 * w * z, 1 - 2 * x2 - 2 * z2, 2 * y * z - 2 * w * x,
        2 * x * z - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x2 - 2 * y2], dim=-1).reshape(quat.shape[:-1] + (3, 3))


def join_rt(r, t):
    T = torch.eye(4, device=r.device, dtype=r.dtype)
    T[0:3, 0:3] = r
    T[0:3, 3] = t
    return T
