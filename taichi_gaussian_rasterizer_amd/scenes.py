"""Synthetic scene generators (CPU torch, seeded) used by bench.py and tests/.

random_camera / random_3d_gaussians / random_2d_gaussians consume torch's generator exactly as the reference's test
generators do, so a torch seed names the same scene in both code bases.  benchmark_scene is the fixed-camera scene
of SURVEY.md section 8(d) that BASELINE.json's configs are defined on.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from .data_types import Gaussians2D, Gaussians3D
from .perspective.params import CameraParams
from .torch_lib import projection as tp


# ---------------------------------------------------------------------------------------------------
# Seeded random scenes.  The number, shape and order of the draws from torch's global generator follow the reference's
# test generators (tests/random_data.py:15-105), so a seed names the same camera / Gaussians there and here.
def _uniform(shape, lo: float, hi: float) -> torch.Tensor:
    return torch.rand(shape) * (hi - lo) + lo


def _unit_rows(t: torch.Tensor) -> torch.Tensor:
    return t / torch.norm(t, dim=1, keepdim=True)


def _screen(image_size) -> torch.Tensor:
    return torch.tensor([float(image_size[0]), float(image_size[1])]).unsqueeze(0)


def random_camera(pos_scale: float = 1., image_size: Optional[Tuple[int, int]] = None,
                  image_size_range=(256, 1024), near_plane=0.1) -> CameraParams:
    """a camera at a random pose, field of view 30..100 degrees, principal point near the image centre"""
    assert near_plane > 0
    orientation = F.normalize(torch.randn((1, 4)))                 # draw 1: quaternion
    centre = torch.randn((3)) * pos_scale                          # draw 2: position
    view = torch.inverse(tp.join_rt(tp.quat_to_mat(orientation), centre))
    if image_size is None:                                         # draw 3 (only without a fixed size)
        image_size = [int(v) for v in torch.randint(size=(2,), low=image_size_range[0], high=image_size_range[1])]
    width, height = image_size
    cx, cy = torch.tensor([width / 2, height / 2]) + torch.randn(2) * (width / 20)   # draw 4
    half_fov = torch.deg2rad(torch.rand(1) * 70 + 30) / 2                             # draw 5
    intrinsics = torch.tensor([width / (2 * torch.tan(half_fov)), height / (2 * torch.tan(half_fov)), cx, cy],
                              dtype=torch.float32)
    return CameraParams(projection=intrinsics, T_camera_world=view, near_plane=near_plane,
                        far_plane=near_plane * 1000., image_size=(width, height))


def random_3d_gaussians(n, camera_params: CameraParams, scale_factor: float = 1.0, alpha_range=(0.1, 0.9),
                        margin=0.0) -> Gaussians3D:
    """n Gaussians scattered through the view frustum (a `margin` fraction of them just outside the image), sized so
    that their projections cover the image about once at scale_factor 1"""
    width, _ = camera_params.image_size
    pixels = (torch.rand(n, 2) * (1 + margin) - margin * 0.5) * _screen(camera_params.image_size)
    z = tp.inverse_ndc_depth(torch.rand(n), camera_params.near_plane, camera_params.far_plane)
    centres = tp.unproject_points(pixels, z.unsqueeze(1), camera_params.T_image_world)
    world_size = (width / math.sqrt(n)) * (z / camera_params.T_image_camera[0, 0]) * scale_factor
    sigmas = (torch.rand(n, 3) + 0.2) * world_size.unsqueeze(1)
    quaternions = F.normalize(torch.randn(n, 4), dim=1)
    opacity = _uniform(n, *alpha_range)
    return Gaussians3D(position=centres, log_scaling=sigmas.log(), rotation=quaternions,
                       alpha_logit=tp.inverse_sigmoid(opacity).unsqueeze(1), feature=torch.rand(n, 3),
                       batch_size=(n,))


def random_2d_gaussians(n, image_size: Tuple[int, int], num_channels=3, scale_factor=1.0, alpha_range=(0.1, 0.9),
                        depth_range=(0.0, 1.0)) -> Gaussians2D:
    """n screen-space Gaussians for the 2D rasterizer tests and the image-fitting example"""
    centres = torch.rand(n, 2) * _screen(image_size)
    z = _uniform((n, 1), *depth_range)
    sigmas = (torch.rand(n, 2) + 0.2) * (scale_factor * image_size[0] / (1 + math.sqrt(n)))
    axes = _unit_rows(torch.randn(n, 2))
    opacity = _uniform(n, *alpha_range)
    return Gaussians2D(position=centres, z_depth=z, log_scaling=sigmas.log(), rotation=axes,
                       alpha_logit=tp.inverse_sigmoid(opacity), feature=torch.rand(n, num_channels),
                       batch_size=(n,))


def benchmark_camera(image_size: Tuple[int, int], fov_deg: float = 60.0, near=0.1, far=100.0) -> CameraParams:
    w, h = image_size
    f = w / (2 * math.tan(math.radians(fov_deg) / 2))
    return CameraParams(projection=torch.tensor([f, f, w / 2, h / 2], dtype=torch.float32),
                        T_camera_world=torch.eye(4, dtype=torch.float32), near_plane=near, far_plane=far,
                        image_size=(w, h))


def benchmark_scene(n: int, image_size: Tuple[int, int], sh_degree: int = 3, seed: int = 0,
                    scale_factor: float = 2.0, margin: float = 0.1, alpha_range=(0.1, 0.9)):
    """identity camera, fov 60, near 0.1, far 100; draw order uv, depth, scale,
    rotation, alpha, colour.  Returns (Gaussians3D with SH feature (N,3,(deg+1)^2), CameraParams)."""
    gen = torch.Generator().manual_seed(seed)
    camera = benchmark_camera(image_size)
    w, h = image_size
    uv = (torch.rand(n, 2, generator=gen) * (1 + margin) - margin * 0.5) * torch.tensor([w, h], dtype=torch.float32)
    z = tp.inverse_ndc_depth(torch.rand(n, generator=gen), camera.near_plane, camera.far_plane)
    position = tp.unproject_points(uv, z.unsqueeze(1), camera.T_image_world)
    fx = float(camera.projection[0])
    scale = (w / math.sqrt(n)) * (z / fx) * scale_factor
    scaling = (torch.rand(n, 3, generator=gen) + 0.2) * scale.unsqueeze(1)
    rotation = F.normalize(torch.randn(n, 4, generator=gen), dim=1)
    lo, hi = alpha_range
    alpha = torch.rand(n, generator=gen) * (hi - lo) + lo
    d = (sh_degree + 1) ** 2
    sh = torch.zeros(n, 3, d)
    sh[:, :, 0] = (torch.rand(n, 3, generator=gen) - 0.5) / 0.2820948
    if d > 1:
        sh[:, :, 1:] = 0.05 * torch.randn(n, 3, d - 1, generator=gen)
    gaussians = Gaussians3D(position=position.contiguous(), log_scaling=torch.log(scaling), rotation=rotation,
                            alpha_logit=tp.inverse_sigmoid(alpha).unsqueeze(1), feature=sh, batch_size=(n,))
    return gaussians, camera
