"""Synthetic scene generators (CPU torch, seeded) used by bench.py and tests/.

random_camera / random_3d_gaussians / random_2d_gaussians restate the reference's test
generators (tests/random_data.py:15-105) draw for draw, so a given torch seed produces the
same scene the reference's tests would see.  benchmark_scene is the fixed-camera scene of
SURVEY.md section 8(d) that BASELINE.json's configs are defined on.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from .data_types import Gaussians2D, Gaussians3D
from .perspective.params import CameraParams
from .torch_lib import projection as tp


def random_camera(pos_scale: float = 1., image_size: Optional[Tuple[int, int]] = None,
                  image_size_range=(256, 1024), near_plane=0.1) -> CameraParams:
    assert near_plane > 0
    q = F.normalize(torch.randn((1, 4)))
    t = torch.randn((3)) * pos_scale
    T_world_camera = tp.join_rt(tp.quat_to_mat(q), t)
    T_camera_world = torch.inverse(T_world_camera)
    if image_size is None:
        lo, hi = image_size_range
        image_size = [x.item() for x in torch.randint(size=(2,), low=lo, high=hi)]
    w, h = image_size
    cx, cy = torch.tensor([w / 2, h / 2]) + torch.randn(2) * (w / 20)
    fov = torch.deg2rad(torch.rand(1) * 70 + 30)
    fx = w / (2 * torch.tan(fov / 2))
    fy = h / (2 * torch.tan(fov / 2))
    projection = torch.tensor([fx, fy, cx, cy], dtype=torch.float32)
    return CameraParams(T_camera_world=T_camera_world, projection=projection, image_size=(w, h),
                        near_plane=near_plane, far_plane=near_plane * 1000.)


def random_3d_gaussians(n, camera_params: CameraParams, scale_factor: float = 1.0, alpha_range=(0.1, 0.9),
                        margin=0.0) -> Gaussians3D:
    w, h = camera_params.image_size
    uv_pos = (torch.rand(n, 2) * (1 + margin) - margin * 0.5) * torch.tensor([w, h], dtype=torch.float32).unsqueeze(0)
    depth = tp.inverse_ndc_depth(torch.rand(n), camera_params.near_plane, camera_params.far_plane)
    position = tp.unproject_points(uv_pos, depth.unsqueeze(1), camera_params.T_image_world)
    fx = camera_params.T_image_camera[0, 0]
    scale = (w / math.sqrt(n)) * (depth / fx) * scale_factor
    scaling = (torch.rand(n, 3) + 0.2) * scale.unsqueeze(1)
    rotation = F.normalize(torch.randn(n, 4), dim=1)
    low, high = alpha_range
    alpha = torch.rand(n) * (high - low) + low
    return Gaussians3D(position=position, log_scaling=torch.log(scaling), rotation=rotation,
                       alpha_logit=tp.inverse_sigmoid(alpha).unsqueeze(1), feature=torch.rand(n, 3),
                       batch_size=(n,))


def random_2d_gaussians(n, image_size: Tuple[int, int], num_channels=3, scale_factor=1.0, alpha_range=(0.1, 0.9),
                        depth_range=(0.0, 1.0)) -> Gaussians2D:
    w, h = image_size
    position = torch.rand(n, 2) * torch.tensor([w, h], dtype=torch.float32).unsqueeze(0)
    depth = torch.rand((n, 1)) * (depth_range[1] - depth_range[0]) + depth_range[0]
    density_scale = scale_factor * w / (1 + math.sqrt(n))
    scaling = (torch.rand(n, 2) + 0.2) * density_scale
    rotation = torch.randn(n, 2)
    rotation = rotation / torch.norm(rotation, dim=1, keepdim=True)
    low, high = alpha_range
    alpha = torch.rand(n) * (high - low) + low
    return Gaussians2D(position=position, z_depth=depth, log_scaling=torch.log(scaling), rotation=rotation,
                       alpha_logit=tp.inverse_sigmoid(alpha), feature=torch.rand(n, num_channels), batch_size=(n,))


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
