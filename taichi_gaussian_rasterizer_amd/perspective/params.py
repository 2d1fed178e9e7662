"""The pinhole camera handed to the renderer (reference perspective/params.py:10-102: same constructor fields and
derived quantities, so cameras built for the reference work unchanged).

Two tensors describe a view: `projection` = [fx, fy, cx, cy] in pixels and `T_camera_world`, the 4x4 world -> camera
matrix; the clipping planes and the image size are plain Python values.
"""
from __future__ import annotations

import dataclasses
from typing import Tuple

import torch


def _intrinsic_matrix(projection: torch.Tensor, size: int) -> torch.Tensor:
    """K as a (size, size) matrix (3: pixel <- camera ray, 4: homogeneous)"""
    K = torch.eye(size, device=projection.device, dtype=projection.dtype)
    K[0, 0], K[1, 1] = projection[0], projection[1]
    K[0, 2], K[1, 2] = projection[2], projection[3]
    return K


@dataclasses.dataclass
class CameraParams:
    projection: torch.Tensor      # (4): fx, fy, cx, cy
    T_camera_world: torch.Tensor  # (4, 4) view matrix
    near_plane: float
    far_plane: float
    image_size: Tuple[int, int]   # (width, height)

    def __post_init__(self):
        for name in ("projection", "T_camera_world"):
            if not isinstance(getattr(self, name), torch.Tensor):
                raise TypeError(f"{name} must be a torch.Tensor")
        assert self.projection.shape == (4,), f"Expected shape (4,), got {self.projection.shape}"
        assert self.T_camera_world.shape == (4, 4), f"Expected shape (4, 4), got {self.T_camera_world.shape}"
        assert len(self.image_size) == 2, f"image_size is (width, height), got {self.image_size}"
        assert 0 < self.near_plane < self.far_plane, \
            f"clipping planes must satisfy 0 < near < far, got {self.near_plane}, {self.far_plane}"

    # ---- plain accessors
    device = property(lambda self: self.projection.device)
    dtype = property(lambda self: self.projection.dtype)
    depth_range = property(lambda self: (self.near_plane, self.far_plane))
    focal_length = property(lambda self: self.projection[:2])
    principal_point = property(lambda self: self.projection[2:])

    # ---- matrices
    @property
    def T_image_camera(self) -> torch.Tensor:
        return _intrinsic_matrix(self.projection, 3)

    @property
    def T_image_world(self) -> torch.Tensor:
        return _intrinsic_matrix(self.projection, 4) @ self.T_camera_world

    @property
    def camera_position(self) -> torch.Tensor:
        """camera centre in world coordinates = translation column of the inverse view matrix.  A 4x4 LU on the GPU
        costs more than the SH kernel it feeds, so the inverse is taken on the host unless gradients flow through it
        (the fused frame computes it in the projection kernel and never comes here)."""
        T = self.T_camera_world
        if T.is_cuda and not T.requires_grad:
            return torch.linalg.inv(T.detach().cpu())[:3, 3].to(T.device)
        return torch.linalg.inv(T)[:3, 3]

    # ---- derived cameras
    def _with(self, **changes) -> "CameraParams":
        return dataclasses.replace(self, **changes)

    def transformed(self, t: torch.Tensor) -> "CameraParams":
        """the same camera after moving the world by `t` (4x4)"""
        return self._with(T_camera_world=t @ self.T_camera_world)

    def scale_image(self, scale: float) -> "CameraParams":
        width, height = self.image_size
        return self._with(image_size=(int(width * scale), int(height * scale)), projection=self.projection * scale)

    def detach(self) -> "CameraParams":
        return self._with(projection=self.projection.detach(), T_camera_world=self.T_camera_world.detach())

    def to(self, device=None, dtype=None) -> "CameraParams":
        move = dict(device=device, dtype=dtype)
        return self._with(projection=self.projection.to(**move), T_camera_world=self.T_camera_world.to(**move))

    def requires_grad_(self, requires_grad: bool) -> "CameraParams":
        for t in (self.projection, self.T_camera_world):
            t.requires_grad_(requires_grad)
        return self

    def __repr__(self) -> str:
        fx, fy, cx, cy = (float(v) for v in self.projection.detach().cpu())
        centre = ", ".join(f"{float(v):.3f}" for v in self.camera_position.detach().cpu())
        return (f"CameraParams({self.image_size[0]}x{self.image_size[1]}, fx={fx:.4f}, fy={fy:.4f}, cx={cx:.4f}, "
                f"cy={cy:.4f}, clipping={self.near_plane:.4f}-{self.far_plane:.4f}, position=({centre}))")
