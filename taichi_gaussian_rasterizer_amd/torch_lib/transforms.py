"""Rigid-transform helpers callers of the reference import from `torch_lib.transforms` (reference
torch_lib/transforms.py:4-48): quaternion (x, y, z, w) -> rotation matrix, 4x4 <-> (R, t), homogeneous points."""
from __future__ import annotations

from typing import Tuple

import torch

from .projection import join_rt, make_homog, quat_to_mat, transform44

__all__ = ["quat_to_mat", "split_rt", "join_rt", "make_homog", "transform44", "transform33"]


def split_rt(transform: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(…, 4, 4) -> rotation (…, 3, 3) and translation (…, 3), both contiguous"""
    return transform[..., :3, :3].contiguous(), transform[..., :3, 3].contiguous()


def transform33(transform: torch.Tensor, points: torch.Tensor) -> torch.Tensor:
    """one 3x3 matrix applied to (N, 3) points"""
    return (points.reshape(-1, 3) @ transform.reshape(3, 3).transpose(0, 1)).reshape(-1, 3)
