"""Differentiable row gather (reference misc/indexing.py:52-58 `index_features`: a Taichi kernel there).  On this
stack torch's own indexing kernel and its scatter-add adjoint are the native path; trailing dimensions are kept."""
from __future__ import annotations

import torch


def index_features(features: torch.Tensor, indexes: torch.Tensor) -> torch.Tensor:
    if not isinstance(features, torch.Tensor) or not isinstance(indexes, torch.Tensor):
        raise TypeError("features and indexes must be torch.Tensor")
    return features.index_select(0, indexes.contiguous())
