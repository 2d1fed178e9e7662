"""Morton-order argsort of 3D points (reference misc/morton_sort.py:121-164): the ordering a trainer
applies to its Gaussians from time to time so that neighbours in memory are neighbours in space, which
is what the rasterizer's row gathers and the mapper's region binning like.  Codes by a HIP kernel,
ordering by the package's radix sort (hip_lib.radix_argsort)."""
from __future__ import annotations

import ctypes

import torch

from .. import _native as nv
from ..hip_lib import radix_sort_pairs


@nv.on_tensor_device
def morton_codes(points: torch.Tensor, resolution: float, size: int = 2 ** 20) -> torch.Tensor:
    """(N) int64 tensor holding the unsigned 63-bit codes of the reference's Grid.morton_code64 for the
    grid `grid_at_resolution(points, resolution, size)` (lower = per-axis minimum, cell edge = resolution)."""
    nv.require_device(points, what="morton_codes")
    assert points.ndim == 2 and points.shape[1] == 3, f"points must be (N,3), got {points.shape}"
    pts = points.contiguous()
    n = pts.shape[0]
    codes = torch.empty((n,), dtype=torch.int64, device=pts.device)
    if n == 0:
        return codes
    lower = pts.min(dim=0).values.cpu()
    # reference: inc = (upper - lower) / size with upper = lower + size * resolution, evaluated in f32
    upper = lower + torch.tensor(float(size) * float(resolution), dtype=torch.float32)
    inc = float(((upper - lower) / float(size))[0])
    lo = (ctypes.c_float * 3)(*[float(x) for x in lower])
    nv.check(nv.lib().gs_morton_codes64(n, nv.ptr(pts), lo, inc, int(size), nv.ptr(codes), nv.stream()),
             "gs_morton_codes64")
    return codes


def argsort(points: torch.Tensor, resolution: float) -> torch.Tensor:
    codes = morton_codes(points, resolution)
    idx = torch.arange(points.shape[0], dtype=torch.int32, device=points.device)
    _, idx = radix_sort_pairs(codes, idx, 0, 63)
    return idx


def sort(points: torch.Tensor, resolution: float) -> torch.Tensor:
    return points[argsort(points, resolution).long()]


def argsort_dedup(points: torch.Tensor, resolution: float) -> torch.Tensor:
    """Indices (into the Morton-sorted order's source) of one representative point per occupied cell."""
    codes = morton_codes(points, resolution)
    idx = torch.arange(points.shape[0], dtype=torch.int32, device=points.device)
    codes_sorted, idx = radix_sort_pairs(codes, idx, 0, 63)
    _, counts = torch.unique_consecutive(codes_sorted, return_counts=True)
    last = torch.cumsum(counts, dim=0) - 1
    return idx[last].long()


def sort_dedup(points: torch.Tensor, resolution: float) -> torch.Tensor:
    return points[argsort_dedup(points, resolution)]
