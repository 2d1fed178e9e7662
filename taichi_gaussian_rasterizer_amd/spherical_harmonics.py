"""View-dependent colour from real spherical harmonics (HIP).

Operator interface of the reference spherical_harmonics.py:167-178:
evaluate_sh_at(sh_params (N,C,D), positions (N,3), indexes (V), camera_pos (3)) -> (V,C), degree 0..3,
out = clamp(sum_d Y_d(dir) * sh[idx,c,d] + 0.5, 0, 1); differentiable w.r.t. sh_params, positions
and camera_pos (:149-161).
"""
from __future__ import annotations

import math

import torch

from . import _native as nv


def check_sh_degree(sh_features):
    assert len(sh_features.shape) == 3, f"SH features must have 3 dimensions, got {sh_features.shape}"
    n_sh = sh_features.shape[2]
    n = int(math.sqrt(n_sh))
    assert n * n == n_sh, f"SH feature count must be square, got {n_sh} ({sh_features.shape})"
    return n - 1


class _SHFunction(torch.autograd.Function):
    @staticmethod
    @nv.on_tensor_device
    def forward(ctx, params, points, indexes, camera_pos, degree, unique, slot_of):
        nv.require_device(params, points, camera_pos, what="evaluate_sh_at")
        nv.require_device(indexes, dtype=torch.int64, what="evaluate_sh_at indexes")
        lib = nv.lib()
        v, C = indexes.shape[0], params.shape[1]
        out = torch.empty((v, C), dtype=torch.float32, device=params.device)
        nv.check(lib.gs_sh_fwd(v, None, C, degree, nv.ptr(params), nv.ptr(points), nv.ptr(indexes),
                               nv.ptr(camera_pos), nv.ptr(out), C, nv.stream()), "gs_sh_fwd")
        ctx.save_for_backward(params, points, indexes, camera_pos, out)
        ctx.degree, ctx.unique, ctx.slot_of = degree, unique, slot_of
        return out

    @staticmethod
    @nv.on_tensor_device
    def backward(ctx, doutput):
        params, points, indexes, camera_pos, out = ctx.saved_tensors
        lib = nv.lib()
        n, C = params.shape[0], params.shape[1]
        need_pts, need_cam = ctx.needs_input_grad[1], ctx.needs_input_grad[3]
        d_params = torch.empty_like(params)
        d_points = torch.empty_like(points) if need_pts else None
        d_cam = torch.empty_like(camera_pos) if need_cam else None
        go = doutput.contiguous()
        nv.require_device(go, what="evaluate_sh_at backward")
        nv.check(lib.gs_sh_bwd(n, indexes.shape[0], C, ctx.degree, nv.ptr(params), nv.ptr(points), nv.ptr(indexes),
                               int(ctx.unique), nv.ptr(ctx.slot_of), nv.ptr(camera_pos), nv.ptr(go), C,
                               nv.ptr(out), C, nv.ptr(d_params), nv.ptr(d_points), nv.ptr(d_cam), nv.stream()),
                 "gs_sh_bwd")
        return d_params, d_points, None, d_cam, None, None, None


def evaluate_sh_at(sh_params: torch.Tensor,   # M, K, (degree + 1)^2  (usually K=3, for RGB)
                   positions: torch.Tensor,   # M, 3
                   indexes: torch.Tensor,     # N  (indexes to gaussians) 0 to M
                   camera_pos: torch.Tensor   # 3
                   ) -> torch.Tensor:         # N, K
    for name, t in (("sh_params", sh_params), ("positions", positions), ("indexes", indexes),
                    ("camera_pos", camera_pos)):
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"{name} must be a torch.Tensor, got {type(t).__name__}")
    degree = check_sh_degree(sh_params)
    assert 0 <= degree <= 3, f"SH degree must be between 0 and 3, got {degree}"
    unique = getattr(indexes, "_gs_unique", None)
    if unique is None and sh_params.requires_grad and indexes.is_cuda and indexes.shape[0] > 1:
        # strictly ascending indexes (an arange, any sorted visible list) hit every row at most once: the backward
        # can then store gradient rows instead of adding 48 floats per Gaussian atomically (10x faster).  One
        # device read per index tensor; the verdict is cached on the tensor.
        unique = bool((indexes[1:] > indexes[:-1]).all().item())
        try:
            indexes._gs_unique = unique
        except AttributeError:
            pass
    unique = bool(unique)
    slot_of = getattr(indexes, "_gs_slot_of", None)  # inverse of the projection's visible list
    if slot_of is not None and slot_of.shape[0] != sh_params.shape[0]:
        slot_of = None
    return _SHFunction.apply(sh_params.contiguous(), positions.contiguous(), indexes.contiguous(),
                             camera_pos.contiguous(), degree, unique, slot_of)
