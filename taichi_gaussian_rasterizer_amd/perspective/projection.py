"""Perspective projection of 3D Gaussians to packed 2D Gaussians (HIP).

Same operator interface as the reference perspective/projection.py:190-248 (`apply`,
`project_to_image`): returns (points (V,7), depth (V,1), indexes (V) int64) for the Gaussians in
view, differentiable w.r.t. position, log_scaling, rotation, alpha_logit, T_camera_world and
projection; `indexes` is non-differentiable (:155).
"""
from __future__ import annotations

from numbers import Integral
from typing import Tuple

import torch

from .. import _native as nv
from ..data_types import Gaussians3D, RasterConfig
from .params import CameraParams


def _config(blur_cov, clamp_margin, alpha_threshold) -> RasterConfig:
    return RasterConfig(blur_cov=float(blur_cov), clamp_margin=float(clamp_margin),
                        alpha_threshold=float(alpha_threshold))


class _ProjectFunction(torch.autograd.Function):
    """forward: gs_project_fwd (project + cull + compact + ndc depth); backward: gs_project_bwd."""

    @staticmethod
    @nv.on_tensor_device
    def forward(ctx, position, log_scaling, rotation, alpha_logit, T_camera_world, projection, image_size,
                depth_range, config: RasterConfig):
        nv.require_device(position, log_scaling, rotation, alpha_logit, T_camera_world, projection,
                          what="project_to_image")
        lib = nv.lib()
        n = position.shape[0]
        dev = position.device
        cfg = nv.make_config(config)
        T = T_camera_world.contiguous()
        proj = projection.contiguous()
        points = torch.empty((n, 7), dtype=torch.float32, device=dev)
        depth = torch.empty((n, 1), dtype=torch.float32, device=dev)
        ndc = torch.empty((n, 1), dtype=torch.float32, device=dev)
        indexes = torch.empty((n,), dtype=torch.int64, device=dev)
        slot_of = torch.empty((n,), dtype=torch.int32, device=dev)
        count = torch.empty((1,), dtype=torch.int32, device=dev)
        nbytes = lib.gs_project_scratch_bytes(n)
        scratch = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=dev)
        nv.check(lib.gs_project_fwd(n, nv.ptr(position), nv.ptr(log_scaling), nv.ptr(rotation), nv.ptr(alpha_logit),
                                    nv.ptr(T), nv.ptr(proj), int(image_size[0]), int(image_size[1]),
                                    float(depth_range[0]), float(depth_range[1]), cfg, nv.ptr(points), nv.ptr(depth),
                                    nv.ptr(ndc), nv.ptr(indexes), nv.ptr(slot_of), nv.ptr(count), None, 0,
                                    None, nv.ptr(scratch), nbytes, nv.stream()), "gs_project_fwd")
        v = int(count.item())  # the one host sync of the stage (the reference's torch.nonzero, :146)
        points, depth, ndc, indexes = points[:v], depth[:v], ndc[:v], indexes[:v]
        ctx.image_size = (int(image_size[0]), int(image_size[1]))
        ctx.config = config
        ctx.num_visible = v
        ctx.save_for_backward(position, log_scaling, rotation, alpha_logit, T, proj, slot_of)
        ctx.mark_non_differentiable(indexes, ndc)
        ctx.slot_of = slot_of
        return points, depth, indexes, ndc

    @staticmethod
    @nv.on_tensor_device
    def backward(ctx, dpoints, ddepth, _dindexes, _dndc):
        position, log_scaling, rotation, alpha_logit, T, proj, slot_of = ctx.saved_tensors
        lib = nv.lib()
        n = position.shape[0]
        dev = position.device
        need_T, need_proj = ctx.needs_input_grad[4], ctx.needs_input_grad[5]
        d_pos = torch.empty_like(position)
        d_ls = torch.empty_like(log_scaling)
        d_rot = torch.empty_like(rotation)
        d_al = torch.empty_like(alpha_logit)
        d_T = torch.empty((4, 4), dtype=torch.float32, device=dev) if need_T else None
        d_proj = torch.empty((4,), dtype=torch.float32, device=dev) if need_proj else None
        nbytes = lib.gs_project_bwd_scratch_bytes(n) if (need_T or need_proj) else 0
        scratch = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=dev)
        gp = dpoints.contiguous() if dpoints is not None else None
        gd = ddepth.contiguous() if ddepth is not None else None
        nv.require_device(gp, gd, what="project_to_image backward")
        nv.check(lib.gs_project_bwd(n, ctx.num_visible, nv.ptr(position), nv.ptr(log_scaling), nv.ptr(rotation),
                                    nv.ptr(alpha_logit), nv.ptr(T), nv.ptr(proj), ctx.image_size[0],
                                    ctx.image_size[1], nv.make_config(ctx.config), nv.ptr(slot_of), nv.ptr(gp), 7,
                                    nv.ptr(gd), None, 1, nv.ptr(d_pos), nv.ptr(d_ls), nv.ptr(d_rot), nv.ptr(d_al),
                                    nv.ptr(d_T), nv.ptr(d_proj), nv.ptr(scratch), nbytes, nv.stream()),
                 "gs_project_bwd")
        return d_pos, d_ls, d_rot, d_al, d_T, d_proj, None, None, None


def _check_inputs(position, log_scaling, rotation, alpha_logit, T_camera_world, projection, image_size, depth_range):
    for name, t in (("position", position), ("log_scaling", log_scaling), ("rotation", rotation),
                    ("alpha_logit", alpha_logit), ("T_camera_world", T_camera_world), ("projection", projection)):
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"{name} must be a torch.Tensor, got {type(t).__name__}")
    if not (len(image_size) == 2 and all(isinstance(x, Integral) for x in image_size)):
        raise TypeError(f"image_size must be Tuple[Integral, Integral], got {image_size!r}")
    if not (len(depth_range) == 2 and all(isinstance(x, float) for x in depth_range)):
        raise TypeError(f"depth_range must be Tuple[float, float], got {depth_range!r}")
    n = position.shape[0]
    assert position.shape == (n, 3) and log_scaling.shape == (n, 3) and rotation.shape == (n, 4) \
        and alpha_logit.shape == (n, 1), "gaussian tensors must be (N,3),(N,3),(N,4),(N,1)"
    assert T_camera_world.shape == (4, 4) and projection.shape == (4,)


def project_with_ndc(position, log_scaling, rotation, alpha_logit, T_camera_world, projection, image_size,
                     depth_range, config: RasterConfig):
    """(points, depth, indexes, ndc_depth): the fused kernel also emits the sort depth."""
    _check_inputs(position, log_scaling, rotation, alpha_logit, T_camera_world, projection, image_size, depth_range)
    points, depth, indexes, ndc = _ProjectFunction.apply(
        position.contiguous(), log_scaling.contiguous(), rotation.contiguous(), alpha_logit.contiguous(),
        T_camera_world, projection, image_size, depth_range, config)
    indexes._gs_unique = True  # ascending, no repeats: lets evaluate_sh_at's backward skip atomics
    node = points.grad_fn
    slot_of = getattr(node, "slot_of", None) if node is not None else None
    if slot_of is not None:
        indexes._gs_slot_of = slot_of  # inverse map: evaluate_sh_at's backward becomes one dense pass
    return points, depth, indexes, ndc


def apply(position: torch.Tensor, log_scaling: torch.Tensor, rotation: torch.Tensor, alpha_logit: torch.Tensor,
          T_camera_world: torch.Tensor, projection: torch.Tensor, image_size: Tuple[Integral, Integral],
          depth_range: Tuple[float, float], blur_cov: float = 0.0, clamp_margin: float = 0.15,
          alpha_threshold: float = 1. / 255.) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Reference perspective/projection.py:190-215."""
    points, depth, indexes, _ = project_with_ndc(position, log_scaling, rotation, alpha_logit, T_camera_world,
                                                 projection, image_size, depth_range,
                                                 _config(blur_cov, clamp_margin, alpha_threshold))
    return points, depth, indexes


def project_to_image(gaussians: Gaussians3D, camera_params: CameraParams, config: RasterConfig
                     ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Project 3D gaussians to 2D gaussians in image space (EWA approximation, Zwicker et al. 2003).

    Returns:
      points:  (V, 7) packed 2D gaussians [mean.xy, axis.xy, sigma.xy, alpha]
      depths:  (V, 1) camera-space depth
      indexes: (V)    indexes of the gaussians in view (int64, ascending)
    """
    if not isinstance(gaussians, Gaussians3D):
        raise TypeError(f"gaussians must be Gaussians3D, got {type(gaussians).__name__}")
    if not isinstance(camera_params, CameraParams):
        raise TypeError(f"camera_params must be CameraParams, got {type(camera_params).__name__}")
    if not isinstance(config, RasterConfig):
        raise TypeError(f"config must be RasterConfig, got {type(config).__name__}")
    return apply(*gaussians.shape_tensors(), camera_params.T_camera_world, camera_params.projection,
                 camera_params.image_size, camera_params.depth_range, config.blur_cov, config.clamp_margin,
                 config.alpha_threshold)
