"""Tile rasterizer: front-to-back alpha blending and its gradient (HIP).

Operator interface of the reference rasterizer/function.py:96-161: `rasterize_with_tiles`,
`rasterize`, `RasterOut`.  image / image_weight are (H,W,F) / (H,W); image_weight, visibility and
point_heuristic are non-differentiable (:72); gradients flow to gaussians2d and features.
"""
from __future__ import annotations

from numbers import Integral
from typing import NamedTuple, Optional, Tuple

import torch

from .. import _native as nv
from ..data_types import RasterConfig
from ..mapper.tile_mapper import map_to_tiles

RasterOut = NamedTuple('RasterOut', [
    ('image', torch.Tensor),
    ('image_weight', torch.Tensor),
    ('point_heuristic', Optional[torch.Tensor]),
    ('visibility', Optional[torch.Tensor])
])


class _RasterFunction(torch.autograd.Function):
    @staticmethod
    @nv.on_tensor_device
    def forward(ctx, gaussians, features, overlap_to_point, tile_overlap_ranges, image_size, config: RasterConfig):
        nv.require_device(gaussians, features, what="rasterize_with_tiles")
        nv.require_device(overlap_to_point, tile_overlap_ranges, dtype=torch.int32, what="rasterize_with_tiles tiles")
        lib = nv.lib()
        dev = features.device
        w, h = int(image_size[0]), int(image_size[1])
        v, F = gaussians.shape[0], features.shape[1]
        g, f = gaussians.contiguous(), features.contiguous()
        o2p, ranges = overlap_to_point.contiguous(), tile_overlap_ranges.contiguous()
        image = torch.empty((h, w, F), dtype=torch.float32, device=dev)
        alpha = torch.empty((h, w), dtype=torch.float32, device=dev)
        # reference function.py:48-59
        heur = (torch.zeros((v, 2), dtype=torch.float32, device=dev) if config.compute_point_heuristic
                else torch.empty((0, 2), dtype=torch.float32, device=dev))
        want_vis = config.compute_visibility or config.compute_point_heuristic
        vis = (torch.zeros((v,), dtype=torch.float32, device=dev) if want_vis
               else torch.empty((0,), dtype=torch.float32, device=dev))
        nv.check(lib.gs_raster_fwd(v, F, nv.ptr(g), nv.ptr(f), nv.ptr(ranges), nv.ptr(o2p), o2p.shape[0], w, h,
                                   nv.make_config(config), None, None, nv.ptr(image), nv.ptr(alpha),
                                   nv.ptr(vis) if want_vis else None, None, nv.stream()), "gs_raster_fwd")
        if not config.compute_visibility:
            vis_out = torch.empty((0,), dtype=torch.float32, device=dev) if not want_vis else vis
        else:
            vis_out = vis
        ctx.image_size, ctx.config = (w, h), config
        ctx.heur = heur
        ctx.mark_non_differentiable(alpha, vis_out, heur)
        ctx.save_for_backward(g, f, o2p, ranges, image)
        return image, alpha, heur, vis_out

    @staticmethod
    @nv.on_tensor_device
    def backward(ctx, grad_image, _ga, _gh, _gv):
        g, f, o2p, ranges, image = ctx.saved_tensors
        lib = nv.lib()
        v, F = g.shape[0], f.shape[1]
        w, h = ctx.image_size
        config = ctx.config
        gi = grad_image.contiguous()
        nv.require_device(gi, what="rasterize backward")
        row = lib.gs_grad_row_floats(F)
        rows = torch.zeros((v, row), dtype=torch.float32, device=g.device)
        nv.check(lib.gs_raster_bwd(v, F, nv.ptr(g), nv.ptr(f), nv.ptr(ranges), nv.ptr(o2p), o2p.shape[0], w, h,
                                   nv.make_config(config), None, None, nv.ptr(image), nv.ptr(gi), nv.ptr(rows),
                                   None, nv.stream()), "gs_raster_bwd")
        grad_g = torch.empty_like(g)
        grad_f = torch.empty_like(f)
        heur = ctx.heur if config.compute_point_heuristic else None
        nv.check(lib.gs_raster_bwd_unpack(v, F, nv.ptr(rows), nv.ptr(grad_g), nv.ptr(grad_f), nv.ptr(heur),
                                          nv.stream()), "gs_raster_bwd_unpack")
        return grad_g, grad_f, None, None, None, None


def _validate(gaussians2d, features, overlap_to_point, tile_overlap_ranges, image_size, config):
    for name, t in (("gaussians2d", gaussians2d), ("features", features), ("overlap_to_point", overlap_to_point),
                    ("tile_overlap_ranges", tile_overlap_ranges)):
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"{name} must be a torch.Tensor, got {type(t).__name__}")
    if not (len(image_size) == 2 and all(isinstance(x, Integral) for x in image_size)):
        raise TypeError(f"image_size must be Tuple[Integral, Integral], got {image_size!r}")
    if not isinstance(config, RasterConfig):
        raise TypeError(f"config must be RasterConfig, got {type(config).__name__}")
    assert gaussians2d.ndim == 2 and gaussians2d.shape[1] == 7, f"gaussians2d must be Nx7, got {gaussians2d.shape}"
    assert features.ndim == 2 and features.shape[0] == gaussians2d.shape[0], \
        f"Size mismatch: got {gaussians2d.shape}, {features.shape}"
    ts = config.tile_size
    tiles = (-(-int(image_size[0]) // ts)) * (-(-int(image_size[1]) // ts))
    assert tile_overlap_ranges.ndim == 2 and tile_overlap_ranges.shape == (tiles, 2), \
        f"tile_overlap_ranges must be ({tiles}, 2) for image size {tuple(image_size)}, got {tuple(tile_overlap_ranges.shape)}"


def rasterize_with_tiles(gaussians2d: torch.Tensor, features: torch.Tensor, overlap_to_point: torch.Tensor,
                         tile_overlap_ranges: torch.Tensor, image_size: Tuple[Integral, Integral],
                         config: RasterConfig) -> RasterOut:
    """Rasterize an image given 2d gaussians, features and tile overlap information.

    Parameters:
        gaussians2d: (N, 7)  packed gaussians
        features: (N, F)
        tile_overlap_ranges: (TH * TW, 2) maps tile index to a range of overlap indices
        overlap_to_point: (K, ) maps overlap index to point index
        image_size: (width, height)
        config: RasterConfig

    Returns RasterOut(image (H,W,F), image_weight (H,W), point_heuristic (N,2), visibility (N,))
    """
    _validate(gaussians2d, features, overlap_to_point, tile_overlap_ranges, image_size, config)
    image, image_weight, point_heuristic, visibility = _RasterFunction.apply(
        gaussians2d, features, overlap_to_point, tile_overlap_ranges, image_size, config)
    return RasterOut(image, image_weight, point_heuristic, visibility)


def rasterize(gaussians2d: torch.Tensor, depth: torch.Tensor, features: torch.Tensor,
              image_size: Tuple[Integral, Integral], config: RasterConfig, use_depth16: bool = False) -> RasterOut:
    """Rasterize an image given 2d gaussians, depths (for sorting) and features."""
    assert gaussians2d.shape[0] == depth.shape[0] == features.shape[0], \
        f"Size mismatch: got {gaussians2d.shape}, {depth.shape}, {features.shape}"
    overlap_to_point, tile_overlap_ranges = map_to_tiles(
        gaussians2d, depth, image_size=image_size, config=config, use_depth16=use_depth16)
    return rasterize_with_tiles(gaussians2d, features, tile_overlap_ranges=tile_overlap_ranges.view(-1, 2),
                                overlap_to_point=overlap_to_point, image_size=image_size, config=config)
