"""Multi-GPU rendering: screen tiles shard across ranks, one RCCL all-reduce of per-Gaussian
gradients (the reference has no distributed code at all).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Every rank holds the
full, replicated Gaussian set and for each frame:

  1. projects all N Gaussians and evaluates SH colour (replicated: 236 B/Gaussian read, cheaper than
     exchanging projected splats, and it keeps the visible set identical everywhere);
  2. owns a contiguous strip of tile rows.  The strip is rendered as an ordinary (W, strip_h) image
     after shifting the projected means by the strip origin (an exact f32 subtraction: the origin is a
     multiple of the tile size), so the unmodified mapper / rasterizer kernels are reused;
  3. evaluates the loss on its strip; the rasterizer backward yields PARTIAL gradients for the
     projected splats (V,7), their features (V,C) and depths (V,1);
  4. all-reduces those partial gradients -- ONE collective of 4*(7+C+1)*V bytes
     (40 B/Gaussian, vs 236 B/Gaussian if the final parameter gradients were reduced instead);
  5. runs SH / projection backward redundantly, so every rank ends with identical, complete
     parameter gradients (drop-in for a replicated optimizer).

The stage operators are injectable (`ops`) so the sharding + collective logic is exercised on CPU
with the gloo backend in tests/ (stage callables backed by the CPU oracle there).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional, Tuple

import torch
import torch.distributed as dist

from .data_types import Gaussians3D, RasterConfig
from .perspective.params import CameraParams


def tile_rows(image_height: int, tile_size: int) -> int:
    return -(-int(image_height) // tile_size)


def strip_rows(rank: int, world: int, num_tile_rows: int) -> Tuple[int, int]:
    """Contiguous, balanced split of tile rows: rank r owns rows [start, end)."""
    base, rem = divmod(num_tile_rows, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def strip_pixels(rank: int, world: int, image_height: int, tile_size: int) -> Tuple[int, int]:
    r0, r1 = strip_rows(rank, world, tile_rows(image_height, tile_size))
    return min(r0 * tile_size, image_height), min(r1 * tile_size, image_height)


class _AllReduceGrads(torch.autograd.Function):
    """Identity in the forward; in the backward the gradients of all inputs are packed into one
    buffer and summed over the process group with a single all-reduce."""

    @staticmethod
    def forward(ctx, group, *tensors):
        ctx.group = group
        ctx.shapes = [t.shape for t in tensors]
        return tuple(t.view_as(t) for t in tensors)

    @staticmethod
    def backward(ctx, *grads):
        ref = next(g for g in grads if g is not None)
        rows = ref.shape[0]
        widths = [int(torch.Size(s[1:]).numel()) if len(s) > 1 else 1 for s in ctx.shapes]
        packed = torch.zeros((rows, sum(widths)), dtype=ref.dtype, device=ref.device)
        col = 0
        for g, w in zip(grads, widths):
            if g is not None:
                packed[:, col:col + w] = g.reshape(rows, w)
            col += w
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(ctx.group) > 1:
            dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=ctx.group)
        out, col = [], 0
        for s, w in zip(ctx.shapes, widths):
            out.append(packed[:, col:col + w].reshape(s).contiguous())
            col += w
        return (None, *out)


def default_ops() -> SimpleNamespace:
    """The HIP operators of this package."""
    from .mapper.tile_mapper import map_to_tiles
    from .perspective.projection import project_with_ndc
    from .rasterizer.function import rasterize_with_tiles
    from .spherical_harmonics import evaluate_sh_at
    return SimpleNamespace(project_with_ndc=project_with_ndc, evaluate_sh_at=evaluate_sh_at,
                           map_to_tiles=map_to_tiles, rasterize_with_tiles=rasterize_with_tiles)


def render_gaussians_sharded(gaussians: Gaussians3D, camera_params: CameraParams,
                             config: RasterConfig = RasterConfig(), use_sh: bool = False, render_depth: bool = False,
                             use_depth16: bool = False, group=None, rank: Optional[int] = None,
                             world_size: Optional[int] = None, ops: Optional[SimpleNamespace] = None):
    """Render this rank's strip of the frame.  Returns a `Rendering` whose image tensors cover rows
    [y0, y1) of the full image (`rendering.strip == (y0, y1)`); after `.backward()` of a loss summed
    over strips, every rank holds the full parameter gradients.  `point_visibility` / `point_heuristic` are this
    strip's share: `reduce_point_statistics` sums them over the ranks."""
    from .renderer import Rendering, compute_depth_variance
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size(group) if dist.is_initialized() else 1
    w, h = (int(x) for x in camera_params.image_size)
    ts = config.tile_size
    y0, y1 = strip_pixels(rank, world_size, h, ts)

    if ops is None:
        from .fused import fused_supported, render_fused
        if y1 > y0 and fused_supported(gaussians, camera_params, use_sh, False):
            # the fused frame (fused.py) on this rank's strip; its backward carries the all-reduce
            r = render_fused(gaussians, camera_params, config, render_depth, use_depth16, strip=(y0, y1), group=group)
            object.__setattr__(r, "strip", (y0, y1))
            return r
        ops = default_ops()

    gaussians2d, depths, indexes, ndc_depths = ops.project_with_ndc(
        *gaussians.shape_tensors(), camera_params.T_camera_world, camera_params.projection,
        camera_params.image_size, camera_params.depth_range, config)
    if use_sh:
        features = ops.evaluate_sh_at(gaussians.feature, gaussians.position.detach(), indexes,
                                      camera_params.camera_position)
    else:
        features = gaussians.feature[indexes]

    # everything upstream of this point is replicated; gradients arriving here are partial sums
    g2d_r, features_r, depths_r = _AllReduceGrads.apply(group, gaussians2d, features, depths)

    strip_h = y1 - y0
    if strip_h <= 0:  # more ranks than tile rows: this rank renders nothing but still joins the collective
        zero = (g2d_r.sum() + features_r.sum() + depths_r.sum()) * 0.0
        image = torch.zeros((0, w, features.shape[1]), dtype=features.dtype, device=features.device) + zero
        r = Rendering(image=image, image_weight=image[..., 0].detach(), camera=camera_params, config=config,
                      points_in_view=indexes, point_depth=depths, gaussians2d=gaussians2d)
        object.__setattr__(r, "strip", (y0, y1))
        return r

    shift = torch.zeros((7,), dtype=g2d_r.dtype, device=g2d_r.device)
    shift[1] = float(y0)
    local2d = g2d_r - shift
    raster_features = torch.cat([depths_r, depths_r ** 2, features_r], dim=1) if render_depth else features_r

    overlap_to_point, ranges = ops.map_to_tiles(local2d, ndc_depths, image_size=(w, strip_h), config=config,
                                                use_depth16=use_depth16)
    raster = ops.rasterize_with_tiles(local2d, raster_features, tile_overlap_ranges=ranges.view(-1, 2),
                                      overlap_to_point=overlap_to_point, image_size=(w, strip_h), config=config)
    image, img_depth, img_var = raster.image, None, None
    if render_depth:
        img_depth, img_var = compute_depth_variance(image[..., :2], raster.image_weight)
        image = image[..., 2:]
    r = Rendering(image=image, image_weight=raster.image_weight, depth=img_depth, depth_var=img_var,
                  camera=camera_params, config=config,
                  point_visibility=raster.visibility if config.compute_visibility else None,
                  point_heuristic=raster.point_heuristic if config.compute_point_heuristic else None,
                  points_in_view=indexes, point_depth=depths, gaussians2d=gaussians2d)
    object.__setattr__(r, "strip", (y0, y1))
    return r


def reduce_point_statistics(rendering, group=None):
    """`point_visibility` / `point_heuristic` of a sharded rendering cover this rank's strip only (they are sums
    over pixels).  Returns (visibility, heuristic) summed over the ranks -- one small all-reduce of (V,3) floats,
    to be called after `.backward()` (the heuristic is filled in by the backward pass); None where not computed."""
    vis, heur = rendering.point_visibility, rendering.point_heuristic
    parts = [t.reshape(t.shape[0], -1) for t in (vis, heur) if t is not None]
    if not parts:
        return None, None
    packed = torch.cat(parts, dim=1).contiguous()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    out_vis = packed[:, 0].contiguous() if vis is not None else None
    out_heur = packed[:, (1 if vis is not None else 0):].contiguous() if heur is not None else None
    return out_vis, out_heur


def gather_image(strip_image: torch.Tensor, image_height: int, tile_size: int, group=None) -> torch.Tensor:
    """All-gather the strips into the full (H, W, C) image on every rank (only when a caller needs it)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return strip_image
    heights = [strip_pixels(r, world, image_height, tile_size) for r in range(world)]
    max_h = max(b - a for a, b in heights)
    pad = torch.zeros((max_h, *strip_image.shape[1:]), dtype=strip_image.dtype, device=strip_image.device)
    pad[:strip_image.shape[0]] = strip_image
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:b - a] for p, (a, b) in zip(parts, heights)], dim=0)
