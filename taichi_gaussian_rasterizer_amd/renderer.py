"""The 3D renderer entry points and their result record.

`render_gaussians` is what a trainer calls once per view (reference renderer.py:134-171); it runs as the single fused
frame of fused.py, or -- an empty scene, more than 30 feature channels -- as the sequence
project -> features -> `render_projected` (reference renderer.py:183-231) of this file, every stage a HIP operator.
`Rendering` carries the images plus the per-splat by-products a trainer prunes and densifies with (reference
renderer.py:28-131: same field and property names).
"""
from __future__ import annotations

import dataclasses
from functools import cached_property
from numbers import Integral
from typing import Optional, Tuple

import torch

from .data_types import Gaussians3D, RasterConfig
from .mapper.tile_mapper import map_to_tiles
from .perspective.params import CameraParams
from .perspective.projection import project_with_ndc
from .rasterizer.function import rasterize_with_tiles
from .spherical_harmonics import evaluate_sh_at
from .torch_lib.projection import ndc_depth

_NEED_HEURISTIC = "No point heuristic information available (use config.compute_point_heuristic=True)"
_NEED_VISIBILITY = "No visibility information available (use config.compute_visibility=True)"


def unpack(record) -> dict:
    """field name -> value of a dataclass instance (shallow)"""
    return {f.name: getattr(record, f.name) for f in dataclasses.fields(record)}


class _SplatColumns:
    """read-only view of columns [lo, hi) of the packed projected splats (mean 0:2, axis 2:4, sigma 4:6, alpha 6)"""

    def __init__(self, lo: int, hi: Optional[int] = None):
        self.index = lo if hi is None else slice(lo, hi)

    def __get__(self, rendering, owner=None):
        return self if rendering is None else rendering.gaussians2d[:, self.index]


class _HeuristicColumn:
    """one column of point_heuristic; only there when the config asked for it"""

    def __init__(self, column: int):
        self.column = column

    def __get__(self, rendering, owner=None):
        if rendering is None:
            return self
        assert rendering.config.compute_point_heuristic, _NEED_HEURISTIC
        return rendering.point_heuristic[:, self.column]


@dataclasses.dataclass(frozen=True, kw_only=True)
class Rendering:
    """What one view produced.  `depth` / `depth_var` exist with render_depth=True, `median_depth` with
    render_median_depth=True, `point_visibility` / `point_heuristic` with the corresponding RasterConfig switches
    (the heuristic is written by the backward pass)."""
    image: torch.Tensor                              # (H, W, C) blended features
    image_weight: torch.Tensor                       # (H, W) accumulated alpha
    points_in_view: torch.Tensor                     # (V) int64: which Gaussians survived the cull
    point_depth: torch.Tensor                        # (V, 1) camera-space z of those
    point_visibility: Optional[torch.Tensor] = None  # (V) summed blend weight per splat
    point_heuristic: Optional[torch.Tensor] = None   # (V, 2) prune cost, split score
    camera: CameraParams
    config: RasterConfig
    depth: Optional[torch.Tensor] = None             # (H, W)
    depth_var: Optional[torch.Tensor] = None         # (H, W)
    median_depth: Optional[torch.Tensor] = None      # (H, W)
    gaussians2d: torch.Tensor                        # (V, 7) packed projected splats

    # ---- columns of the projected splats, statistics of the backward pass
    point_scale = _SplatColumns(4, 6)
    point_opacity = _SplatColumns(6)
    prune_cost = _HeuristicColumn(0)
    split_score = _HeuristicColumn(1)

    def _to_ndc(self, z: torch.Tensor) -> torch.Tensor:
        return ndc_depth(z, self.camera.near_plane, self.camera.far_plane)

    @cached_property
    def ndc_depth(self) -> torch.Tensor:
        return self._to_ndc(self.depth)

    @cached_property
    def ndc_median_depth(self) -> torch.Tensor:
        return self._to_ndc(self.median_depth)

    @property
    def ndc_point_depth(self) -> torch.Tensor:
        return self._to_ndc(self.point_depth)

    @property
    def gaussian_scale(self) -> torch.Tensor:
        """how many sigmas out a splat still reaches alpha_threshold: the extent the culling uses
        (the original 3DGS takes a constant 3)"""
        return (2.0 * torch.log(self.point_opacity / self.config.alpha_threshold)).sqrt()

    @property
    def point_radii(self) -> torch.Tensor:
        return torch.amax(self.point_scale, dim=1)

    @property
    def _point_visibility(self) -> torch.Tensor:
        assert self.point_visibility is not None, _NEED_VISIBILITY
        return self.point_visibility

    @cached_property
    def visible_mask(self) -> torch.Tensor:
        return self._point_visibility > 0

    @cached_property
    def visible_indices(self) -> torch.Tensor:
        return self.points_in_view[self.visible_mask]

    @cached_property
    def visible(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(indexes into the scene, visibility) of the splats that contributed to some pixel"""
        return self.visible_indices, self._point_visibility[self.visible_mask]

    @property
    def image_size(self) -> Tuple[Integral, Integral]:
        return self.camera.image_size

    @property
    def num_points(self) -> int:
        return int(self.points_in_view.shape[0])

    def detach(self) -> "Rendering":
        cut = {name: (value.detach() if hasattr(value, "detach") else value) for name, value in unpack(self).items()}
        return Rendering(**cut)


def _check_call(gaussians, camera_params, config, flags: dict) -> None:
    for value, kind, name in ((gaussians, Gaussians3D, "gaussians"), (camera_params, CameraParams, "camera_params"),
                              (config, RasterConfig, "config")):
        if not isinstance(value, kind):
            raise TypeError(f"{name} must be {kind.__name__}, got {type(value).__name__}")
    for name, value in flags.items():
        if not isinstance(value, bool):
            raise TypeError(f"{name} must be bool")


def render_gaussians(gaussians: Gaussians3D, camera_params: CameraParams, config: RasterConfig = RasterConfig(),
                     use_sh: bool = False, render_depth: bool = False, use_depth16: bool = False,
                     render_median_depth: bool = False) -> Rendering:
    """Render one view.  `gaussians.feature` holds (N, C) features, or (N, 3, (D+1)^2) SH coefficients with
    use_sh=True.  render_depth adds depth and depth variance images, render_median_depth a second,
    non-blended pass that picks the depth at half opacity, use_depth16 sorts on 16-bit depth codes."""
    _check_call(gaussians, camera_params, config, dict(use_sh=use_sh, render_depth=render_depth,
                                                      use_depth16=use_depth16,
                                                      render_median_depth=render_median_depth))
    from .fused import fused_supported, render_fused
    if fused_supported(gaussians, camera_params, use_sh, render_median_depth):
        return render_fused(gaussians, camera_params, config, render_depth, use_depth16,
                            render_median_depth=render_median_depth)

    splats, depths, visible, sort_depths = project_with_ndc(
        *gaussians.shape_tensors(), camera_params.T_camera_world, camera_params.projection,
        camera_params.image_size, camera_params.depth_range, config)
    if use_sh:  # the view direction is not differentiated through (reference renderer.py:164)
        colours = evaluate_sh_at(gaussians.feature, gaussians.position.detach(), visible,
                                 camera_params.camera_position)
    else:
        colours = gaussians.feature[visible]
        assert colours.dim() == 2, f"Features must be (N, C) if use_sh=False, got {colours.shape}"
    return render_projected(visible, splats, colours, depths, camera_params, config, render_depth=render_depth,
                            use_depth16=use_depth16, render_median_depth=render_median_depth,
                            ndc_depths=sort_depths)


def compute_depth_variance(depth_depthsq: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6):
    """(…, 2) blended [z, z^2] and the accumulated alpha -> expected depth and its variance"""
    total = weight + eps  # true divisions: bit-identical to the fused frame's gs_depth_split_fwd
    mean = depth_depthsq[..., 0] / total
    return mean, depth_depthsq[..., 1] / total - mean * mean


def render_projected(indexes: torch.Tensor, gaussians2d: torch.Tensor, features: torch.Tensor, depths: torch.Tensor,
                     camera_params: CameraParams, config: RasterConfig, render_depth: bool = False,
                     use_depth16: bool = False, render_median_depth: bool = False, use_ndc_depth: bool = False,
                     ndc_depths: Optional[torch.Tensor] = None) -> Rendering:
    """Tile-map and rasterize splats that are already projected.  `ndc_depths` (the sort depth) comes from the
    projection kernel when the caller has it; otherwise it is derived from `depths` here."""
    size = camera_params.image_size
    if ndc_depths is None:
        ndc_depths = ndc_depth(depths.detach(), camera_params.near_plane, camera_params.far_plane)
    channels = features
    raster_config = config
    if render_depth:  # two leading channels carry z and z^2 through the blend
        z = ndc_depths if use_ndc_depth else depths
        channels = torch.cat((z, z * z, features), dim=1)
        if not use_ndc_depth:
            # what the forward's early stop drops is bounded by forward_cut * max|feature|, and z^2 reaches far^2
            # (data_types.RasterConfig.forward_cut; the fused frame applies the same scale)
            raster_config = dataclasses.replace(
                config, forward_cut=config.forward_cut / max(float(camera_params.far_plane) ** 2, 1.0))

    overlap_to_point, tile_ranges = map_to_tiles(gaussians2d, ndc_depths, image_size=size, config=config,
                                                 use_depth16=use_depth16)
    tiles = dict(tile_overlap_ranges=tile_ranges.view(-1, 2), overlap_to_point=overlap_to_point, image_size=size)
    raster = rasterize_with_tiles(gaussians2d, channels, config=raster_config, **tiles)

    median = None
    if render_median_depth:  # first splat that takes a pixel past half opacity, no blending
        pick = dataclasses.replace(config, use_alpha_blending=False, saturate_threshold=0.5)
        median = rasterize_with_tiles(gaussians2d, depths, config=pick, **tiles).image.squeeze(-1)

    image, mean_z, var_z = raster.image, None, None
    if render_depth:
        mean_z, var_z = compute_depth_variance(image[..., :2], raster.image_weight)
        image = image[..., 2:]
    return Rendering(image=image, image_weight=raster.image_weight, depth=mean_z, depth_var=var_z, median_depth=median,
                     camera=camera_params, config=config, points_in_view=indexes, point_depth=depths,
                     gaussians2d=gaussians2d,
                     point_visibility=raster.visibility if config.compute_visibility else None,
                     point_heuristic=raster.point_heuristic if config.compute_point_heuristic else None)


def viewspace_gradient(gaussians2d: torch.Tensor) -> torch.Tensor:
    """length of dL/d(mean) per projected splat, the classic densification signal; needs `gaussians2d.retain_grad()`
    before the backward pass"""
    assert gaussians2d.shape[1] == 7, f"Expected packed 2D gaussians (N, 7), got {gaussians2d.shape}"
    assert gaussians2d.grad is not None, \
        "Expected gradients on gaussians2d, run backward first with gaussians2d.retain_grad()"
    return gaussians2d.grad[:, :2].norm(dim=1)
