"""Rasterizer operator on a fixed tile mapping: forward (plain, with visibility), backward to the features / the
splats / both, backward with the densification heuristics (the phases of the reference's
benchmarks/bench_rasterizer.py)."""
from __future__ import annotations

import dataclasses

import torch

from ..data_types import RasterConfig
from ..mapper.tile_mapper import map_to_tiles
from ..misc.renderer2d import project_gaussians2d
from ..rasterizer.function import rasterize_with_tiles
from ..scenes import random_2d_gaussians
from .util import Phases, clear_grads, make_parser, overlap_statistics

parse_args = make_parser(("profile", "image_size", "device", "n", "num_channels", "scale_factor", "tile_size", "seed",
                          "iters", "antialias", "debug", "skip_forward", "saturate_threshold", "alpha_threshold",
                          "pixel_stride"), scale_factor=4)


def bench_rasterizer(args):
    torch.manual_seed(args.seed)
    scene = random_2d_gaussians(args.n, args.image_size, num_channels=args.num_channels,
                                scale_factor=args.scale_factor, alpha_range=(0.75, 1.0),
                                depth_range=(0.1, 100.0)).to(args.device)
    config = RasterConfig(tile_size=args.tile_size, antialias=args.antialias, pixel_stride=args.pixel_stride,
                          saturate_threshold=args.saturate_threshold, alpha_threshold=args.alpha_threshold)
    splats, features = project_gaussians2d(scene), scene.feature
    overlap_to_point, tile_ranges = map_to_tiles(splats, depth=scene.z_depth, image_size=args.image_size, config=config)
    stats = overlap_statistics(tile_ranges, args.n)
    print(overlap_to_point.shape)
    print(f"scale_factor={args.scale_factor}, n={args.n}, tile_size={args.tile_size} "
          f"point_overlap={stats['point_overlap']:.2f} tile_points={stats['tile_points']:.2f}")
    print("-" * 58)
    mapping = dict(tile_overlap_ranges=tile_ranges.view(-1, 2), overlap_to_point=overlap_to_point,
                   image_size=args.image_size)

    def raster(**changes):
        return rasterize_with_tiles(gaussians2d=splats, features=features,
                                    config=dataclasses.replace(config, **changes), **mapping)

    def raster_and_differentiate(**changes):
        clear_grads(splats, features)
        raster(**changes).image.sum().backward()

    phases = Phases(args)
    if not args.skip_forward:
        with torch.no_grad():
            phases.run("forward", raster, iters_scale=4)
            phases.run("forward_vis", lambda: raster(compute_visibility=True), iters_scale=4)
    for name, wrt_features, wrt_splats in (("backward (features)", True, False), ("backward (gaussians)", False, True),
                                           ("backward (all)", True, True)):
        features.requires_grad_(wrt_features)
        splats.requires_grad_(wrt_splats)
        phases.run(name, raster_and_differentiate)
    phases.run("backward (compute_point_heuristic)", lambda: raster_and_differentiate(compute_point_heuristic=True))
    return phases.results


def main():
    return bench_rasterizer(parse_args())


if __name__ == "__main__":
    main()
