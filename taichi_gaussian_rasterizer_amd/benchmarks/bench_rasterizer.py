"""rasterize_with_tiles forward / backward variants (reference benchmarks/bench_rasterizer.py: same flags, same
phases: forward, forward_vis, backward (features | gaussians | all | compute_point_heuristic))."""
from __future__ import annotations

import argparse
from dataclasses import replace

import torch

from ..data_types import RasterConfig
from ..mapper.tile_mapper import map_to_tiles
from ..misc.renderer2d import project_gaussians2d
from ..rasterizer.function import rasterize_with_tiles
from ..scenes import random_2d_gaussians
from .util import benchmarked, image_size_arg


def parse_args(args=None):
    p = argparse.ArgumentParser()
    p.add_argument("--profile", action="store_true")
    p.add_argument("--image_size", type=str, default="1024,768")
    p.add_argument("--device", type=str, default="cuda:0")
    p.add_argument("--n", type=int, default=1000000)
    p.add_argument("--num_channels", type=int, default=3)
    p.add_argument("--scale_factor", type=int, default=4)
    p.add_argument("--tile_size", type=int, default=16)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--iters", type=int, default=1000)
    p.add_argument("--antialias", action="store_true")
    p.add_argument("--debug", action="store_true")
    p.add_argument("--skip_forward", action="store_true")
    p.add_argument("--saturate_threshold", type=float, default=0.9999)
    p.add_argument("--alpha_threshold", type=float, default=1 / 255)
    p.add_argument("--pixel_stride", type=str, default="2,2")
    ns = p.parse_args(args)
    ns.image_size = image_size_arg(ns.image_size)
    ns.pixel_stride = image_size_arg(ns.pixel_stride)
    return ns


def bench_rasterizer(args):
    torch.manual_seed(args.seed)
    results = {}
    gaussians = random_2d_gaussians(args.n, args.image_size, num_channels=args.num_channels,
                                    scale_factor=args.scale_factor, alpha_range=(0.75, 1.0),
                                    depth_range=(0.1, 100.0)).to(args.device)
    config = RasterConfig(tile_size=args.tile_size, antialias=args.antialias, pixel_stride=args.pixel_stride,
                          saturate_threshold=args.saturate_threshold, alpha_threshold=args.alpha_threshold)
    gaussians2d = project_gaussians2d(gaussians)
    features = gaussians.feature
    overlap_to_point, tile_ranges = map_to_tiles(gaussians2d, depth=gaussians.z_depth, image_size=args.image_size,
                                                 config=config)
    per_tile = tile_ranges[:, :, 1] - tile_ranges[:, :, 0]
    print(overlap_to_point.shape)
    print(f"scale_factor={args.scale_factor}, n={args.n}, tile_size={args.tile_size} "
          f"point_overlap={float(per_tile.sum()) / args.n:.2f} tile_points={float(per_tile.float().mean()):.2f}")
    print("----------------------------------------------------------")

    def raster(cfg=config):
        return rasterize_with_tiles(gaussians2d=gaussians2d, features=features,
                                    tile_overlap_ranges=tile_ranges.view(-1, 2), overlap_to_point=overlap_to_point,
                                    image_size=args.image_size, config=cfg)

    if not args.skip_forward:
        with torch.no_grad():
            results["forward"] = benchmarked("forward", raster, profile=args.profile, iters=args.iters * 4)
            vis_cfg = replace(config, compute_visibility=True)
            results["forward_vis"] = benchmarked("forward_vis", lambda: raster(vis_cfg), profile=args.profile,
                                                 iters=args.iters * 4)

    def backward(cfg=config):
        gaussians2d.grad = features.grad = None
        raster(cfg).image.sum().backward()

    for name, grad_features, grad_splats in (("backward (features)", True, False),
                                             ("backward (gaussians)", False, True), ("backward (all)", True, True)):
        features.requires_grad_(grad_features)
        gaussians2d.requires_grad_(grad_splats)
        results[name] = benchmarked(name, backward, profile=args.profile, iters=args.iters)
    heur_cfg = replace(config, compute_point_heuristic=True)
    results["backward (compute_point_heuristic)"] = benchmarked(
        "backward (compute_point_heuristic)", lambda: backward(heur_cfg), profile=args.profile, iters=args.iters)
    return results


def main():
    return bench_rasterizer(parse_args())


if __name__ == "__main__":
    main()
