"""map_to_tiles (reference benchmarks/bench_tilemapper.py: same flags; prints the overlap statistics it prints)."""
from __future__ import annotations

import argparse

import torch

from ..data_types import RasterConfig
from ..mapper import tile_mapper
from ..misc.renderer2d import project_gaussians2d
from ..scenes import random_2d_gaussians
from .util import benchmarked, image_size_arg


def parse_args(args=None):
    p = argparse.ArgumentParser()
    p.add_argument("--profile", action="store_true")
    p.add_argument("--image_size", type=str, default="1024,768")
    p.add_argument("--device", type=str, default="cuda:0")
    p.add_argument("--n", type=int, default=1000000)
    p.add_argument("--scale_factor", type=float, default=2)
    p.add_argument("--tile_size", type=int, default=16)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--iters", type=int, default=1000)
    p.add_argument("--debug", action="store_true")
    p.add_argument("--depth16", action="store_true")
    ns = p.parse_args(args)
    ns.image_size = image_size_arg(ns.image_size)
    return ns


def bench_tilemapper(args):
    torch.manual_seed(args.seed)
    gaussians = random_2d_gaussians(args.n, args.image_size, scale_factor=args.scale_factor, alpha_range=(0.5, 1.0),
                                    depth_range=(0.1, 100.0)).to(args.device)
    config = RasterConfig(tile_size=args.tile_size)
    gaussians2d = project_gaussians2d(gaussians)

    def map_to_tiles():
        return tile_mapper.map_to_tiles(gaussians2d, depth=gaussians.z_depth, image_size=args.image_size,
                                        config=config, use_depth16=args.depth16)

    _, tile_ranges = map_to_tiles()
    per_tile = tile_ranges[:, :, 1] - tile_ranges[:, :, 0]
    print(f"tile_mapper: scale_factor={args.scale_factor}, n={args.n}, tile_size={args.tile_size} "
          f"point_overlap={float(per_tile.sum()) / args.n:.2f} tile_points={float(per_tile.float().mean()):.2f}")
    result = {"tile_mapper": benchmarked("tile_mapper", map_to_tiles, profile=args.profile, iters=args.iters)}
    print("----------------------------------------------------------")
    return result


def main():
    return bench_tilemapper(parse_args())


if __name__ == "__main__":
    main()
