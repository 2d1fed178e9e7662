"""Tile mapper operator on random 2D splats (the reference's benchmarks/bench_tilemapper.py; prints the same
overlap statistics before timing)."""
from __future__ import annotations

import torch

from ..data_types import RasterConfig
from ..mapper import tile_mapper
from ..misc.renderer2d import project_gaussians2d
from ..scenes import random_2d_gaussians
from .util import Phases, make_parser, overlap_statistics

parse_args = make_parser(("profile", "image_size", "device", "n", "scale_factor", "tile_size", "seed", "iters",
                          "debug", "depth16"))


def bench_tilemapper(args):
    torch.manual_seed(args.seed)
    scene = random_2d_gaussians(args.n, args.image_size, scale_factor=args.scale_factor, alpha_range=(0.5, 1.0),
                                depth_range=(0.1, 100.0)).to(args.device)
    splats = project_gaussians2d(scene)
    options = dict(depth=scene.z_depth, image_size=args.image_size, config=RasterConfig(tile_size=args.tile_size),
                   use_depth16=args.depth16)

    def mapper():
        return tile_mapper.map_to_tiles(splats, **options)

    stats = overlap_statistics(mapper()[1], args.n)
    print(f"tile_mapper: scale_factor={args.scale_factor}, n={args.n}, tile_size={args.tile_size} "
          f"point_overlap={stats['point_overlap']:.2f} tile_points={stats['tile_points']:.2f}")
    phases = Phases(args)
    phases.run("tile_mapper", mapper)
    print("-" * 58)
    return phases.results


def main():
    return bench_tilemapper(parse_args())


if __name__ == "__main__":
    main()
