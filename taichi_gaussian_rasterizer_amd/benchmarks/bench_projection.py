"""Projection operator: forward, and backward with respect to the Gaussians / the extrinsics / the intrinsics / all
(the phases of the reference's benchmarks/bench_projection.py)."""
from __future__ import annotations

import torch

from ..data_types import RasterConfig
from ..perspective import projection
from ..scenes import random_3d_gaussians, random_camera
from .util import Phases, clear_grads, make_parser

parse_args = make_parser(("profile", "image_size", "device", "n", "seed", "iters", "margin", "debug"), n=2000000)


def bench_projection(args):
    torch.manual_seed(args.seed)
    camera = random_camera(image_size=args.image_size)
    scene = random_3d_gaussians(args.n, camera, margin=args.margin).to(args.device)
    camera = camera.to(device=args.device)
    config, phases = RasterConfig(), Phases(args)
    leaves = (*scene.shape_tensors(), camera.T_camera_world, camera.projection)

    def project():
        return projection.project_to_image(scene, camera, config)

    def project_and_differentiate():
        clear_grads(*leaves)
        points, depth, _ = project()
        (points.sum() + depth.sum()).backward()

    with torch.no_grad():
        print(args)
        print(f"benchmarking {args.n} points ({project()[2].shape[0]} visible) points")
        phases.run("forward", project)
    #             phase                     gaussians  extrinsics  intrinsics
    for name, wrt in (("backward (gaussians)", (True, False, False)), ("backward (extrinsics)", (False, True, False)),
                      ("backward (intrinsics)", (False, False, True)), ("backward (everything)", (True, True, True))):
        scene.requires_grad_(wrt[0])
        camera.T_camera_world.requires_grad_(wrt[1])
        camera.projection.requires_grad_(wrt[2])
        phases.run(name, project_and_differentiate)
    return phases.results


def main():
    return bench_projection(parse_args())


if __name__ == "__main__":
    main()
