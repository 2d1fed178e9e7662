"""project_to_image forward and backward (reference benchmarks/bench_projection.py: same flags, same phases)."""
from __future__ import annotations

import argparse

import torch

from ..data_types import RasterConfig
from ..perspective import projection
from ..scenes import random_3d_gaussians, random_camera
from .util import benchmarked, image_size_arg


def parse_args(args=None):
    p = argparse.ArgumentParser()
    p.add_argument("--profile", action="store_true")
    p.add_argument("--image_size", type=str, default="1024,768")
    p.add_argument("--device", type=str, default="cuda:0")
    p.add_argument("--n", type=int, default=2000000)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--iters", type=int, default=1000)
    p.add_argument("--margin", type=float, default=0.5, help="controls random points (non visible) margin")
    p.add_argument("--debug", action="store_true")
    ns = p.parse_args(args)
    ns.image_size = image_size_arg(ns.image_size)
    return ns


def bench_projection(args):
    torch.manual_seed(args.seed)
    results = {}
    camera = random_camera(image_size=args.image_size)
    gaussians = random_3d_gaussians(args.n, camera, margin=args.margin).to(args.device)
    camera = camera.to(device=args.device)
    config = RasterConfig()
    with torch.no_grad():
        _, _, visible = projection.project_to_image(gaussians, camera, config)
        print(args)
        print(f"benchmarking {args.n} points ({visible.shape[0]} visible) points")
        results["forward"] = benchmarked("forward", lambda: projection.project_to_image(gaussians, camera, config),
                                         profile=args.profile, iters=args.iters)

    def backward():
        for t in (*gaussians.shape_tensors(), camera.T_camera_world, camera.projection):
            t.grad = None
        points, depth, _ = projection.project_to_image(gaussians, camera, config)
        (points.sum() + depth.sum()).backward()

    gaussians.requires_grad_(True)
    results["backward (gaussians)"] = benchmarked("backward (gaussians)", backward, profile=args.profile,
                                                  iters=args.iters)
    for name, extrinsics, intrinsics, splats in (("backward (extrinsics)", True, False, False),
                                                 ("backward (intrinsics)", False, True, False),
                                                 ("backward (everything)", True, True, True)):
        gaussians.requires_grad_(splats)
        camera.T_camera_world.requires_grad_(extrinsics)
        camera.projection.requires_grad_(intrinsics)
        results[name] = benchmarked(name, backward, profile=args.profile, iters=args.iters)
    return results


def main():
    return bench_projection(parse_args())


if __name__ == "__main__":
    main()
