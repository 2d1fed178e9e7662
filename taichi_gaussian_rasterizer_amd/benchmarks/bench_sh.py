"""evaluate_sh_at forward and backward (reference benchmarks/bench_sh.py: same flags, same phases)."""
from __future__ import annotations

import argparse

import torch

from .. import spherical_harmonics
from .util import benchmarked, image_size_arg


def parse_args(args=None):
    p = argparse.ArgumentParser()
    p.add_argument("--profile", action="store_true")
    p.add_argument("--image_size", type=str, default="1024,768")
    p.add_argument("--device", type=str, default="cuda:0")
    p.add_argument("--n", type=int, default=1000000)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--iters", type=int, default=200)
    p.add_argument("--degree", type=int, default=3)
    p.add_argument("--debug", action="store_true")
    ns = p.parse_args(args)
    ns.image_size = image_size_arg(ns.image_size)
    return ns


def bench_sh(args):
    torch.manual_seed(args.seed)
    results = {}
    sh_features = torch.randn(args.n, 3, (args.degree + 1) ** 2, device=args.device)
    points = torch.randn(args.n, 3, device=args.device)
    indexes = torch.arange(args.n, device=args.device)
    camera_pos = torch.zeros(3, device=args.device)
    with torch.no_grad():
        results["forward"] = benchmarked(
            "forward", lambda: spherical_harmonics.evaluate_sh_at(sh_features, points, indexes, camera_pos),
            profile=args.profile, iters=args.iters)

    def backward():
        for t in (sh_features, points, camera_pos):
            t.grad = None
        spherical_harmonics.evaluate_sh_at(sh_features, points, indexes, camera_pos).sum().backward()

    sh_features.requires_grad_(True)
    results["backward (sh_features)"] = benchmarked("backward (sh_features)", backward, profile=args.profile,
                                                    iters=args.iters)
    points.requires_grad_(True)
    camera_pos.requires_grad_(True)
    results["backward (all)"] = benchmarked("backward (all)", backward, profile=args.profile, iters=args.iters)
    return results


def main():
    return bench_sh(parse_args())


if __name__ == "__main__":
    main()
