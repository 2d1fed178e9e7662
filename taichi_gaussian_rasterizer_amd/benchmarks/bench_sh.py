"""Spherical-harmonics colour operator: forward, backward to the coefficients, backward to everything
(the phases of the reference's benchmarks/bench_sh.py)."""
from __future__ import annotations

import torch

from ..spherical_harmonics import evaluate_sh_at
from .util import Phases, clear_grads, make_parser

parse_args = make_parser(("profile", "image_size", "device", "n", "seed", "iters", "degree", "debug"), iters=200)


def bench_sh(args):
    torch.manual_seed(args.seed)
    on = dict(device=args.device)
    coefficients = torch.randn(args.n, 3, (args.degree + 1) ** 2, **on)
    centres = torch.randn(args.n, 3, **on)
    everyone = torch.arange(args.n, **on)
    eye = torch.zeros(3, **on)
    phases = Phases(args)

    def colours():
        return evaluate_sh_at(coefficients, centres, everyone, eye)

    def colours_and_gradients():
        clear_grads(coefficients, centres, eye)
        colours().sum().backward()

    with torch.no_grad():
        phases.run("forward", colours)
    coefficients.requires_grad_(True)
    phases.run("backward (sh_features)", colours_and_gradients)
    centres.requires_grad_(True)
    eye.requires_grad_(True)
    phases.run("backward (all)", colours_and_gradients)
    return phases.results


def main():
    return bench_sh(parse_args())


if __name__ == "__main__":
    main()
