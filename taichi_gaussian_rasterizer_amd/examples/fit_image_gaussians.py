"""Fit 2D Gaussians to an image with pruning and splitting (the workflow of the reference's
examples/fit_image_gaussians.py: BASELINE config 1), on the HIP rasterizer and this package's optimizers.

    python -m taichi_gaussian_rasterizer_amd.examples.fit_image_gaussians [image] --n 1000 --target 4000 --iters 800

Without an image file (or without PIL to read it) a synthetic smooth test pattern is fitted; `--write_frames DIR`
saves the rendering after every epoch as PNG when PIL is present.  Each epoch trains, then removes the cheapest
Gaussians (prune cost from the rasterizer's backward) and splits the ones with the highest split score until the
population reaches `--target`; `ParameterClass` carries the optimizer state across the surgery.
"""
from __future__ import annotations

import argparse
import math
import time
from pathlib import Path

import torch

from .. import Gaussians2D, RasterConfig, rasterize
from ..misc.renderer2d import point_basis, project_gaussians2d, uniform_split_gaussians2d
from ..optim import ParameterClass, VisibilityAwareLaProp
from ..scenes import random_2d_gaussians


def parse_args(args=None):
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    p.add_argument("image_file", type=str, nargs="?", default=None)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--tile_size", type=int, default=16)
    p.add_argument("--n", type=int, default=1000, help="initial number of Gaussians")
    p.add_argument("--target", type=int, default=None, help="population to grow to (default: stay at n)")
    p.add_argument("--iters", type=int, default=2000)
    p.add_argument("--max_lr", type=float, default=0.5)
    p.add_argument("--min_lr", type=float, default=0.1)
    p.add_argument("--epoch", type=int, default=8, help="iterations of the first epoch (epochs grow to --max_epoch)")
    p.add_argument("--max_epoch", type=int, default=32)
    p.add_argument("--prune_rate", type=float, default=0.025, help="fraction pruned per epoch, decaying to 0")
    p.add_argument("--opacity_reg", type=float, default=0.00001)
    p.add_argument("--scale_reg", type=float, default=0.1)
    p.add_argument("--antialias", action="store_true")
    p.add_argument("--size", type=str, default="256,256", help="size of the synthetic target when no image is given")
    p.add_argument("--device", type=str, default="cuda:0")
    p.add_argument("--write_frames", type=Path, default=None)
    return p.parse_args(args)


def load_target(args) -> torch.Tensor:
    """(H, W, 3) float image in [0, 1] on the device"""
    if args.image_file is not None:
        try:
            from PIL import Image
            import numpy as np
            pixels = np.asarray(Image.open(args.image_file).convert("RGB"), dtype="float32") / 255.0
            return torch.from_numpy(pixels).to(args.device)
        except ImportError:
            print("PIL is not available: fitting the synthetic pattern instead")
    w, h = (int(v) for v in args.size.split(","))
    ys, xs = torch.meshgrid(torch.linspace(0, 1, h), torch.linspace(0, 1, w), indexing="ij")
    waves = [0.5 + 0.5 * torch.sin(2 * math.pi * (fx * xs + fy * ys) + ph)
             for fx, fy, ph in ((3, 1, 0.0), (1, 4, 1.0), (5, 5, 2.0))]
    disc = ((xs - 0.6) ** 2 + (ys - 0.4) ** 2 < 0.04).float()
    return (torch.stack(waves, dim=-1) * (0.6 + 0.4 * disc.unsqueeze(-1))).to(args.device)


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    return float(10 * torch.log10(1 / torch.nn.functional.mse_loss(a, b)))


def epoch_sizes(total: int, first: int, last: int):
    """geometric growth from `first` to `last` iterations per epoch, the final epoch absorbing the remainder"""
    done, sizes = 0, []
    while done < total:
        t = done / total
        size = math.ceil(math.exp(math.log(last) * t + math.log(first) * (1 - t)))
        if done + 2 * size > total:
            size = total - done
        sizes.append(size)
        done += size
    return sizes


def as_gaussians(params: ParameterClass) -> Gaussians2D:
    return Gaussians2D(**{name: params[name] for name in params.keys()}, batch_size=tuple(params.batch_size))


def train_epoch(params: ParameterClass, target: torch.Tensor, config: RasterConfig, iters: int, opacity_reg: float,
                scale_reg: float):
    h, w = target.shape[:2]
    raster = None
    for _ in range(iters):
        params.zero_grad()
        gaussians = as_gaussians(params)
        raster = rasterize(project_gaussians2d(gaussians), gaussians.z_depth.clamp(0, 1), gaussians.feature, (w, h),
                           config)
        size = torch.exp(gaussians.log_scaling) / min(w, h)
        loss = (torch.nn.functional.mse_loss(raster.image.sigmoid(), target)
                + opacity_reg * gaussians.opacity.mean() + scale_reg * size.pow(2).mean())
        loss.backward()
        seen = (raster.visibility > 1e-8).nonzero().squeeze(1)
        params.step(indexes=seen, visibility=raster.visibility[seen], basis=point_basis(gaussians[seen]))
        params = params.replace(rotation=torch.nn.functional.normalize(params.rotation.detach()),
                                log_scaling=params.log_scaling.detach().clamp(min=-1.0, max=4.0))
    return params, raster


def top_mask(values: torch.Tensor, count: int, largest: bool) -> torch.Tensor:
    mask = torch.zeros_like(values, dtype=torch.bool)
    if count > 0:
        mask[torch.topk(values, k=min(count, values.shape[0]), largest=largest).indices] = True
    return mask


def split_and_prune(params: ParameterClass, progress: float, target: int, prune_rate: float, raster):
    """drop the Gaussians that cost least to remove, split the ones the loss pulls on hardest"""
    count = int(params.batch_size[0])
    prune_cost, split_score = raster.point_heuristic[:, 0], raster.point_heuristic[:, 1]
    prune = top_mask(prune_cost, int(prune_rate * count * (1 - progress)), largest=False)
    split = top_mask(split_score, max(0, target - count + int(prune.sum())), largest=True) & ~prune
    parents = Gaussians2D(**{name: params[name].detach()[split] for name in params.keys()},
                          batch_size=(int(split.sum()),))
    survivors = params[(~(split | prune)).nonzero().squeeze(1)]
    if int(split.sum()) > 0:
        children = uniform_split_gaussians2d(parents, random_axis=True)
        survivors = survivors.append_tensors(dict(children.items()))
    return survivors, dict(split=int(split.sum()), prune=int(prune.sum()))


def main(args=None):
    args = parse_args(args)
    torch.manual_seed(args.seed)
    target = load_target(args)
    h, w = target.shape[:2]
    goal = args.target or args.n
    scene = random_2d_gaussians(args.n, (w, h), alpha_range=(0.5, 1.0), scale_factor=0.5).to(args.device)
    groups = dict(position=dict(lr=args.max_lr, type="local_vector"), log_scaling=dict(lr=0.1), rotation=dict(lr=1.0),
                  alpha_logit=dict(lr=0.1), feature=dict(lr=0.1, type="vector"))
    params = ParameterClass(dict(scene.items()), groups, optimizer=VisibilityAwareLaProp, vis_smooth=0.1, vis_beta=0.8,
                            betas=(0.9, 0.9), eps=1e-16, bias_correction=True)
    config = RasterConfig(compute_point_heuristic=True, compute_visibility=True, tile_size=args.tile_size,
                          blur_cov=0.0 if args.antialias else 0.3, antialias=args.antialias)
    sizes = epoch_sizes(args.iters, args.epoch, args.max_epoch)
    done, history = 0, []
    for number, size in enumerate(sizes):
        progress = done / max(args.iters, 1)
        lr = math.exp(math.log(args.min_lr) * progress + math.log(args.max_lr) * (1 - progress))
        params.set_learning_rate(position=lr)
        torch.cuda.synchronize()
        started = time.perf_counter()
        params, raster = train_epoch(params, target, config, size, args.opacity_reg, args.scale_reg)
        torch.cuda.synchronize()
        rate = size / (time.perf_counter() - started)
        quality = psnr(raster.image.sigmoid().detach(), target)
        done += size
        surgery = dict(split=0, prune=0)
        if number + 1 < len(sizes):
            params, surgery = split_and_prune(params, done / args.iters, goal, args.prune_rate, raster)
        history.append(quality)
        print(f"epoch {number:3d}: {done:5d} iterations, n={int(params.batch_size[0]):6d} psnr={quality:6.2f} dB "
              f"{rate:7.1f} it/s split={surgery['split']} prune={surgery['prune']}")
        if args.write_frames is not None:
            try:
                from PIL import Image
                args.write_frames.mkdir(parents=True, exist_ok=True)
                frame = (raster.image.sigmoid().detach().clamp(0, 1) * 255).to(torch.uint8).cpu().numpy()
                Image.fromarray(frame).save(args.write_frames / f"{number:04d}.png")
            except ImportError:
                pass
    return history


if __name__ == "__main__":
    main()
