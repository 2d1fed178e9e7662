"""Shared pieces of the operator benchmarks: the flag table (the reference's CLI flags, benchmarks/bench_*.py),
the timing harness (reference benchmarks/util.py:7-50 `benchmarked`) and a small phase recorder."""
from __future__ import annotations

import argparse
from typing import Callable, Dict, Iterable

import torch

# flag -> argparse keywords; each benchmark lists the flags it takes and may override defaults
FLAGS = {
    "profile": dict(action="store_true", help="print the torch profiler's kernel table instead of a rate"),
    "debug": dict(action="store_true", help="accepted for compatibility; there is no debug build to switch to"),
    "image_size": dict(type=str, default="1024,768", help="width,height"),
    "device": dict(type=str, default="cuda:0"),
    "n": dict(type=int, default=1000000, help="number of Gaussians"),
    "seed": dict(type=int, default=0),
    "iters": dict(type=int, default=1000),
    "margin": dict(type=float, default=0.5, help="controls random points (non visible) margin"),
    "degree": dict(type=int, default=3, help="SH degree"),
    "scale_factor": dict(type=float, default=2),
    "tile_size": dict(type=int, default=16),
    "depth16": dict(action="store_true"),
    "num_channels": dict(type=int, default=3),
    "antialias": dict(action="store_true"),
    "skip_forward": dict(action="store_true"),
    "saturate_threshold": dict(type=float, default=0.9999),
    "alpha_threshold": dict(type=float, default=1 / 255),
    "pixel_stride": dict(type=str, default="2,2"),
}
_PAIRS = ("image_size", "pixel_stride")


def int_pair(text: str):
    a, b = (int(x) for x in text.split(","))
    return a, b


def make_parser(flags: Iterable[str], **defaults) -> Callable:
    """parse_args(args=None) for a benchmark taking `flags`, with per-benchmark default overrides"""
    flags = tuple(flags)

    def parse_args(args=None):
        parser = argparse.ArgumentParser()
        for name in flags:
            options = dict(FLAGS[name])
            if name in defaults:
                options["default"] = defaults[name]
            parser.add_argument(f"--{name}", **options)
        ns = parser.parse_args(args)
        for name in _PAIRS:
            if name in flags:
                setattr(ns, name, int_pair(getattr(ns, name)))
        return ns

    return parse_args


def timed_benchmark(name: str, f, iters: int = 100, warmup: int = 10) -> float:
    for _ in range(warmup):
        f()
    first, last = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    first.record()
    for _ in range(iters):
        f()
    last.record()
    torch.cuda.synchronize()
    seconds = first.elapsed_time(last) * 1e-3
    print(f"{name}  {iters} iterations in {seconds:.3f}s at {iters / max(seconds, 1e-12):.1f} iters/sec "
          f"({1e3 * seconds / iters:.4f} ms each)")
    return 1e3 * seconds / iters


def profiled_benchmark(name: str, f, iters: int = 100, warmup: int = 1) -> float:
    from torch.profiler import ProfilerActivity, profile
    for _ in range(warmup):
        f()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(iters):
            f()
        torch.cuda.synchronize()
    print(name)
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=100))
    return float("nan")


def benchmarked(name: str, f, iters: int = 100, warmup: int = 10, profile: bool = False) -> float:
    return profiled_benchmark(name, f, iters, min(warmup, 1)) if profile else timed_benchmark(name, f, iters, warmup)


class Phases:
    """runs named phases with one set of options and keeps {name: ms per call}"""

    def __init__(self, args):
        self.args, self.results = args, {}

    def run(self, name: str, f, iters_scale: int = 1) -> None:
        self.results[name] = benchmarked(name, f, iters=self.args.iters * iters_scale, profile=self.args.profile)


def clear_grads(*tensors) -> None:
    for t in tensors:
        t.grad = None


def overlap_statistics(tile_ranges: torch.Tensor, n: int) -> Dict[str, float]:
    per_tile = (tile_ranges[..., 1] - tile_ranges[..., 0]).float()
    return dict(point_overlap=float(per_tile.sum()) / n, tile_points=float(per_tile.mean()))
