"""Timing harness of the operator benchmarks (reference benchmarks/util.py:7-50: `benchmarked(name, f, iters,
warmup, profile)`).  `profile=True` prints the torch profiler's kernel table instead of a rate."""
from __future__ import annotations

import torch


def timed_benchmark(name: str, f, iters: int = 100, warmup: int = 10) -> float:
    for _ in range(warmup):
        f()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        f()
    end.record()
    torch.cuda.synchronize()
    seconds = start.elapsed_time(end) / 1000.0
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


def image_size_arg(text: str):
    w, h = (int(x) for x in text.split(","))
    return w, h
