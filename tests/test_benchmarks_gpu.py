"""The per-operator benchmarks run (mirror of the reference's tests/test_benchmarks.py, at sizes that take seconds)."""
import math

import pytest

pytestmark = pytest.mark.gpu


def _ok(results, names):
    assert set(names) <= set(results), results
    assert all(math.isfinite(v) and v > 0 for v in results.values()), results


def test_bench_rasterizer():
    from taichi_gaussian_rasterizer_amd.benchmarks import bench_rasterizer
    args = bench_rasterizer.parse_args(["--n", "50000", "--iters", "5", "--image_size", "512,384"])
    _ok(bench_rasterizer.bench_rasterizer(args),
        ["forward", "forward_vis", "backward (features)", "backward (gaussians)", "backward (all)",
         "backward (compute_point_heuristic)"])


def test_bench_sh():
    from taichi_gaussian_rasterizer_amd.benchmarks import bench_sh
    _ok(bench_sh.bench_sh(bench_sh.parse_args(["--n", "100000", "--iters", "5"])),
        ["forward", "backward (sh_features)", "backward (all)"])


def test_bench_tilemapper():
    from taichi_gaussian_rasterizer_amd.benchmarks import bench_tilemapper
    _ok(bench_tilemapper.bench_tilemapper(bench_tilemapper.parse_args(["--n", "100000", "--iters", "5"])),
        ["tile_mapper"])


def test_bench_projection():
    from taichi_gaussian_rasterizer_amd.benchmarks import bench_projection
    _ok(bench_projection.bench_projection(bench_projection.parse_args(["--n", "200000", "--iters", "5"])),
        ["forward", "backward (gaussians)", "backward (extrinsics)", "backward (intrinsics)", "backward (everything)"])
