"""BASELINE.json config 1: the 2D fit_image_gaussians loop, 256x256, n = 2000, on CPU -- plumbing.
The caller shape is the reference example (examples/fit_image_gaussians.py:101-123):
Gaussians2D -> project_gaussians2d -> rasterize -> sigmoid -> MSE -> backward -> step, with
compute_visibility / compute_point_heuristic on (:289-296).  Compute here is the CPU oracle behind
the package's operator signatures (tests/oracle_ops.py); the same loop runs on the GPU through the
HIP operators in tests/test_gpu_parity.py::test_config1_fit_loop_gpu."""
import torch

import oracle_ops
from taichi_gaussian_rasterizer_amd import RasterConfig, scenes
from taichi_gaussian_rasterizer_amd.misc.renderer2d import project_gaussians2d


def fit_loop(rasterize, device, steps, n=2000, size=(256, 256)):
    torch.manual_seed(0)
    w, h = size
    g = scenes.random_2d_gaussians(n, size, alpha_range=(0.5, 1.0), scale_factor=0.5).to(device)
    target = torch.rand(h, w, 3, generator=torch.Generator().manual_seed(1)).to(device)
    params = {k: v.clone().requires_grad_(True) for k, v in g.items()}
    opt = torch.optim.Adam([params[k] for k in ("position", "log_scaling", "rotation", "alpha_logit", "feature")],
                           lr=0.02)
    cfg = RasterConfig(compute_visibility=True, compute_point_heuristic=True)
    losses, last = [], None
    for _ in range(steps):
        opt.zero_grad()
        gg = type(g)(**params, batch_size=(n,))
        g2d = project_gaussians2d(gg)
        raster = rasterize(g2d, gg.z_depth.clamp(0, 1), gg.feature, size, cfg)
        loss = torch.nn.functional.mse_loss(raster.image.sigmoid(), target)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        last = raster
    return losses, last


def test_fit_image_gaussians_cpu_plumbing():
    losses, raster = fit_loop(oracle_ops.rasterize, "cpu", steps=6)
    assert raster.image.shape == (256, 256, 3) and raster.image_weight.shape == (256, 256)
    assert raster.visibility.shape == (2000,) and raster.point_heuristic.shape == (2000, 2)
    assert (raster.visibility >= 0).all() and float(raster.visibility.sum()) > 0
    assert float(raster.point_heuristic.abs().sum()) > 0
    assert all(b < a for a, b in zip(losses, losses[1:])), losses
