import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, taichi_gaussian_rasterizer_amd as gs
from taichi_gaussian_rasterizer_amd import RasterConfig, scenes
for size, n in (((8192, 4096), 2_000_000), ((16000, 300), 300_000), ((300, 9000), 300_000)):
    g, cam = scenes.benchmark_scene(n, size, sh_degree=1, seed=0)
    gg = g.to('cuda:0').requires_grad_(True)
    r = gs.render_gaussians(gg, cam.to(device='cuda:0'), RasterConfig(), use_sh=True)
    r.image.sum().backward()
    torch.cuda.synchronize()
    print(size, n, "visible", r.points_in_view.shape[0], "image mean", float(r.image.mean()), "grad finite", bool(torch.isfinite(gg.position.grad).all()))
