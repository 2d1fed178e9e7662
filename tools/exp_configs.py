"""C3 frame under other RasterConfig settings (tile size, antialias, visibility + heuristics, depth16)"""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import taichi_gaussian_rasterizer_amd as gs
from taichi_gaussian_rasterizer_amd import RasterConfig, scenes, _native as nv
n, size = 1_000_000, (2048, 2048)
g, cam = scenes.benchmark_scene(n, size, sh_degree=3, seed=0)
cam = cam.to(device='cuda:0')
G = torch.rand(size[1], size[0], 3, device='cuda:0')
gg = g.to('cuda:0').requires_grad_(True)
cases = [("default", RasterConfig(), {}), ("tile 8", RasterConfig(tile_size=8), {}), ("tile 32", RasterConfig(tile_size=32), {}),
         ("antialias", RasterConfig(antialias=True, blur_cov=0.0), {}),
         ("visibility+heuristics", RasterConfig(compute_visibility=True, compute_point_heuristic=True), {}),
         ("depth16", RasterConfig(), dict(use_depth16=True)), ("render_depth", RasterConfig(), dict(render_depth=True)),
         ("median_depth", RasterConfig(), dict(render_median_depth=True))]
for name, cfg, kw in cases:
    def step():
        for _, t in gg.items(): t.grad = None
        r = gs.render_gaussians(gg, cam, cfg, use_sh=True, **kw)
        r.image.backward(G)
    for _ in range(4): step()
    torch.cuda.synchronize()
    nv.timer.reset(); nv.timer.only = None; nv.timer.enabled = True
    for _ in range(8): step()
    torch.cuda.synchronize(); nv.timer.enabled = False
    st = {k[3:]: round(v[1] / 8, 3) for k, v in nv.timer.summary().items()}  # raster_fwd of median_depth = both passes
    print(f"{name:22s} total {sum(st.values()):.3f} ms  {st}", flush=True)
