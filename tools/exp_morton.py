"""experiment: does Morton-ordering the Gaussians (what a trainer does periodically) change the frame time?"""
import sys, time
sys.path.insert(0, '.')
import torch
import taichi_gaussian_rasterizer_amd as gs
from taichi_gaussian_rasterizer_amd import RasterConfig, scenes
from taichi_gaussian_rasterizer_amd.misc import morton_sort
g, cam = scenes.benchmark_scene(1_000_000, (2048, 2048), sh_degree=3, seed=0)
cam = cam.to(device='cuda:0'); cfg = RasterConfig()
G = torch.rand(2048, 2048, 3, device='cuda:0')
def run(gg, label):
    gg = gg.to('cuda:0').requires_grad_(True)
    for it in range(25):
        if it == 5:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        for _, t in gg.items(): t.grad = None
        r = gs.render_gaussians(gg, cam, cfg, use_sh=True)
        (r.image * G).sum().backward()
    torch.cuda.synchronize()
    print(label, (time.perf_counter() - t0) / 20 * 1e3, "ms/frame")
run(g, "draw order  ")
order = morton_sort.argsort(g.position.to('cuda:0'), 0.01).long().cpu()
run(g[order], "morton order")
run(g, "draw order  ")
