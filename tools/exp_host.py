"""host-side cost of one frame: a scene so small that the GPU time is negligible"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cProfile, pstats
import taichi_gaussian_rasterizer_amd as gs
from taichi_gaussian_rasterizer_amd import RasterConfig, scenes
g, cam = scenes.benchmark_scene(2000, (64, 64), sh_degree=3, seed=0)
cam = cam.to(device='cuda:0'); cfg = RasterConfig()
G = torch.rand(64, 64, 3, device='cuda:0')
gg = g.to('cuda:0').requires_grad_(True)
def step():
    for _, t in gg.items(): t.grad = None
    r = gs.render_gaussians(gg, cam, cfg, use_sh=True)
    r.image.backward(G)
for _ in range(20): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize(); print("host-bound ms/step", (time.perf_counter() - t0) / 200 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
