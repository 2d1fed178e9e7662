"""map_scan_kernel with its scan workgroup only (the composed mapper passes no launch order): under rocprofv3 its time
against the fused frame's two-workgroup launch tells which of the two bounds the kernel"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import taichi_gaussian_rasterizer_amd as gs
from taichi_gaussian_rasterizer_amd import RasterConfig, scenes
n, size = 1_000_000, (2048, 2048)
g, cam = scenes.benchmark_scene(n, size, sh_degree=3, seed=0)
cam = cam.to(device='cuda:0'); cfg = RasterConfig()
r = gs.render_gaussians(g.to('cuda:0'), cam, cfg, use_sh=True)
g2d, depth = r.gaussians2d.detach(), r.point_depth.detach()
for _ in range(30):
    gs.map_to_tiles(g2d, depth, size, cfg)
torch.cuda.synchronize()
