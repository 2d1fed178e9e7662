"""rasterizer wave-region variants (GS_RASTER_NB = 1 | 2 | 4 sub-blocks per wave) across image sizes"""
import os, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import taichi_gaussian_rasterizer_amd as gs
from taichi_gaussian_rasterizer_amd import RasterConfig, scenes, _native as nv

cfg = RasterConfig()
cases = [(2000, (256, 256)), (50_000, (512, 512)), (200_000, (1024, 768)), (300_000, (1280, 960)), (400_000, (1600, 1200)),
         (1_000_000, (2048, 2048))]
for n, size in cases:
    g, cam = scenes.benchmark_scene(n, size, sh_degree=3, seed=0)
    cam = cam.to(device='cuda:0')
    gg = g.to('cuda:0').requires_grad_(True)
    G = torch.rand(size[1], size[0], 3, device='cuda:0')
    tiles = -(-size[0] // 16) * -(-size[1] // 16)
    line = f"n={n:8d} {size[0]}x{size[1]} tiles={tiles:6d}:"
    for nb in (1, 2, 4):
        nv.TUNING["wave_sub_blocks"] = int(nb)
        def step():
            for _, t in gg.items(): t.grad = None
            r = gs.render_gaussians(gg, cam, cfg, use_sh=True)
            r.image.backward(G)
        for _ in range(5): step()
        torch.cuda.synchronize()
        nv.timer.reset(); nv.timer.only = {"gs_raster_fwd", "gs_raster_bwd"}; nv.timer.enabled = True
        for _ in range(20): step()
        torch.cuda.synchronize(); nv.timer.enabled = False
        st = {k: v[1] / v[0] for k, v in nv.timer.summary().items()}
        line += f"  NB{nb}: fwd {st['gs_raster_fwd']*1e3:6.1f} bwd {st['gs_raster_bwd']*1e3:6.1f} us"
    print(line, flush=True)
