"""the c3 frame with larger splats (scale_factor 2 = c3; 4, 6: ~4x / ~9x the overlaps, like dense captured scenes)"""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import taichi_gaussian_rasterizer_amd as gs
from taichi_gaussian_rasterizer_amd import RasterConfig, scenes, fused, _native as nv
n, size = 1_000_000, (2048, 2048)
G = torch.rand(size[1], size[0], 3, device='cuda:0')
for sf in ([float(a) for a in sys.argv[1:]] or [2.0, 4.0, 6.0]):
    g, cam = scenes.benchmark_scene(n, size, sh_degree=3, seed=0, scale_factor=sf)
    cam = cam.to(device='cuda:0'); cfg = RasterConfig()
    gg = g.to('cuda:0').requires_grad_(True)
    def step():
        for _, t in gg.items(): t.grad = None
        r = gs.render_gaussians(gg, cam, cfg, use_sh=True)
        r.image.backward(G)
    for _ in range(4): step()
    torch.cuda.synchronize()
    nv.timer.reset(); nv.timer.only = None; nv.timer.enabled = True
    for _ in range(8): step()
    torch.cuda.synchronize(); nv.timer.enabled = False
    st = {k[3:]: round(v[1] / 8, 3) for k, v in nv.timer.summary().items()}
    hint = [v for k, v in fused._K_HINT.items() if k[0] == n][-1]
    print(f"scale {sf}: K={hint[0]} fullest tile={hint[1]} total {sum(st.values()):.3f} ms  {st}", flush=True)
    fused._K_HINT.clear()
