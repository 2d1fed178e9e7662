import sys, os
sys.path.insert(0, os.getcwd())
import torch
import taichi_gaussian_rasterizer_amd as gs
from taichi_gaussian_rasterizer_amd import RasterConfig, scenes, _native as nv
n, size = 1_000_000, (2048, 2048)
g, cam = scenes.benchmark_scene(n, size, sh_degree=3, seed=0)
cam = cam.to(device='cuda:0')
G = torch.rand(size[1], size[0], 3, device='cuda:0')
gg = g.to('cuda:0').requires_grad_(True)
cfg = RasterConfig(antialias=True, blur_cov=0.0)
def step():
    for _, t in gg.items(): t.grad = None
    r = gs.render_gaussians(gg, cam, cfg, use_sh=True)
    r.image.backward(G)
for _ in range(3): step()
torch.cuda.synchronize()
nv.timer.reset(); nv.timer.only = None; nv.timer.enabled = True
for _ in range(6): step()
torch.cuda.synchronize(); nv.timer.enabled = False
st = {k[3:]: round(v[1] / 6, 3) for k, v in nv.timer.summary().items()}
print(os.environ.get("GS_LIB_PATH","default")[-12:], st["raster_fwd"], st["raster_bwd"])
