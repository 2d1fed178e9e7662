"""rasterizer stage times with the mapper's heaviest-first launch order (fused frame) against XCD-contiguous spatial bands
(the composed operators pass no order)"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import taichi_gaussian_rasterizer_amd as gs
from taichi_gaussian_rasterizer_amd import RasterConfig, scenes, fused, _native as nv
fused.FRAME_CALLS = False  # stage by stage: the per-entry-point timer sees the rasterizer calls
from taichi_gaussian_rasterizer_amd.mapper.tile_mapper import map_to_tiles
from taichi_gaussian_rasterizer_amd.rasterizer.function import rasterize_with_tiles
n, size = 1_000_000, (2048, 2048)
g, cam = scenes.benchmark_scene(n, size, sh_degree=3, seed=0)
cam = cam.to(device='cuda:0'); cfg = RasterConfig()
G = torch.rand(size[1], size[0], 3, device='cuda:0')
gg = g.to('cuda:0').requires_grad_(True)
r = gs.render_gaussians(gg, cam, cfg, use_sh=True)
g2d = r.gaussians2d.detach().clone().requires_grad_(True)
feats = torch.rand(g2d.shape[0], 3, device='cuda:0', requires_grad=True)
depth = r.point_depth.detach()
o2p, ranges = map_to_tiles(g2d.detach(), depth, size, cfg)
def composed():
    g2d.grad = None; feats.grad = None
    out = rasterize_with_tiles(g2d, feats, o2p, ranges.reshape(-1, 2), size, cfg)
    out.image.backward(G)
def fused():
    for _, t in gg.items(): t.grad = None
    gs.render_gaussians(gg, cam, cfg, use_sh=True).image.backward(G)
for name, step in (("spatial bands (no order)", composed), ("heaviest first (fused)", fused), ("spatial bands (no order)", composed)):
    for _ in range(5): step()
    torch.cuda.synchronize()
    nv.timer.reset(); nv.timer.only = None; nv.timer.enabled = True
    for _ in range(10): step()
    torch.cuda.synchronize(); nv.timer.enabled = False
    st = {k: round(v[1] / v[0], 4) for k, v in nv.timer.summary().items() if "raster" in k}
    print(name, st, flush=True)
