"""raster stage times of one rank's strip of the C3 frame (emulated, no collective) for each wave-region variant"""
import os, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from taichi_gaussian_rasterizer_amd import RasterConfig, scenes, parallel, _native as nv
n, size = 1_000_000, (2048, 2048)
g, cam = scenes.benchmark_scene(n, size, sh_degree=3, seed=0)
cam = cam.to(device='cuda:0'); cfg = RasterConfig()
G = torch.rand(size[1], size[0], 3, device='cuda:0')
gg = g.to('cuda:0').requires_grad_(True)
for world in (2, 4, 8, 16):
    rank = world // 2
    line = f"world {world:2d} (tiles {128 * 128 // world:5d}):"
    for nb in ("1", "2", "4"):
        nv.TUNING["wave_sub_blocks"] = int(nb)
        def step():
            for _, t in gg.items(): t.grad = None
            r = parallel.render_gaussians_sharded(gg, cam, cfg, use_sh=True, rank=rank, world_size=world)
            y0, y1 = r.strip
            r.image.backward(G[y0:y1])
        for _ in range(4): step()
        torch.cuda.synchronize()
        nv.timer.reset(); nv.timer.only = {"gs_raster_fwd", "gs_raster_bwd"}; nv.timer.enabled = True
        for _ in range(10): step()
        torch.cuda.synchronize(); nv.timer.enabled = False
        st = {k: v[1] / v[0] for k, v in nv.timer.summary().items()}
        line += f"  NB{nb}: fwd {st['gs_raster_fwd']*1e3:6.1f} bwd {st['gs_raster_bwd']*1e3:6.1f}"
    print(line, flush=True)
