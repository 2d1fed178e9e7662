"""per-rank stage times of a sharded frame, emulated on one GPU (no process group: the all-reduce is a no-op)"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import taichi_gaussian_rasterizer_amd as gs
from taichi_gaussian_rasterizer_amd import RasterConfig, scenes, parallel, _native as nv
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
interleave = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n, size = (1_000_000, (2048, 2048)) if wl == "c3" else (6_000_000, (4096, 4096))
g, cam = scenes.benchmark_scene(n, size, sh_degree=3, seed=0)
cam = cam.to(device='cuda:0'); cfg = RasterConfig()
G = torch.rand(size[1], size[0], 3, device='cuda:0')
gg = g.to('cuda:0').requires_grad_(True)
rows_of = {}
base = None
for world in (1, 2, 4, 8):
    rank = world // 2
    def step():
        for _, t in gg.items(): t.grad = None
        r = parallel.render_gaussians_sharded(gg, cam, cfg, use_sh=True, rank=rank, world_size=world,
                                              interleave=interleave)
        if world not in rows_of:
            rows_of[world] = G[parallel.owned_pixel_rows(r.bands).cuda()].contiguous()
        r.image.backward(rows_of[world])
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
    nv.timer.reset(); nv.timer.only = None; nv.timer.enabled = True
    for _ in range(5): step()
    torch.cuda.synchronize(); nv.timer.enabled = False
    st = {k: round(v[1] / 5, 3) for k, v in nv.timer.summary().items()}
    base = base or dt
    print(f"world {world} rank {rank}: {dt:.3f} ms/frame (no collective), {base / dt:.2f}x of one GPU ", st)
