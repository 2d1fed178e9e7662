import os, sys
sys.path.insert(0, os.getcwd())
import torch
import taichi_gaussian_rasterizer_amd as gs
from taichi_gaussian_rasterizer_amd import RasterConfig, scenes, parallel, _native as nv
from taichi_gaussian_rasterizer_amd.torch_lib import projection as tp
n, size = 1_000_000, (2048, 2048)
tag = os.environ.get("GS_LIB_PATH", "default")[-12:]
def scene(m, keep):
    g, cam = scenes.benchmark_scene(n, size, sh_degree=3, seed=0)
    gen = torch.Generator().manual_seed(9)
    if m:
        uv = torch.tensor([700.0, 900.0]) + torch.randn(m, 2, generator=gen) * torch.tensor([120.0, 90.0])
        z = tp.inverse_ndc_depth(torch.rand(m, generator=gen), cam.near_plane, cam.far_plane)
        z_old = g.position[:m, 2].clone()
        g.position[:m] = tp.unproject_points(uv, z.unsqueeze(1), cam.T_image_world)
        if keep:
            g.log_scaling[:m] += torch.log(z / z_old).unsqueeze(1)
    return g, cam.to(device='cuda:0')
G = torch.rand(size[1], size[0], 3, device='cuda:0')
cfg = RasterConfig()
def timeit(step):
    for _ in range(4): step()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(12): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 12 * 1e3
out = []
for name, m, keep in (("uniform", 0, True), ("cluster50k", 50000, True), ("cluster200k", 200000, True), ("floaters50k", 50000, False)):
    g, cam = scene(m, keep)
    gg = g.to('cuda:0').requires_grad_(True)
    def step():
        for _, t in gg.items(): t.grad = None
        r = gs.render_gaussians(gg, cam, cfg, use_sh=True)
        r.image.backward(G)
    out.append(f"{name} {timeit(step):.3f}")
    if name in ("uniform", "cluster200k"):
        rows = {}
        def step8():
            for _, t in gg.items(): t.grad = None
            r = parallel.render_gaussians_sharded(gg, cam, cfg, use_sh=True, rank=4, world_size=8)
            if 8 not in rows: rows[8] = G[parallel.owned_pixel_rows(r.bands).cuda()].contiguous()
            r.image.backward(rows[8])
        out.append(f"{name}/8ranks {timeit(step8):.3f}")
print(tag, " | ".join(out))
