import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
import taichi_gaussian_rasterizer_amd as gs
from taichi_gaussian_rasterizer_amd import RasterConfig, scenes
from taichi_gaussian_rasterizer_amd.optim import VisibilityAwareLaProp
n, size = 200_000, (800, 600)
g, cam = scenes.benchmark_scene(n, size, sh_degree=3, seed=0)
cam = cam.to(device='cuda:0')
params = {k: torch.nn.Parameter(v.to('cuda:0')) for k, v in g.items()}
groups = [dict(params=[params[k]], name=k, lr=lr, type=t) for k, lr, t in (("position", 1e-4, "vector"), ("log_scaling", 1e-3, "vector"),
          ("rotation", 1e-3, "vector"), ("alpha_logit", 1e-2, "scalar"), ("feature", 1e-3, "scalar"))]
opt = VisibilityAwareLaProp(groups)
cfg = RasterConfig(compute_visibility=True, compute_point_heuristic=True)
target = torch.rand(size[1], size[0], 3, device='cuda:0')
mem = []
t0 = time.time()
for it in range(1500):
    opt.zero_grad()
    gg = type(g)(**params, batch_size=(n,))
    r = gs.render_gaussians(gg, cam, cfg, use_sh=True, render_depth=(it % 2 == 0))
    loss = torch.nn.functional.l1_loss(r.image, target)
    loss.backward()
    vis = r.point_visibility
    keep = vis > 1e-8
    opt.step(r.points_in_view[keep], vis[keep])
    if it % 250 == 0:
        torch.cuda.synchronize(); mem.append(torch.cuda.memory_allocated() >> 20)
        print(it, float(loss), mem[-1], "MiB", f"{(time.time()-t0):.1f}s", flush=True)
torch.cuda.synchronize()
assert max(mem[3:]) <= max(mem[1:3]) + 32, mem   # no upward trend (the level itself varies with render_depth / K)
assert all(torch.isfinite(p).all() for p in params.values())
print("stress ok", mem)
