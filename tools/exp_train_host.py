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
T = dict(fwd=0.0, loss_bwd=0.0, select=0.0, step=0.0)
def it():
    t0 = time.perf_counter()
    opt.zero_grad()
    gg = type(g)(**params, batch_size=(n,))
    r = gs.render_gaussians(gg, cam, cfg, use_sh=True)
    t1 = time.perf_counter()
    torch.nn.functional.l1_loss(r.image, target).backward()
    t2 = time.perf_counter()
    vis = r.point_visibility
    keep = vis > 1e-8
    idx, w = r.points_in_view[keep], vis[keep]
    t3 = time.perf_counter()
    opt.step(idx, w)
    t4 = time.perf_counter()
    T['fwd'] += t1 - t0; T['loss_bwd'] += t2 - t1; T['select'] += t3 - t2; T['step'] += t4 - t3
for _ in range(50): it()
torch.cuda.synchronize()
for k in T: T[k] = 0.0
t0 = time.perf_counter()
for _ in range(300): it()
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / 300 * 1e3
print(f"iteration {tot:.3f} ms; host time per section (ms):", {k: round(v / 300 * 1e3, 3) for k, v in T.items()})
