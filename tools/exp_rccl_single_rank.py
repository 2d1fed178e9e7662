import os, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
import taichi_gaussian_rasterizer_amd as gs
from taichi_gaussian_rasterizer_amd import RasterConfig, scenes, parallel
real = dist.get_world_size
dist.get_world_size = lambda group=None: 2      # force the collective code path with a 1-rank RCCL communicator
g, cam = scenes.benchmark_scene(50000, (512, 512), sh_degree=3, seed=0)
g = g.to("cuda:0").requires_grad_(True); cam = cam.to(device="cuda:0")
G = torch.rand(512, 512, 3, device="cuda:0")
for it in range(3):
    for _, t in g.items(): t.grad = None
    r = parallel.render_gaussians_sharded(g, cam, RasterConfig(), use_sh=True, rank=0, world_size=1)
    r.image.backward(G)
torch.cuda.synchronize()
a = {k: v.grad.clone() for k, v in g.items()}
for _, t in g.items(): t.grad = None
r2 = gs.render_gaussians(g, cam, RasterConfig(), use_sh=True)
r2.image.backward(G)
torch.cuda.synchronize()
for k, v in g.items():
    err = float((a[k] - v.grad).norm() / v.grad.norm().clamp_min(1e-30))
    print(k, err)
    assert err < 1e-3
dist.get_world_size = real
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl single-rank collective path ok")
