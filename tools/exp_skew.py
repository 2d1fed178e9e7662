"""a frame with a dense cluster (20 % of the Gaussians inside 3 % of the screen): heavy-tile split on / off"""
import os, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import taichi_gaussian_rasterizer_amd as gs
from taichi_gaussian_rasterizer_amd import RasterConfig, scenes, _native as nv
from taichi_gaussian_rasterizer_amd.torch_lib import projection as tp
n, size = 1_000_000, (2048, 2048)
g, cam = scenes.benchmark_scene(n, size, sh_degree=3, seed=0)
gen = torch.Generator().manual_seed(9)
m = int(sys.argv[1]) if len(sys.argv) > 1 else n // 5
uv = torch.tensor([700.0, 900.0]) + torch.randn(m, 2, generator=gen) * torch.tensor([120.0, 90.0])
z = tp.inverse_ndc_depth(torch.rand(m, generator=gen), cam.near_plane, cam.far_plane)
z_old = g.position[:m, 2].clone()
g.position[:m] = tp.unproject_points(uv, z.unsqueeze(1), cam.T_image_world)
if os.environ.get('SKEW_KEEP_SIZE', '1') == '1':  # keep the projected size of the moved Gaussians
    g.log_scaling[:m] += torch.log(z / z_old).unsqueeze(1)
cam = cam.to(device='cuda:0'); cfg = RasterConfig()
G = torch.rand(size[1], size[0], 3, device='cuda:0')
gg = g.to('cuda:0').requires_grad_(True)
for heavy in ("1", "0"):
    nv.TUNING["no_heavy_split"] = 1 if heavy == "0" else 0
    def step():
        for _, t in gg.items(): t.grad = None
        r = gs.render_gaussians(gg, cam, cfg, use_sh=True)
        r.image.backward(G)
        return r
    for _ in range(5): r = step()
    torch.cuda.synchronize()
    nv.timer.reset(); nv.timer.only = None; nv.timer.enabled = True
    for _ in range(10): step()
    torch.cuda.synchronize(); nv.timer.enabled = False
    st = {k[3:]: round(v[1] / v[0], 3) for k, v in nv.timer.summary().items()}
    from taichi_gaussian_rasterizer_amd import fused
    hint = [v for k, v in fused._K_HINT.items() if k[0] == n][-1]
    print(f"split={heavy}  K={hint[0]} fullest tile={hint[1]}  total {sum(st.values()):.3f} ms  {st}", flush=True)
