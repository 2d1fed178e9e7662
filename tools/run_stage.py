"""Dev tool: run the C3 (or other) frame a few times so that rocprofv3 (--kernel-trace / --pmc)
sees steady-state launches of every kernel.   python3 tools/run_stage.py [workload] [frames]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import taichi_gaussian_rasterizer_amd as gs
from bench import WORKLOADS
from taichi_gaussian_rasterizer_amd import RasterConfig, scenes

wl = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
W, H = wl["size"]
g, cam = scenes.benchmark_scene(wl["n"], wl["size"], sh_degree=wl["sh_degree"], seed=0)
g = g.to("cuda:0").requires_grad_(True)
cam = cam.to(device="cuda:0")
G = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(1)).cuda()
for _ in range(frames):
    for _, t in g.items():
        t.grad = None
    r = gs.render_gaussians(g, cam, RasterConfig(), use_sh=True, render_depth=wl["depth"])
    if wl["backward"]:
        r.image.backward(G)
torch.cuda.synchronize()
print("done", int(r.points_in_view.shape[0]))
