"""host-side cost of one frame: a scene so small that the GPU time is negligible"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cProfile, pstats
import taichi_gaussian_rasterizer_amd as gs
from taichi_gaussian_rasterizer_amd import RasterConfig, scenes, fused
g, cam = scenes.benchmark_scene(2000, (64, 64), sh_degree=3, seed=0)
cam = cam.to(device='cuda:0'); cfg = RasterConfig()
G = torch.rand(64, 64, 3, device='cuda:0')
gg = g.to('cuda:0').requires_grad_(True)
def step():
    for _, t in gg.items(): t.grad = None
    r = gs.render_gaussians(gg, cam, cfg, use_sh=True)
    r.image.backward(G)
for round_ in range(3):  # alternating, so that clock / box effects hit both alike
    for calls in (True, False):
        fused.FRAME_CALLS = calls
        for _ in range(20): step()
        torch.cuda.synchronize(); t0 = time.perf_counter(); c0 = time.process_time()
        for _ in range(300): step()
        c1 = time.process_time(); torch.cuda.synchronize()
        print(f"round {round_}: ms/step ({'gs_frame_fwd / gs_frame_bwd' if calls else 'one entry point per stage'}):",
              round((time.perf_counter() - t0) / 300 * 1e3, 4), " process CPU ms/step:", round((c1 - c0) / 300 * 1e3, 4))
fused.FRAME_CALLS = True
# forward and backward apart (each followed by a device synchronisation)
tf = tb = 0.0
for _ in range(200):
    for _, t in gg.items(): t.grad = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = gs.render_gaussians(gg, cam, cfg, use_sh=True)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    r.image.backward(G)
    t3 = time.perf_counter()
    tf += t1 - t0; tb += t3 - t2
print(f"forward call returns after {tf / 200 * 1e3:.4f} ms (includes the wait for the mapper's counts), backward call after {tb / 200 * 1e3:.4f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
