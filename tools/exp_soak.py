"""Determinism soak: the same frame rendered many times -- images must be bit-identical (fixed per-pixel blend order),
tile lists identical, gradients equal up to the order of the float atomics.  Catches rare races in the sorts / splits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import taichi_gaussian_rasterizer_amd as gs
from taichi_gaussian_rasterizer_amd import RasterConfig, scenes

def soak(n, size, frames, scale_factor, cfg, **kw):
    g, cam = scenes.benchmark_scene(n, size, sh_degree=3, seed=0, scale_factor=scale_factor)
    cam = cam.to(device='cuda:0')
    gg = g.to('cuda:0').requires_grad_(True)
    G = torch.rand(size[1], size[0], 3, device='cuda:0')
    ref_img = ref_grad = None
    worst = 0.0
    for it in range(frames):
        for _, t in gg.items(): t.grad = None
        r = gs.render_gaussians(gg, cam, cfg, use_sh=True, **kw)
        r.image.backward(G)
        if ref_img is None:
            ref_img, ref_grad = r.image.detach().clone(), {k: t.grad.clone() for k, t in gg.items()}
            continue
        assert torch.equal(r.image, ref_img), f"frame {it}: image differs"
        for k, t in gg.items():
            err = float((t.grad - ref_grad[k]).norm() / ref_grad[k].norm().clamp_min(1e-30))
            worst = max(worst, err)
            assert err < 1e-3, (it, k, err)
    torch.cuda.synchronize()
    print(f"n={n} {size} scale={scale_factor} {kw}: {frames} identical frames, worst gradient deviation {worst:.2e}", flush=True)

soak(1_000_000, (2048, 2048), 300, 2.0, RasterConfig())
soak(1_000_000, (2048, 2048), 150, 6.0, RasterConfig())                      # crowded tiles: merge-sort classes
soak(300_000, (1024, 768), 300, 3.0, RasterConfig(compute_visibility=True, compute_point_heuristic=True), render_depth=True)
soak(100_000, (640, 480), 300, 4.0, RasterConfig(tile_size=32))
soak(20_000, (256, 192), 300, 6.0, RasterConfig())                            # small grid: 8x8 wave regions
