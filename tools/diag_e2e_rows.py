"""diagnostic for the end-to-end gradient bar: per-row error of the GPU pipeline vs the CPU oracle pipeline, binned by the
conditioning of each Gaussian's 2D eigen-decomposition (relative eigenvalue gap of the projected covariance)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import parity_util as pu
import taichi_gaussian_rasterizer_amd as gs
from taichi_gaussian_rasterizer_amd import RasterConfig, scenes

CASES = [(0, 2000, (160, 120), 3, False), (1, 20000, (320, 240), 3, True), (2, 5000, (200, 200), 0, False),
         (3, 3000, (129, 65), 2, True)]
for seed, n, size, deg, depth_mode in CASES:
    g, camera = scenes.benchmark_scene(n, size, sh_degree=deg, seed=seed)
    cfg = RasterConfig()
    gi = torch.rand(size[1], size[0], 3, generator=torch.Generator().manual_seed(seed + 7))
    ref = pu.oracle_render(g, camera, cfg, use_sh=True, render_depth=depth_mode, grads=dict(image=gi.numpy()))
    gd = g.to("cuda:0").requires_grad_(True)
    r = gs.render_gaussians(gd, camera.to(device="cuda:0"), cfg, use_sh=True, render_depth=depth_mode)
    (r.image * gi.to("cuda:0")).sum().backward()
    cov = pu.cov_form(ref["points"])
    c00, c01, c11 = cov[:, 2], cov[:, 3], cov[:, 4]
    relgap_v = np.sqrt((c00 - c11) ** 2 + 4 * c01 ** 2) / (c00 + c11)
    relgap = np.full(n, np.inf)
    relgap[ref["indexes"]] = relgap_v
    print(f"case seed={seed} n={n} V={len(relgap_v)} depth={depth_mode}")
    for name, key in (("position", "d_position"), ("log_scaling", "d_log_scaling"), ("rotation", "d_rotation"),
                      ("alpha_logit", "d_alpha_logit"), ("feature", "d_feature")):
        hip = pu.to_np(getattr(gd, name).grad).astype(np.float64).reshape(n, -1)
        rf = ref[key].astype(np.float64).reshape(n, -1)
        err = np.linalg.norm(hip - rf, axis=1); mag = np.linalg.norm(rf, axis=1)
        floor = 1e-3 * float(np.median(mag[mag > 0]))
        rel = err / (mag + floor)
        vis = np.isfinite(relgap)
        line = f"  {name:12s}"
        for gap_t in (0.0, 1e-3, 1e-2, 3e-2, 1e-1):
            keep = vis & (relgap >= gap_t)
            bad = rel[keep] > 1e-2
            line += f" | gap>={gap_t:g}: excl {1 - keep.sum() / vis.sum():.4f} fail@1e-2 {bad.mean():.5f} max {rel[keep].max():.2e}"
        print(line)
        worst = np.argsort(-rel)[:5]
        print("     worst rows: " + ", ".join(f"rel {rel[i]:.2e} gap {relgap[i]:.2e} mag {mag[i]:.2e}" for i in worst))
