"""BUILD CONTAINER ONLY (reads /root/reference through oracle/_ref_loader.py; never shipped to or run on the GPU
box): time the genuine reference `torch_lib` projection + SH on this container's CPU cores, beside the CPU oracle
(oracle/gsplat_oracle.cpp, the "port") on the same scene.  SURVEY 8(d): the only timing of real reference code that
is possible here -- the reference's Taichi kernels cannot run (no Taichi, no CUDA).

    python tools/time_reference_torch_lib.py [n]        (default n = 200 000, the per-Gaussian stages of C2)

Reference functions timed: torch_lib/projection.py:156-191 `apply` (project + cull + gather) and
torch_lib/spherical_harmonics.py:32-44 `evaluate_sh_at`, forward only and forward + autograd backward.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import oracle as orc  # noqa: E402
from oracle._ref_loader import load_reference_torch_lib  # noqa: E402
from taichi_gaussian_rasterizer_amd import scenes  # noqa: E402


def best_of(f, reps=3):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    size = (1920, 1080)
    ref = load_reference_torch_lib()
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    orc.set_num_threads(cores)
    g, cam = scenes.benchmark_scene(n, size, sh_degree=3, seed=0)
    args = [*g.shape_tensors(), cam.T_camera_world, cam.projection]
    kw = dict(blur_cov=0.3, clamp_margin=0.15, alpha_threshold=1. / 255.)

    def ref_proj_fwd():
        with torch.no_grad():
            return ref.projection.apply(*args, cam.image_size, cam.depth_range, **kw)

    def ref_proj_fwd_bwd():
        ins = [a.detach().clone().requires_grad_(True) for a in args]
        p, d, _ = ref.projection.apply(*ins, cam.image_size, cam.depth_range, **kw)
        (p.mean() + d.mean()).backward()

    points, depth, idx = ref_proj_fwd()
    cam_pos = cam.camera_position

    def ref_sh_fwd():
        with torch.no_grad():
            return ref.sh.evaluate_sh_at(g.feature, g.position, idx, cam_pos)

    def ref_sh_fwd_bwd():
        f = g.feature.detach().clone().requires_grad_(True)
        ref.sh.evaluate_sh_at(f, g.position, idx, cam_pos).mean().backward()

    npa = [a.numpy() for a in args]

    def orc_proj_fwd():
        return orc.project(*npa, size, cam.depth_range, **kw)

    p_o, d_o, idx_o = orc_proj_fwd()
    gp, gd = np.full_like(p_o, 1.0 / p_o.size), np.full_like(d_o, 1.0 / d_o.size)

    def orc_proj_fwd_bwd():
        orc.project(*npa, size, cam.depth_range, **kw)
        orc.project_backward(*npa, size, idx_o, gp, gd, blur_cov=0.3, clamp_margin=0.15)

    feat, pos, cpos = g.feature.numpy(), g.position.numpy(), cam_pos.numpy()

    def orc_sh_fwd():
        return orc.evaluate_sh_at(feat, pos, idx_o, cpos)

    go = np.full((idx_o.shape[0], 3), 1.0 / (3 * idx_o.shape[0]), np.float32)

    def orc_sh_fwd_bwd():
        orc.evaluate_sh_at(feat, pos, idx_o, cpos)
        orc.evaluate_sh_at_backward(feat, pos, idx_o, cpos, go)

    assert (idx.numpy() == idx_o).all(), "visible sets differ"
    rows = [("projection fwd", ref_proj_fwd, orc_proj_fwd), ("projection fwd+bwd", ref_proj_fwd_bwd, orc_proj_fwd_bwd),
            ("SH deg 3 fwd", ref_sh_fwd, orc_sh_fwd), ("SH deg 3 fwd+bwd", ref_sh_fwd_bwd, orc_sh_fwd_bwd)]
    print(f"n = {n} Gaussians ({idx_o.shape[0]} visible), {cores} CPU threads, f32")
    print(f"{'stage':22s} {'reference torch_lib':>20s} {'oracle (port)':>16s} {'ratio':>8s}")
    for name, fr, fo in rows:
        tr, to = best_of(fr), best_of(fo)
        print(f"{name:22s} {tr * 1e3:17.1f} ms {to * 1e3:13.1f} ms {tr / to:7.1f}x")


if __name__ == "__main__":
    main()
