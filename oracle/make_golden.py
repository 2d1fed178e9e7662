"""Generate tests/golden/*.npz from the REFERENCE's own pure-torch oracles.

Dev-only: runs in the authoring container (reads /root/reference at run time through
oracle/_ref_loader.py; nothing of the reference is copied).  The committed .npz files hold
numbers only: seeded inputs and the outputs / autograd gradients that
taichi_splatting/torch_lib/projection.py:156-191 (apply), torch_lib/spherical_harmonics.py:32-44
(evaluate_sh_at) and torch_lib/projection.py:120-129 (ndc_depth, inverse_ndc_depth) produce for
them.  Scenes follow the reference tests (tests/test_projection.py:22-33,
tests/test_spherical_harmonics.py:15-31) and the loss is that of tests/util.py:10-33
(sum of .mean() of the float outputs).

    python oracle/make_golden.py            # rewrites tests/golden/
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from _ref_loader import load_reference_torch_lib  # noqa: E402
from taichi_gaussian_rasterizer_amd import scenes  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def projection_case(ref, seed, n, blur_cov, dtype):
    torch.manual_seed(seed)
    camera = scenes.random_camera()
    g = scenes.random_3d_gaussians(n=n, camera_params=camera, margin=0.5, scale_factor=0.1)
    inputs64 = [t.to(torch.float64) for t in (*g.shape_tensors(), camera.T_camera_world, camera.projection)]
    inputs = [t.to(dtype).detach().clone().requires_grad_(True) for t in inputs64]
    points, depth, idx = ref.projection.apply(*inputs, camera.image_size, camera.depth_range, blur_cov=blur_cov,
                                              clamp_margin=0.15, alpha_threshold=1. / 255.)
    loss = points.mean() + depth.mean()
    loss.backward()
    names = ["position", "log_scaling", "rotation", "alpha_logit", "T_camera_world", "projection"]
    out = {f"in_{k}": v.numpy() for k, v in zip(names, inputs64)}
    out.update({f"grad_{k}": (t.grad if t.grad is not None else torch.zeros_like(t)).detach().numpy()
                for k, t in zip(names, inputs)})
    out.update(points=points.detach().numpy(), depth=depth.detach().numpy(), indexes=idx.numpy(),
               image_size=np.array(camera.image_size, np.int64), depth_range=np.array(camera.depth_range, np.float64),
               blur_cov=np.float64(blur_cov))
    return out


def sh_case(ref, seed, degree, C, n, dtype):
    torch.manual_seed(seed)
    params = torch.rand(n, C, (degree + 1) ** 2, dtype=torch.float64)
    points = torch.randn(n, 3, dtype=torch.float64)
    camera_pos = torch.randn(3, dtype=torch.float64)
    indexes = torch.randint(0, n, (max(n // 2, 1),))
    ins = [t.to(dtype).clone().requires_grad_(True) for t in (params, points, camera_pos)]
    out = ref.sh.evaluate_sh_at(ins[0], ins[1], indexes, ins[2])
    out.mean().backward()
    grads = [(t.grad if t.grad is not None else torch.zeros_like(t)).numpy() for t in ins]  # degree 0: no dir grads
    return dict(in_params=params.numpy(), in_points=points.numpy(), in_camera_pos=camera_pos.numpy(),
                indexes=indexes.numpy(), out=out.detach().numpy(), grad_params=grads[0],
                grad_points=grads[1], grad_camera_pos=grads[2])


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = load_reference_torch_lib()

    proj = {}
    # seeds 0..7, n in {1, 17, 1000} (+ the 200-point cases of round 1); the 1000-point scenes with the
    # default blur only, to keep the fixture at a few MB
    for seed in range(8):
        for n in (1, 17, 200, 1000):
            for blur in ((0.3,) if n == 1000 else (0.0, 0.3)):
                for dtype, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
                    case = projection_case(ref, seed, n, blur, dtype)
                    key = f"s{seed}_n{n}_b{int(blur * 10)}_{tag}"
                    for k, v in case.items():
                        if tag == "f32" and k.startswith("in_"):
                            continue  # inputs are stored once (f64); the f32 run used their f32 cast
                        proj[f"{key}/{k}"] = v
    np.savez_compressed(os.path.join(OUT, "projection.npz"), **proj)

    sh = {}
    for seed in range(8):
        degree = seed % 4
        for C in (1, 3):
            n = [1, 2, 7, 33, 64, 101, 50, 90][seed]
            for dtype, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
                case = sh_case(ref, seed, degree, C, n, dtype)
                key = f"s{seed}_d{degree}_c{C}_{tag}"
                for k, v in case.items():
                    if tag == "f32" and k.startswith("in_"):
                        continue
                    sh[f"{key}/{k}"] = v
    np.savez_compressed(os.path.join(OUT, "sh.npz"), **sh)

    torch.manual_seed(0)
    near, far = 0.1, 100.0
    ndc_in = torch.rand(1000, dtype=torch.float32)
    depth = ref.projection.inverse_ndc_depth(ndc_in, near, far)
    ndc = ref.projection.ndc_depth(depth, near, far)
    np.savez_compressed(os.path.join(OUT, "ndc_depth.npz"), ndc_in=ndc_in.numpy(), depth=depth.numpy(),
                        ndc=ndc.numpy(), near=np.float64(near), far=np.float64(far))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
