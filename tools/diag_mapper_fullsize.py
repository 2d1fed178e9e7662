"""Which (gaussian, tile) pairs does the HIP mapper list differently from the CPU oracle on a full-size frame?
usage (GPU box): python tools/diag_mapper_fullsize.py [c5]   -- prints the differing tiles, the splats, and which of the
two HIP paths (fused / reference-staged) agrees with the oracle."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as pu  # noqa: E402
from oracle import oracle as orc  # noqa: E402

import taichi_gaussian_rasterizer_amd as gs  # noqa: E402
from taichi_gaussian_rasterizer_amd import RasterConfig, scenes  # noqa: E402
from taichi_gaussian_rasterizer_amd.mapper.tile_mapper import map_to_tiles_reference_stages  # noqa: E402
from taichi_gaussian_rasterizer_amd.perspective import projection as hip_proj  # noqa: E402

WL = {"c3": (1_000_000, (2048, 2048)), "c5": (6_000_000, (4096, 4096))}
n, size = WL[sys.argv[1] if len(sys.argv) > 1 else "c5"]
orc.set_num_threads(os.cpu_count() or 1)
g, camera = scenes.benchmark_scene(n, size, sh_degree=0, seed=0)
cfg = RasterConfig()
ocfg = orc.OracleConfig.of(cfg)
cam = camera.to(device="cuda:0")
gd = g.to("cuda:0")
g2d, depths, idx, ndc = hip_proj.project_with_ndc(*gd.shape_tensors(), cam.T_camera_world, cam.projection,
                                                  cam.image_size, cam.depth_range, cfg)
p_np, ndc_np = pu.to_np(g2d), pu.to_np(ndc)
o2p_ref, ranges_ref = orc.map_to_tiles(p_np, ndc_np, size, ocfg)
cnt_ref = (ranges_ref[..., 1] - ranges_ref[..., 0]).reshape(-1)
for name, fn in (("fused", gs.map_to_tiles), ("reference-staged", map_to_tiles_reference_stages)):
    o2p, ranges = fn(g2d, ndc, size, cfg)
    o2p, ranges = pu.to_np(o2p), pu.to_np(ranges)
    cnt = (ranges[..., 1] - ranges[..., 0]).reshape(-1)
    bad = np.nonzero(cnt != cnt_ref)[0]
    print(f"{name}: K = {o2p.shape[0]} (oracle {o2p_ref.shape[0]}); tiles with a different count: {bad.tolist()[:20]}")
    tw = ranges.shape[1]
    for t in bad[:8]:
        a = set(o2p[ranges.reshape(-1, 2)[t, 0]:ranges.reshape(-1, 2)[t, 1]].tolist())
        b = set(o2p_ref[ranges_ref.reshape(-1, 2)[t, 0]:ranges_ref.reshape(-1, 2)[t, 1]].tolist())
        for s in sorted(a ^ b):
            row = p_np[s]
            print(f"  tile {t} = ({t % tw}, {t // tw}) splat {s} in {'hip' if s in a else 'oracle'} only: "
                  f"row {[float(x).hex() for x in row]} = {row.tolist()}")
    if bad.size == 0:
        same = (o2p == o2p_ref).all() and (ranges == ranges_ref).all()
        print(f"  counts equal; o2p / ranges identical: {bool(same)}")
