"""Host-side mirror of the reference operator API: types, validation, error behaviour.
No GPU: the operators must refuse CPU tensors loudly (there is no CPU path in the product)."""
import dataclasses

import pytest
import torch

import taichi_gaussian_rasterizer_amd as gs
from taichi_gaussian_rasterizer_amd import (CameraParams, Gaussians2D, Gaussians3D, RasterConfig, Rendering,
                                            TaichiQueue, pad_to_tile, scenes, taichi_queue)
from taichi_gaussian_rasterizer_amd.taichi_queue import queued


def test_public_names_match_reference_init():
    # reference taichi_splatting/__init__.py:17-33
    for name in ("render_gaussians", "Rendering", "map_to_tiles", "pad_to_tile", "Gaussians2D", "Gaussians3D",
                 "RasterConfig", "evaluate_sh_at", "rasterize", "rasterize_with_tiles", "perspective", "TaichiQueue"):
        assert hasattr(gs, name), name
    assert hasattr(gs.perspective, "project_to_image") and hasattr(gs.perspective, "CameraParams")
    assert hasattr(gs.cuda_lib, "full_cumsum") and hasattr(gs.cuda_lib, "radix_sort_pairs")


def test_raster_config_defaults_and_hashing():
    c = RasterConfig()
    assert (c.tile_size, c.pixel_stride, c.clamp_margin, c.antialias, c.blur_cov) == (16, (2, 2), 0.15, False, 0.3)
    assert (c.clamp_max_alpha, c.saturate_threshold, c.use_alpha_blending) == (0.99, 0.9999, True)
    assert abs(c.alpha_threshold - 1 / 255) < 1e-12 and not c.compute_visibility and not c.compute_point_heuristic
    assert hash(c) == hash(RasterConfig()) and c == RasterConfig()
    with pytest.raises(dataclasses.FrozenInstanceError):
        c.tile_size = 8
    with pytest.raises(TypeError):
        RasterConfig(tile_size=16.0)
    with pytest.raises(TypeError):
        RasterConfig(16)  # kw_only, as the reference
    assert dataclasses.replace(c, use_alpha_blending=False, saturate_threshold=0.5).saturate_threshold == 0.5


def test_gaussians3d_record():
    torch.manual_seed(0)
    cam = scenes.random_camera()
    g = scenes.random_3d_gaussians(10, cam)
    assert tuple(g.batch_size) == (10,) and g.packed().shape == (10, 11) and len(g.shape_tensors()) == 4
    assert torch.allclose(g.scale, g.log_scaling.exp()) and torch.allclose(g.alpha, g.alpha_logit.sigmoid())
    assert g[2:5].position.shape == (3, 3) and tuple(g[2:5].batch_size) == (3,)
    assert g.concat(g).feature.shape[0] == 20
    assert g.replace(feature=torch.zeros(10, 3, 16)).feature.shape == (10, 3, 16)
    assert g.to(dtype=torch.float64).position.dtype == torch.float64
    g.requires_grad_(True)
    assert all(t.requires_grad for _, t in g.items())
    with pytest.raises(AssertionError):
        Gaussians3D(position=torch.zeros(4, 2), log_scaling=torch.zeros(4, 3), rotation=torch.zeros(4, 4),
                    alpha_logit=torch.zeros(4, 1), feature=torch.zeros(4, 3), batch_size=(4,))
    with pytest.raises(RuntimeError):
        Gaussians3D(position=torch.zeros(4, 3), log_scaling=torch.zeros(5, 3), rotation=torch.zeros(4, 4),
                    alpha_logit=torch.zeros(4, 1), feature=torch.zeros(4, 3), batch_size=(4,))
    g2 = scenes.random_2d_gaussians(7, (32, 32))
    assert isinstance(g2, Gaussians2D) and g2.opacity.shape == (7,) and g2.scaling.shape == (7, 2)
    assert Gaussians2D.from_tensordict(g2.to_tensordict()).position.shape == (7, 2)


def test_camera_params():
    cam = scenes.benchmark_camera((640, 480))
    assert cam.depth_range == (0.1, 100.0) and cam.image_size == (640, 480)
    assert torch.allclose(cam.camera_position, torch.zeros(3))
    assert cam.T_image_world.shape == (4, 4) and cam.scale_image(0.5).image_size == (320, 240)
    with pytest.raises(AssertionError):
        CameraParams(projection=torch.zeros(4), T_camera_world=torch.eye(4), near_plane=0.0, far_plane=1.0,
                     image_size=(4, 4))
    with pytest.raises(AssertionError):
        CameraParams(projection=torch.zeros(3), T_camera_world=torch.eye(4), near_plane=0.1, far_plane=1.0,
                     image_size=(4, 4))


def test_pad_to_tile_and_queue_shim():
    assert pad_to_tile((257, 131), 16) == (272, 144) and pad_to_tile((32, 32), 16) == (32, 32)
    with taichi_queue(arch="anything", log_level=0):
        assert TaichiQueue.run_sync(lambda a, b: a + b, 1, 2) == 3
        assert TaichiQueue.run_async(lambda: 5).result() == 5
    TaichiQueue.init(threaded=True)
    assert queued(lambda x: x * 2)(4) == 8
    TaichiQueue.stop()


def test_operators_refuse_cpu_tensors_loudly():
    g, cam = scenes.benchmark_scene(50, (64, 64), sh_degree=1)
    cfg = RasterConfig()
    with pytest.raises(RuntimeError, match="HIP device"):
        gs.render_gaussians(g, cam, cfg, use_sh=True)
    with pytest.raises(RuntimeError, match="HIP device"):
        gs.perspective.project_to_image(g, cam, cfg)
    with pytest.raises(RuntimeError, match="HIP device"):
        gs.evaluate_sh_at(g.feature, g.position, torch.arange(5), torch.zeros(3))
    g2d = torch.rand(5, 7)
    with pytest.raises(RuntimeError, match="HIP device"):
        gs.map_to_tiles(g2d, torch.rand(5, 1), (64, 64), cfg)
    with pytest.raises(RuntimeError, match="HIP device"):
        gs.rasterize_with_tiles(g2d, torch.rand(5, 3), torch.zeros(0, dtype=torch.int32),
                                torch.zeros((16, 2), dtype=torch.int32), (64, 64), cfg)


def test_argument_validation_mirrors_reference():
    cfg = RasterConfig()
    with pytest.raises(AssertionError, match="Nx7"):
        gs.map_to_tiles(torch.rand(5, 6), torch.rand(5, 1), (64, 64), cfg)  # tile_mapper.py:219
    with pytest.raises(AssertionError, match="Nx1"):
        gs.map_to_tiles(torch.rand(5, 7), torch.rand(5), (64, 64), cfg)     # tile_mapper.py:220
    with pytest.raises(TypeError):
        gs.map_to_tiles(torch.rand(5, 7), torch.rand(5, 1), (64.0, 64), cfg)
    with pytest.raises(AssertionError, match="Size mismatch"):
        gs.rasterize(torch.rand(5, 7), torch.rand(4, 1), torch.rand(5, 3), (64, 64), cfg)  # function.py:149
    with pytest.raises(TypeError):
        gs.render_gaussians("nope", scenes.benchmark_camera((8, 8)))
    with pytest.raises(AssertionError, match="square"):
        gs.evaluate_sh_at(torch.rand(4, 3, 5), torch.rand(4, 3), torch.arange(4), torch.zeros(3))


def test_rendering_properties():
    cam = scenes.benchmark_camera((8, 8))
    cfg = RasterConfig(compute_visibility=True)
    r = Rendering(image=torch.zeros(8, 8, 3), image_weight=torch.zeros(8, 8), points_in_view=torch.arange(4),
                  point_depth=torch.ones(4, 1), point_visibility=torch.tensor([0., 1., 0., 2.]), camera=cam,
                  config=cfg, gaussians2d=torch.rand(4, 7) + 0.1, depth=torch.ones(8, 8))
    assert r.num_points == 4 and r.image_size == (8, 8) and r.point_scale.shape == (4, 2)
    assert r.visible_indices.tolist() == [1, 3] and r.visible[1].tolist() == [1.0, 2.0]
    assert r.point_radii.shape == (4,) and r.gaussian_scale.shape == (4,) and r.ndc_depth.shape == (8, 8)
    with pytest.raises(AssertionError):
        _ = r.prune_cost
    assert r.detach().image.shape == (8, 8, 3)


def test_install_as_taichi_splatting():
    import sys
    gs.install_as_taichi_splatting()
    import taichi_splatting
    from taichi_splatting.rasterizer import rasterize  # noqa: F401
    from taichi_splatting.perspective import CameraParams as C2
    assert taichi_splatting is gs and C2 is CameraParams
    for k in [k for k in sys.modules if k == "taichi_splatting" or k.startswith("taichi_splatting.")]:
        del sys.modules[k]


def test_renderer2d_split_helpers_cpu():
    """the 2D example's densification helpers (reference misc/renderer2d.py:60-132): shapes, shrink factors, offsets
    inside the parent's frame"""
    import math
    from taichi_gaussian_rasterizer_amd import scenes
    from taichi_gaussian_rasterizer_amd.misc import renderer2d as r2d
    torch.manual_seed(0)
    g = scenes.random_2d_gaussians(50, (64, 48))
    s = r2d.split_gaussians2d(g, n=3)
    assert s.batch_size[0] == 150 and s.position.shape == (150, 2)
    assert torch.allclose(s.log_scaling, g.log_scaling.repeat_interleave(3, 0) + math.log(1 / math.sqrt(3)))
    assert (s.z_depth > 0).all() and torch.equal(s.feature, g.feature.repeat_interleave(3, 0))
    u = r2d.uniform_split_gaussians2d(g, n=2, sep=0.7)
    assert u.batch_size[0] == 100
    long_axis = torch.argmax(g.log_scaling, dim=1)
    ratio = (u.scaling / g.scaling.repeat_interleave(2, 0))
    picked = ratio.gather(1, long_axis.repeat_interleave(2).unsqueeze(1)).squeeze(1)
    assert torch.allclose(picked, torch.full_like(picked, math.sqrt(2) / 2), atol=1e-5)
    # the two copies sit at -sep and +sep sigma along the parent's long axis: their midpoint is the parent
    mid = u.position.view(50, 2, 2).mean(1)
    assert torch.allclose(mid, g.position, atol=1e-4)
    off = r2d.sample_gaussians(g)
    assert off.shape == (50, 2) and torch.isfinite(off).all()
    cov = r2d.point_covariance(g)
    assert torch.allclose(cov, cov.transpose(1, 2), atol=1e-5) and (torch.linalg.eigvalsh(cov) > 0).all()


def test_helper_modules_cpu():
    """torch_lib.transforms / torch_lib.util / misc.indexing / optim.autograd (reference modules of the same names)"""
    from taichi_gaussian_rasterizer_amd import scenes
    from taichi_gaussian_rasterizer_amd.misc.indexing import index_features
    from taichi_gaussian_rasterizer_amd.optim.autograd import restore_grad
    from taichi_gaussian_rasterizer_amd.torch_lib import transforms, util
    torch.manual_seed(0)
    q = torch.nn.functional.normalize(torch.randn(5, 4), dim=1)
    R = transforms.quat_to_mat(q)
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3).expand(5, 3, 3), atol=1e-5)
    T = transforms.join_rt(R[0], torch.tensor([1.0, 2.0, 3.0]))
    r, t = transforms.split_rt(T)
    assert torch.equal(r, R[0]) and torch.equal(t, torch.tensor([1.0, 2.0, 3.0]))
    p = torch.randn(7, 3)
    assert torch.allclose(transforms.transform33(R[0], p), p @ R[0].T, atol=1e-6)
    assert torch.allclose(transforms.transform44(T, transforms.make_homog(p))[:, :3], p @ R[0].T + t, atol=1e-5)
    f = torch.randn(6, 2, 3, requires_grad=True)
    idx = torch.tensor([5, 0, 0, 3])
    out = index_features(f, idx)
    assert out.shape == (4, 2, 3)
    out.sum().backward()
    assert torch.equal(f.grad[0], torch.full((2, 3), 2.0)) and torch.equal(f.grad[1], torch.zeros(2, 3))
    g = scenes.random_2d_gaussians(4, (8, 8))
    assert util.count_nonfinite(g, "g") == {}
    g.position[1, 0] = float("nan")
    assert util.count_nonfinite([g, {"x": torch.ones(2)}], "args") == {"args[0].position": 1}
    try:
        util.check_finite(g, "g")
        raise AssertionError("expected ValueError")
    except ValueError:
        pass
    w = torch.ones(3, requires_grad=True)
    w.grad = torch.full((3,), 7.0)
    with restore_grad(w):
        assert torch.equal(w.grad, torch.zeros(3))
        w.grad += 1
    assert torch.equal(w.grad, torch.full((3,), 7.0))


def test_bench_launches_n_ranks_as_a_child_process():
    """bench.py --gpus N without a launcher around it starts torch.distributed.run itself (child process, never exec)
    and a rank whose WORLD_SIZE disagrees with --gpus refuses to print a line"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "7", "--warmup",
                          "2", "--dry-run-launch"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr
    cmd = out.stdout.strip().split()
    assert "torch.distributed.run" in cmd and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(os.path.join(root, "bench.py")) + 1:]
    assert tail[:6] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    # a mislabelled world is an error, not a 1-GPU number printed as n_gpus: 1
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         env=env2, timeout=300)
    assert out.returncode != 0 and "refusing" in out.stderr and out.stdout.strip() == ""


def test_device_guard_logic():
    """every operator runs on the device of its tensors (nv.on_tensor_device) and refuses tensors of one call that
    live on different devices (nv.check_same_device) -- pure host logic, checked with stand-in tensors"""
    import pytest
    import torch
    from taichi_gaussian_rasterizer_amd import _native as nv

    class Fake:
        is_cuda = True

        def __init__(self, index):
            self.device = torch.device("cuda", index)

    nv.check_same_device([Fake(1), None, Fake(1), torch.zeros(2)])  # CPU tensors are judged elsewhere
    with pytest.raises(RuntimeError, match="different devices"):
        nv.check_same_device([Fake(0), Fake(1)])
    # no GPU tensor among the arguments: the wrapped function runs as is
    assert nv.on_tensor_device(lambda a, b: a + b)(1, 2) == 3
    assert nv.on_tensor_device(lambda t: t.sum().item())(torch.ones(3)) == 3.0


def test_row_shard_matches_the_c_side_numbering():
    """GsRowShard (include/gsplat_hip.h): local row of an owned tile row, period 1 and interleaved"""
    from taichi_gaussian_rasterizer_amd import _native as nv, parallel
    s = parallel.shard_for(1, 3, 250, 16, interleave=2)   # 16 tile rows, the last one partial
    assert s.rows() == [2, 3, 8, 9, 14, 15] and s.local_height == 5 * 16 + (250 - 15 * 16)
    assert s.bands == [(32, 64), (128, 160), (224, 250)]
    c = nv.make_shard(s)._obj
    assert (c.row_begin, c.row_end, c.band, c.period, c.phase) == (0, 16, 2, 3, 1)
    s = parallel.shard_for(2, 3, 250, 16)
    assert s.period == 1 and s.rows() == list(range(s.row_begin, s.row_end)) and s.bands[-1][1] == 250
