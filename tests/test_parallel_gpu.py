"""The fused frame under a real process group on ONE MI355X: two ranks (gloo; both on cuda:0 -- RCCL needs a GPU per
rank) render their strips through fused.py with the split gradient all-reduce; every rank must end with the
single-process gradients and the strips must tile the single-process image."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _scene(size, n):
    from taichi_gaussian_rasterizer_amd import scenes
    g, cam = scenes.benchmark_scene(n, size, sh_degree=3, seed=4)
    gi = torch.rand(size[1], size[0], 3, generator=torch.Generator().manual_seed(5))
    gd = torch.rand(size[1], size[0], generator=torch.Generator().manual_seed(6))
    return g, cam, gi, gd


def _run(rank, world, port, size, n, depth_mode, interleave, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from taichi_gaussian_rasterizer_amd import RasterConfig, parallel
        dev = torch.device("cuda:0")
        g, cam, gi, gd = _scene(size, n)
        g = g.to(dev).requires_grad_(True)
        cam = cam.to(device=dev)
        cfg = RasterConfig()
        r = parallel.render_gaussians_sharded(g, cam, cfg, use_sh=True, render_depth=depth_mode,
                                              interleave=interleave)
        rows = parallel.owned_pixel_rows(r.bands).to(dev)
        loss = (r.image * gi.to(dev)[rows]).sum()
        if depth_mode:
            loss = loss + (r.depth * gd.to(dev)[rows]).sum()
        loss.backward()
        full = parallel.gather_image(r.image.detach(), size[1], cfg.tile_size, interleave=interleave)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), image=full.cpu().numpy(),
                 **{f"d_{k}": v.grad.cpu().numpy() for k, v in g.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("depth_mode,interleave,world,size", [(False, 0, 2, (320, 256)), (True, 0, 2, (320, 256)),
                                                              (False, 2, 2, (320, 250)),
                                                              (True, 0, 3, (320, 30))])  # rank 2 owns no tile row
def test_fused_sharded_frame_two_ranks(tmp_path, depth_mode, interleave, world, size):
    import taichi_gaussian_rasterizer_amd as gs
    from taichi_gaussian_rasterizer_amd import RasterConfig
    n = 20000
    mp.spawn(_run, args=(world, _free_port(), size, n, depth_mode, interleave, str(tmp_path)), nprocs=world,
             join=True)
    dev = torch.device("cuda:0")
    g, cam, gi, gd = _scene(size, n)
    g = g.to(dev).requires_grad_(True)
    r = gs.render_gaussians(g, cam.to(device=dev), RasterConfig(), use_sh=True, render_depth=depth_mode)
    loss = (r.image * gi.to(dev)).sum()
    if depth_mode:
        loss = loss + (r.depth * gd.to(dev)).sum()
    loss.backward()
    for rank in range(world):
        z = np.load(tmp_path / f"rank{rank}.npz")
        assert np.array_equal(z["image"], r.image.detach().cpu().numpy())   # same kernels on exactly shifted means
        for k, v in g.items():
            ref = v.grad.cpu().numpy()
            err = np.linalg.norm(z[f"d_{k}"] - ref) / max(np.linalg.norm(ref), 1e-30)
            assert err < 1e-3, (rank, k, err)   # order of float sums differs (strips, collectives)
            # exact zeros stay exact zeros: culled Gaussians, and colour channels the forward clamped -- a rank that
            # did not rasterize a splat never evaluated its colour (gs_sh_fwd_shard), so the clamp mask reaches it
            # only through the pre-masked gradient sum (gs_shard_pack_grads)
            assert np.all(z[f"d_{k}"][ref == 0] == 0), (rank, k)
    clamped = (g.feature.grad[:, :, 0] == 0).any(dim=1) & (g.feature.grad[:, :, 0] != 0).any(dim=1)
    assert int(clamped.sum()) > 100  # the scene does exercise the mask: Gaussians with some channels clamped, some not


def _run_modes(rank, world, port, size, n, depth_mode, interleave, grad_mode, frames, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from taichi_gaussian_rasterizer_amd import RasterConfig, parallel
        dev = torch.device("cuda:0")
        g, cam, gi, gd = _scene(size, n)
        g = g.to(dev)
        cam = cam.to(device=dev)
        owned = None
        if grad_mode == "sharded":
            owned = parallel.split_owned(g, rank, world).requires_grad_(True)
        else:
            g = g.requires_grad_(True)
        holder = owned if owned is not None else g
        for _ in range(frames):  # the second frame of a shape goes through gs_frame_fwd (one call for the forward)
            for _, t in holder.items():
                t.grad = None
            r = parallel.render_gaussians_sharded(g, cam, RasterConfig(), use_sh=True, render_depth=depth_mode,
                                                  interleave=interleave, exchange="sparse", grad_mode=grad_mode,
                                                  owned=owned)
            rows = parallel.owned_pixel_rows(r.bands).to(dev)
            loss = (r.image * gi.to(dev)[rows]).sum()
            if depth_mode:
                loss = loss + (r.depth * gd.to(dev)[rows]).sum()
            loss.backward()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **{f"d_{k}": v.grad.cpu().numpy() for k, v in holder.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("depth_mode,interleave,world,size,grad_mode", [
    (False, 0, 2, (320, 256), "replicated"), (True, 2, 3, (320, 250), "replicated"),
    (True, 0, 3, (320, 30), "replicated"),            # rank 2 owns no tile row: an empty list
    (False, 0, 2, (320, 256), "sharded"), (True, 3, 3, (320, 250), "sharded")])
def test_fused_sharded_frame_sparse_exchange(tmp_path, depth_mode, interleave, world, size, grad_mode):
    """the fused frame's sparse exchange under a real process group (gloo ranks sharing the GPU): lists of the touched
    splats instead of dense rows; "replicated": every rank ends with the single-process gradients and all ranks with the
    SAME bits; "sharded": rank r ends with the complete gradients of its own index range, range-shaped, through one
    all-to-all and adjoints that run on that range only.  Two frames: staged forward, then gs_frame_fwd."""
    import taichi_gaussian_rasterizer_amd as gs
    from taichi_gaussian_rasterizer_amd import RasterConfig, parallel
    n = 20000
    mp.spawn(_run_modes, args=(world, _free_port(), size, n, depth_mode, interleave, grad_mode, 2, str(tmp_path)),
             nprocs=world, join=True)
    dev = torch.device("cuda:0")
    g, cam, gi, gd = _scene(size, n)
    g = g.to(dev).requires_grad_(True)
    r = gs.render_gaussians(g, cam.to(device=dev), RasterConfig(), use_sh=True, render_depth=depth_mode)
    loss = (r.image * gi.to(dev)).sum()
    if depth_mode:
        loss = loss + (r.depth * gd.to(dev)).sum()
    loss.backward()
    first = np.load(tmp_path / "rank0.npz")
    for rank in range(world):
        z = np.load(tmp_path / f"rank{rank}.npz")
        lo, hi = parallel.owned_range(rank, world, n) if grad_mode == "sharded" else (0, n)
        for k, v in g.items():
            ref = v.grad.cpu().numpy()[lo:hi]
            assert z[f"d_{k}"].shape == ref.shape
            err = np.linalg.norm(z[f"d_{k}"] - ref) / max(np.linalg.norm(ref), 1e-30)
            assert err < 1e-3, (rank, k, err)
            assert np.all(z[f"d_{k}"][ref == 0] == 0), (rank, k)   # culled Gaussians, clamped colour channels
            if grad_mode == "replicated":
                assert np.array_equal(z[f"d_{k}"], first[f"d_{k}"]), "replicas must not drift"
