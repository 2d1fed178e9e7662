"""Tile-strip sharding + gradient all-reduce (taichi_gaussian_rasterizer_amd/parallel.py) on CPU:
world_size 2 and 3 over gloo, stage operators backed by the CPU oracle.  Checks that the strips
tile the single-process image exactly and that every rank ends with the single-process gradients."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_ops
from taichi_gaussian_rasterizer_amd import RasterConfig, parallel, scenes


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _scene(size, n, depth_mode):
    g, cam = scenes.benchmark_scene(n, size, sh_degree=2, seed=3)
    gi = torch.rand(size[1], size[0], 3, generator=torch.Generator().manual_seed(5))
    gdm = torch.rand(size[1], size[0], generator=torch.Generator().manual_seed(6)) if depth_mode else None
    return g, cam, gi, gdm


def _run(rank, world, port, size, n, depth_mode, interleave, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g, cam, gi, gdm = _scene(size, n, depth_mode)
        g = g.requires_grad_(True)
        cfg = RasterConfig(compute_visibility=True, compute_point_heuristic=True)
        r = parallel.render_gaussians_sharded(g, cam, cfg, use_sh=True, render_depth=depth_mode, ops=oracle_ops.OPS,
                                              interleave=interleave)
        rows = parallel.owned_pixel_rows(r.bands)
        assert r.image.shape[0] == rows.shape[0] == r.shard.local_height
        loss = (r.image * gi[rows]).sum()
        if depth_mode:
            loss = loss + (r.depth * gdm[rows]).sum()
        loss.backward()
        full = parallel.gather_image(r.image.detach(), size[1], cfg.tile_size, interleave=interleave)
        vis, heur = parallel.reduce_point_statistics(r)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), image=full.numpy(), rows=rows.numpy(),
                 vis=vis.numpy(), heur=heur.numpy(),
                 **{f"d_{k}": v.grad.numpy() for k, v in g.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,size,n,depth_mode,interleave", [
    (2, (96, 80), 1500, False, 0), (3, (64, 112), 800, True, 0),
    (2, (96, 80), 1500, False, 1),     # interleaved: single tile rows dealt round-robin
    (3, (64, 112), 800, True, 2),      # bands of two tile rows, the last band partial
    (3, (64, 30), 400, True, 0),       # more ranks than tile rows: rank 2 owns nothing and still joins the collectives
])
def test_sharded_render_matches_single_process(tmp_path, world, size, n, depth_mode, interleave):
    port = _free_port()
    mp.spawn(_run, args=(world, port, size, n, depth_mode, interleave, str(tmp_path)), nprocs=world, join=True)

    g, cam, gi, gdm = _scene(size, n, depth_mode)
    g = g.requires_grad_(True)
    cfg = RasterConfig(compute_visibility=True, compute_point_heuristic=True)
    r = parallel.render_gaussians_sharded(g, cam, cfg, use_sh=True, render_depth=depth_mode, ops=oracle_ops.OPS,
                                          rank=0, world_size=1)
    loss = (r.image * gi).sum()
    if depth_mode:
        loss = loss + (r.depth * gdm).sum()
    loss.backward()
    covered = []
    for rank in range(world):
        z = np.load(tmp_path / f"rank{rank}.npz")
        covered.append(z["rows"])
        # strips are rendered by the same arithmetic on exactly shifted coordinates
        assert np.allclose(z["image"], r.image.detach().numpy(), rtol=0, atol=1e-6)
        # per-strip visibility / heuristics summed over the ranks = the single-process statistics
        assert np.allclose(z["vis"], r.point_visibility.numpy(), rtol=1e-4, atol=1e-5)
        assert np.allclose(z["heur"], r.point_heuristic.numpy(), rtol=1e-3, atol=1e-4 * max(1.0, float(r.point_heuristic.abs().max())))
        for k, v in g.items():
            ref = v.grad.numpy()
            assert np.allclose(z[f"d_{k}"], ref, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(ref).max())), (rank, k)
    # the ranks' rows partition the image
    assert sorted(np.concatenate(covered).tolist()) == list(range(size[1]))


def _run_modes(rank, world, port, size, n, depth_mode, interleave, exchange, grad_mode, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g, cam, gi, gdm = _scene(size, n, depth_mode)
        cfg = RasterConfig()
        owned = None
        if grad_mode == "sharded":
            owned = parallel.split_owned(g, rank, world).requires_grad_(True)
        else:
            g = g.requires_grad_(True)
        r = parallel.render_gaussians_sharded(g, cam, cfg, use_sh=True, render_depth=depth_mode, ops=oracle_ops.OPS,
                                              interleave=interleave, exchange=exchange, grad_mode=grad_mode, owned=owned)
        rows = parallel.owned_pixel_rows(r.bands)
        loss = (r.image * gi[rows]).sum()
        if depth_mode:
            loss = loss + (r.depth * gdm[rows]).sum()
        loss.backward()
        holder = owned if owned is not None else g
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **{f"d_{k}": v.grad.numpy() for k, v in holder.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,size,n,depth_mode,interleave,exchange,grad_mode", [
    (2, (96, 80), 1500, False, 0, "sparse", "replicated"),
    (3, (64, 112), 800, True, 2, "sparse", "replicated"),
    (3, (64, 30), 400, True, 0, "sparse", "replicated"),   # rank 2 owns no tile row: an empty list, same collectives
    (2, (96, 80), 1500, False, 1, "sparse", "sharded"),
    (3, (64, 112), 803, True, 0, "sparse", "sharded"),      # 803 Gaussians over 3 owners: ranges 268 / 268 / 267
])
def test_sparse_exchange_and_sharded_gradients(tmp_path, world, size, n, depth_mode, interleave, exchange, grad_mode):
    """exchange="sparse": every rank ends with the single-process gradients, and with the SAME ones bit for bit (the
    lists are added in rank order); grad_mode="sharded": rank r ends with the complete gradients of the Gaussians
    owned_range(r) -- range-shaped, on its `owned` leaf tensors -- through one all-to-all."""
    mp.spawn(_run_modes, args=(world, _free_port(), size, n, depth_mode, interleave, exchange, grad_mode,
                               str(tmp_path)), nprocs=world, join=True)
    g, cam, gi, gdm = _scene(size, n, depth_mode)
    g = g.requires_grad_(True)
    r = parallel.render_gaussians_sharded(g, cam, RasterConfig(), use_sh=True, render_depth=depth_mode,
                                          ops=oracle_ops.OPS, rank=0, world_size=1)
    loss = (r.image * gi).sum()
    if depth_mode:
        loss = loss + (r.depth * gdm).sum()
    loss.backward()
    first = np.load(tmp_path / "rank0.npz")
    for rank in range(world):
        z = np.load(tmp_path / f"rank{rank}.npz")
        lo, hi = parallel.owned_range(rank, world, n) if grad_mode == "sharded" else (0, n)
        for k, v in g.items():
            ref = v.grad.numpy()[lo:hi]
            assert z[f"d_{k}"].shape == ref.shape
            assert np.allclose(z[f"d_{k}"], ref, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(v.grad.numpy()).max())), (rank, k)
            if grad_mode == "replicated":
                assert np.array_equal(z[f"d_{k}"], first[f"d_{k}"]), "replicas must not drift"
    assert [parallel.owned_range(r, 3, 803) for r in range(3)] == [(0, 268), (268, 536), (536, 803)]


def test_exchanged_bytes_accounting():
    """the bench line's exchange figures are arithmetic on list lengths: dense ring all-reduce vs the two sparse modes"""
    V, F, world = 6_000_000, 3, 8
    touched = [int(V / world * 1.15)] * world
    dense = parallel.exchanged_bytes("dense", "replicated", world, V, touched, F)
    assert dense["payload"] == 40 * V and dense["sent"] == int(2 * 7 / 8 * 40 * V)
    sp = parallel.exchanged_bytes("sparse", "replicated", world, V, touched, F)
    sh = parallel.exchanged_bytes("sparse", "sharded", world, V, touched, F)
    assert sp["sent"] == 44 * touched[0] and sp["received"] == 44 * touched[0] * 7
    assert sh["sent"] == int(44 * touched[0] * 7 / 8) and sh["sent"] < 0.1 * dense["sent"]
    assert sp["received"] < 0.75 * dense["received"]


def test_strip_partition():
    for world in (1, 2, 3, 8):
        for rows in (1, 7, 8, 128, 129):
            spans = [parallel.strip_rows(r, world, rows) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == rows
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_row_shards_partition_the_tile_rows():
    for world in (1, 2, 3, 8):
        for height, ts in ((16, 16), (100, 16), (2048, 16), (129, 8), (30, 16)):
            rows = parallel.tile_rows(height, ts)
            for interleave in (0, 1, 2, 5):
                shards = [parallel.shard_for(r, world, height, ts, interleave) for r in range(world)]
                owned = sorted(ty for s in shards for ty in s.rows())
                assert owned == list(range(rows)), (world, height, ts, interleave)
                assert sum(s.local_height for s in shards) == height
                px = sorted(int(y) for s in shards for y in parallel.owned_pixel_rows(s.bands))
                assert px == list(range(height))
                for s in shards:  # local row numbering = position in the ascending list of owned rows
                    for local, ty in enumerate(s.rows()):
                        if s.period > 1:
                            assert (ty // (s.band * s.period)) * s.band + ty % s.band == local
                        else:
                            assert ty - s.row_begin == local
