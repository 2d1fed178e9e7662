"""Full-size parity (BASELINE.json configs C2 / C3 / C4 at their real sizes, and the C5 SHAPE: a 65 536-tile grid and
an 8-way row shard) -- HIP path vs the CPU oracle, stage by stage, asserted.

Stagewise means: the oracle is fed the HIP path's own projected splats / features, so f32 rounding of the
projection is not mixed into the rasterizer comparison (the projection and SH stages have their own golden-vector
tests).  Integer stages are compared with ==.  Pixels: every element must be inside the suite's tolerance OR be
PROVEN to be an `alpha > alpha_threshold` decision (reference rasterizer/forward.py:100) that the two f32
implementations round to different sides -- the oracle recomputes, per pixel, how close its walk comes to that
threshold (`orc.raster_flip_margin`); an outlier whose margin is not tiny, or whose size exceeds what flips can
cause, fails the test (`parity_util.assert_pixels_close` with a `FlipProof`).  No blanket outlier budget anywhere.
"""
import numpy as np
import pytest
import torch

import parity_util as pu
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

import taichi_gaussian_rasterizer_amd as gs  # noqa: E402
from taichi_gaussian_rasterizer_amd import RasterConfig, parallel, scenes  # noqa: E402
from taichi_gaussian_rasterizer_amd.perspective import projection as hip_proj  # noqa: E402

DEV = "cuda:0"

FULL = {
    "c2": dict(n=200_000, size=(1920, 1080), deg=0, backward=False, depth=False),
    "c3": dict(n=1_000_000, size=(2048, 2048), deg=3, backward=True, depth=False),
    "c4": dict(n=1_000_000, size=(2048, 2048), deg=3, backward=True, depth=True),
    # BASELINE config 5 on ONE device: the 6 M-Gaussian / 4096^2 frame the 8 ranks share (65 536 tiles -- the grid the
    # reference rejects, tile_mapper.py:29,175); the sharded form of the same frame is checked further down
    "c5": dict(n=6_000_000, size=(4096, 4096), deg=3, backward=True, depth=False),
}


@pytest.mark.parametrize("name", ["c2", "c3", "c4", "c5"])
def test_full_size_stagewise(name):
    wl = FULL[name]
    n, size = wl["n"], wl["size"]
    W, H = size
    orc.set_num_threads(__import__("os").cpu_count() or 1)
    g, camera = scenes.benchmark_scene(n, size, sh_degree=wl["deg"], seed=0)  # the bench.py scene
    cfg = RasterConfig()
    ocfg = orc.OracleConfig.of(cfg)
    gen = torch.Generator().manual_seed(1)
    gi = torch.rand(H, W, 3, generator=gen)
    gdm, gvm = torch.rand(H, W, generator=gen), torch.rand(H, W, generator=gen) * 0.1
    gd = g.to(DEV).requires_grad_(wl["backward"])
    cam = camera.to(device=DEV)

    # ---- HIP stages, holding on to the intermediates
    g2d, depths, idx, ndc = hip_proj.project_with_ndc(*gd.shape_tensors(), cam.T_camera_world, cam.projection,
                                                      cam.image_size, cam.depth_range, cfg)
    feats = gs.evaluate_sh_at(gd.feature, gd.position.detach(), idx, cam.camera_position)
    o2p, ranges = gs.map_to_tiles(g2d.detach(), ndc, size, cfg)
    chan = torch.cat([depths, depths * depths, feats], 1) if wl["depth"] else feats
    chan = chan.detach().requires_grad_(wl["backward"])
    p_t = g2d.detach().requires_grad_(wl["backward"])
    # the cut scale render_gaussians uses for this mode, so that the images below are the frame's own
    from dataclasses import replace
    rcfg = replace(cfg, forward_cut=cfg.forward_cut / max(camera.far_plane ** 2, 1.0)) if wl["depth"] else cfg
    raster = gs.rasterize_with_tiles(p_t, chan, o2p, ranges.view(-1, 2), size, rcfg)

    # ---- the fused frame is exactly this composition
    r2 = gs.render_gaussians(g.to(DEV), cam, cfg, use_sh=True, render_depth=wl["depth"])
    assert torch.equal(r2.points_in_view, idx) and torch.equal(r2.gaussians2d, g2d.detach())
    assert torch.equal(r2.image, raster.image[..., 2:] if wl["depth"] else raster.image)
    assert torch.equal(r2.image_weight, raster.image_weight)

    # ---- mapper: bit-exact at full size
    p_np, d_np, ndc_np, f_np = pu.to_np(g2d), pu.to_np(depths), pu.to_np(ndc), pu.to_np(chan)
    assert (ndc_np == orc.ndc_depth(d_np, camera.near_plane, camera.far_plane)).all()
    o2p_ref, ranges_ref = orc.map_to_tiles(p_np, ndc_np, size, ocfg)
    assert o2p_ref.shape[0] == o2p.shape[0]
    assert (pu.to_np(ranges) == ranges_ref).all() and (pu.to_np(o2p) == o2p_ref).all()

    # ---- rasterizer forward: tolerance or proven flip, per pixel
    image_ref, alpha_ref, _ = orc.rasterize_with_tiles(p_np, f_np, o2p_ref, ranges_ref, size, ocfg)
    proof = pu.flip_proof(p_np, f_np, o2p_ref, ranges_ref, size, ocfg)
    rep = pu.assert_pixels_close(raster.image, image_ref, f"{name} image", flips=proof, scale_atol=True)
    rep_w = pu.assert_pixels_close(raster.image_weight, alpha_ref, f"{name} image_weight", flips=proof.weight())
    print(f"\n{name}: V={p_np.shape[0]} K={o2p_ref.shape[0]} image {rep} weight {rep_w}")
    if not wl["backward"]:
        return

    # ---- rasterizer backward, given the HIP image (so a forward flip is not counted twice)
    if wl["depth"]:
        w_h = raster.image_weight + 1e-6
        d_h = raster.image[..., 0] / w_h
        g_img = torch.empty_like(raster.image)
        g_img[..., 2:] = gi.to(DEV)
        g_img[..., 0] = (gdm.to(DEV) - 2 * d_h * gvm.to(DEV)) / w_h
        g_img[..., 1] = gvm.to(DEV) / w_h
        g_img = g_img.detach()
    else:
        g_img = gi.to(DEV)
    raster.image.backward(g_img)
    gg, gf, _ = orc.rasterize_backward(p_np, f_np, o2p_ref, ranges_ref, size, pu.to_np(raster.image), pu.to_np(g_img),
                                       ocfg)
    pu.assert_grad_close(p_t.grad, gg, f"{name} d gaussians2d")
    pu.assert_grad_close(chan.grad, gf, f"{name} d features")
    # row-wise as well: a normwise bar alone would let a small splat's gradient be arbitrarily wrong
    pu.assert_rows_close(p_t.grad, gg, f"{name} d gaussians2d rows", tol=1e-2, frac=0.999)
    pu.assert_rows_close(chan.grad, gf, f"{name} d features rows", tol=1e-2, frac=0.999)


# ------------------------------------------------------------------------ C5 shape: >= 65 536 tiles
@pytest.mark.parametrize("size,depth16", [((4096, 4096), False), ((4096, 4096), True), ((4112, 4096), True),
                                          ((4112, 4096), False)])
def test_mapper_bit_exact_past_the_reference_tile_limit(size, depth16):
    """4096 x 4096 at tile 16 is 65 536 tiles -- the size the reference itself rejects (tile_mapper.py:29,175:
    `< 65535`, a 16-bit tile field in its 48- / 32-bit sort keys).  The build lifts the limit; the returned
    `sorted_keys` are u64 = tile_id << 32 | f32 depth bits (plain) or tile_id << 16 | 16-bit depth code
    (use_depth16), the tile id taking as many bits as it needs (17 for the 257 x 256 grid here), and the order is
    the reference's: by tile, then depth key, then Gaussian index."""
    from taichi_gaussian_rasterizer_amd.mapper.tile_mapper import map_to_tiles_reference_stages
    g2d, depth, _ = pu.make_2d_scene(21, 50_000, size, scale_factor=1.0, alpha_range=(0.01, 1.0))
    cfg = RasterConfig()
    tiles = (-(-size[0] // 16)) * (-(-size[1] // 16))
    assert tiles >= 65536
    o2p_ref, ranges_ref, keys_ref = orc.map_to_tiles(g2d, depth, size, orc.OracleConfig.of(cfg), depth16,
                                                     return_keys=True)
    shift = 16 if depth16 else 32
    assert int(keys_ref.max() >> shift) >= 65535 - 512  # the scene reaches the last tiles
    for fn in (gs.map_to_tiles, map_to_tiles_reference_stages):
        o2p, ranges, keys = fn(g2d.to(DEV), depth.to(DEV), size, cfg, use_depth16=depth16, return_keys=True)
        assert tuple(ranges.shape) == ranges_ref.shape
        assert (pu.to_np(ranges) == ranges_ref).all(), f"{fn.__name__}: tile ranges differ"
        assert (pu.to_np(o2p) == o2p_ref).all(), f"{fn.__name__}: overlap order differs"
        assert (pu.to_np(keys).view(np.uint64) == keys_ref).all(), f"{fn.__name__}: sort keys differ"


@pytest.mark.parametrize("n,interleave", [(300_000, 0), (300_000, 3), (6_000_000, 0), (6_000_000, 3)])
def test_eight_way_shard_at_c5_image_size(n, interleave):
    """the per-rank work of an 8-GPU frame at 4096 x 4096 (C5's image; with 300 k Gaussians and with C5's own 6 M --
    seed 0, the frame test_full_size_stagewise["c5"] holds against the oracle), ranks emulated one after another on one
    GPU: the ranks' rows tile the unsharded image bit for bit, `gaussians2d` is the same full-image tensor on every
    rank, and the partial gradients sum to the full ones."""
    size, world = (4096, 4096), 8
    g, camera = scenes.benchmark_scene(n, size, sh_degree=3, seed=0 if n > 1_000_000 else 2)
    cam = camera.to(device=DEV)
    cfg = RasterConfig()
    gi = torch.rand(size[1], size[0], 3, generator=torch.Generator().manual_seed(9)).to(DEV)
    full = g.to(DEV).requires_grad_(True)
    r = gs.render_gaussians(full, cam, cfg, use_sh=True)
    (r.image * gi).sum().backward()
    canvas = torch.full_like(r.image, float("nan"))
    sums = None
    for rank in range(world):
        gr = g.to(DEV).requires_grad_(True)
        rr = parallel.render_gaussians_sharded(gr, cam, cfg, use_sh=True, rank=rank, world_size=world,
                                               interleave=interleave)
        rows = parallel.owned_pixel_rows(rr.bands).to(DEV)
        assert rr.image.shape[0] == rows.shape[0]
        assert torch.equal(rr.gaussians2d, r.gaussians2d), "gaussians2d must be in full-image coordinates"
        (rr.image * gi[rows]).sum().backward()
        canvas[rows] = rr.image.detach()
        grads = {k: v.grad.clone() for k, v in gr.items()}
        sums = grads if sums is None else {k: sums[k] + grads[k] for k in sums}
    assert torch.equal(canvas, r.image.detach()), "the ranks' rows do not tile the unsharded image exactly"
    for k, v in full.items():
        # eight partial sums per Gaussian instead of one: the eigen-decomposition adjoint amplifies the different f32
        # rounding of near-isotropic splats (measured normwise 7e-4 on log_scaling with interleaved bands), so the
        # normwise bar is 2e-3 and the row-wise one says that it is a handful of rows
        pu.assert_grad_close(sums[k], v.grad, f"summed partial grad {k}", tol=2e-3)
        pu.assert_rows_close(sums[k], v.grad, f"summed partial grad rows {k}", tol=1e-3, frac=0.999)


def test_viewspace_gradient_on_the_fused_frame(frame_path):
    """`gaussians2d.retain_grad()` + `viewspace_gradient` (reference renderer.py:234-239: the classic densification
    signal) on the default, fused path equals the composed operators' value"""
    from taichi_gaussian_rasterizer_amd.renderer import render_projected, viewspace_gradient
    size, n = (320, 240), 20_000
    g, camera = scenes.benchmark_scene(n, size, sh_degree=2, seed=5)
    cam = camera.to(device=DEV)
    cfg = RasterConfig()
    gi = torch.rand(size[1], size[0], 3, generator=torch.Generator().manual_seed(3)).to(DEV)
    gd = g.to(DEV).requires_grad_(True)
    r = gs.render_gaussians(gd, cam, cfg, use_sh=True)
    r.gaussians2d.retain_grad()
    (r.image * gi).sum().backward()
    fused = viewspace_gradient(r.gaussians2d)

    gc = g.to(DEV).requires_grad_(True)
    g2d, depths, idx, ndc = hip_proj.project_with_ndc(*gc.shape_tensors(), cam.T_camera_world, cam.projection,
                                                      cam.image_size, cam.depth_range, cfg)
    feats = gs.evaluate_sh_at(gc.feature, gc.position.detach(), idx, cam.camera_position)
    g2d.retain_grad()
    rc = render_projected(idx, g2d, feats, depths, cam, cfg, ndc_depths=ndc)
    (rc.image * gi).sum().backward()
    composed = viewspace_gradient(g2d)
    assert fused.shape == composed.shape and float(composed.max()) > 0
    pu.assert_grad_close(fused, composed, "viewspace gradient", tol=1e-4)
    # and the parameter gradients are unaffected by the publication
    for k, v in gd.items():
        pu.assert_grad_close(v.grad, getattr(gc, k).grad, f"grad {k}", tol=1e-4)


def test_gaussians2d_grad_accumulates_over_backward_passes(frame_path):
    """two losses backpropagated one after the other through the same Rendering, one of them attached to the projected
    splats themselves: `gaussians2d.grad` ends up as the SUM, as on the reference's composed graph (where the tensor is
    an ordinary non-leaf with retain_grad); an empty view publishes a (0, 7) gradient instead of none"""
    from taichi_gaussian_rasterizer_amd.renderer import render_projected
    size, n = (160, 120), 5_000
    g, camera = scenes.benchmark_scene(n, size, sh_degree=1, seed=6)
    cam = camera.to(device=DEV)
    cfg = RasterConfig()
    gen = torch.Generator().manual_seed(4)
    gi, gj = (torch.rand(size[1], size[0], 3, generator=gen).to(DEV) for _ in range(2))

    def two_passes(render):
        r, g2d = render()
        g2d.retain_grad()
        (r.image * gi).sum().backward(retain_graph=True)
        ((r.image * gj).sum() + (g2d[:, :2] ** 2).sum() * 1e-6).backward()
        return g2d.grad.clone()

    def fused():
        r = gs.render_gaussians(g.to(DEV).requires_grad_(True), cam, cfg, use_sh=True)
        return r, r.gaussians2d

    def composed():
        gc = g.to(DEV).requires_grad_(True)
        g2d, depths, idx, ndc = hip_proj.project_with_ndc(*gc.shape_tensors(), cam.T_camera_world, cam.projection,
                                                          cam.image_size, cam.depth_range, cfg)
        feats = gs.evaluate_sh_at(gc.feature, gc.position.detach(), idx, cam.camera_position)
        return render_projected(idx, g2d, feats, depths, cam, cfg, ndc_depths=ndc), g2d

    pu.assert_grad_close(two_passes(fused), two_passes(composed), "accumulated gaussians2d.grad", tol=1e-4)

    # nothing in view: every Gaussian behind the camera
    behind = g.to(DEV)
    behind.position[:, 2] = -behind.position[:, 2].abs() - 1.0
    r = gs.render_gaussians(behind.requires_grad_(True), cam, cfg, use_sh=True)
    assert r.gaussians2d.shape[0] == 0
    r.gaussians2d.retain_grad()
    r.image.sum().backward()
    assert r.gaussians2d.grad is not None and tuple(r.gaussians2d.grad.shape) == (0, 7)


@pytest.mark.parametrize("cut", [0.0, 2.0 ** -20])
def test_forward_cut_is_a_config_field(cut):
    """forward_cut = 0 reproduces the reference's forward literally (no region is ever abandoned while anything can
    still change): == the oracle within the plain pixel tolerance even on a scene that saturates every tile, and the
    default cut stays inside forward_cut * max|feature| of it."""
    size, n = (128, 128), 30_000  # ~1800 splats per tile: deep saturation
    g2d, depth, feat = pu.make_2d_scene(7, n, size, scale_factor=1.5, alpha_range=(0.5, 0.99))
    feat = feat * 50.0  # large features: the dropped tail scales with them
    cfg = RasterConfig(forward_cut=cut)
    ocfg = orc.OracleConfig.of(cfg)
    o2p_ref, ranges_ref = orc.map_to_tiles(g2d, depth, size, ocfg)
    image_ref, alpha_ref, _ = orc.rasterize_with_tiles(g2d.numpy(), feat.numpy(), o2p_ref, ranges_ref, size, ocfg)
    out = gs.rasterize(g2d.to(DEV), depth.to(DEV), feat.to(DEV), size, cfg)
    exact = gs.rasterize(g2d.to(DEV), depth.to(DEV), feat.to(DEV), size, RasterConfig(forward_cut=0.0))
    if cut == 0.0:
        proof = pu.flip_proof(g2d, feat, o2p_ref, ranges_ref, size, ocfg)
        pu.assert_pixels_close(out.image, image_ref, "forward_cut=0 image", flips=proof, scale_atol=True)
    diff = (out.image - exact.image).abs().max().item()
    assert diff <= cut * float(np.abs(feat.numpy()).max()) * 1.01, (diff, cut)
