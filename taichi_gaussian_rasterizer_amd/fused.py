"""Fused frame: the whole render_gaussians pipeline as ONE autograd node.

Same kernels as the operator-by-operator composition in renderer.py, but enqueued back to back
on the stream with the intermediate counts (visible Gaussians V, overlaps K) left on the device:

  * no host read-back between the stages -- the single synchronisation is at the END of the forward,
    when the tensor shapes of the result (V rows) have to be known to Python;
  * no torch glue on the path: the SH colours and the depth features are written straight into the
    rasterizer's feature rows, the rasterizer's 64-byte gradient rows are consumed in place by the SH
    and projection backward (no unpack, no cat/index backward, no zero-filled dense temporaries);
  * the pair / overlap buffers are sized from the previous frame's K (x1.3); the mapper clamps to the
    capacity and raises a flag, in which case the frame is re-run once with exact sizes.

This is SURVEY.md 8(f)-1: in the reference that glue is ~24 % of the forward+backward GPU time
(profiles/bicycle_2048.txt:38,42,44,47).  Results are bit-identical to the composed operators.
"""
from __future__ import annotations

import ctypes

import torch

from . import _native as nv
from .data_types import RasterConfig
from .spherical_harmonics import check_sh_degree

_K_HINT = {}  # (n, w, h, tile_size, use_depth16) -> (max overlaps, max tile population) seen for that shape
# True: gs_frame_fwd / gs_frame_bwd for every frame but the first of its shape; False: always the stages; "always":
# the frame calls also for a first frame, after an untracked sizing pass (tests/conftest.py frame_path)
FRAME_CALLS = True
_PINNED = {}  # device index -> ring of pinned int32[8] host buffers for the asynchronous count read-back


def _pinned_counts(dev: torch.device) -> torch.Tensor:
    ring = _PINNED.get(dev.index)
    if ring is None:
        ring = _PINNED[dev.index] = dict(bufs=[torch.empty((8,), dtype=torch.int32).pin_memory() for _ in range(4)],
                                         at=0)
    ring["at"] = (ring["at"] + 1) % len(ring["bufs"])
    return ring["bufs"][ring["at"]]


def _off(t: torch.Tensor, floats: int) -> ctypes.c_void_p:
    return ctypes.c_void_p(t.data_ptr() + 4 * floats)


class _FusedRender(torch.autograd.Function):
    @staticmethod
    @nv.on_tensor_device
    def forward(ctx, position, log_scaling, rotation, alpha_logit, feature, T_camera_world, projection,
                image_size, depth_range, config: RasterConfig, render_depth: bool, use_depth16: bool,
                render_median: bool = False, shard=None, group=None, holder=None, exchange: str = "dense",
                grad_mode: str = "replicated", owned_range=None):
        nv.require_device(position, log_scaling, rotation, alpha_logit, feature, T_camera_world, projection,
                          what="render_gaussians")
        lib = nv.lib()
        dev = position.device
        n = position.shape[0]
        w, full_h = int(image_size[0]), int(image_size[1])
        # sharded frame (parallel.RowShard): everything stays in full-image coordinates; the mapper and the
        # rasterizer skip the tile rows this rank does not own and the images hold the owned pixel rows only
        h = full_h if shard is None else shard.local_height
        sh = nv.make_shard(shard)
        C = feature.shape[1]
        degree = check_sh_degree(feature) if feature.dim() == 3 else -1  # -1: plain (N, C) features, no SH
        F = C + (2 if render_depth else 0)
        col0 = F - C
        cfg = nv.make_config(config)
        # what the forward's early stop may drop is bounded by forward_cut * max|feature|: z^2 reaches far^2
        rcfg = nv.make_config(config, cut_scale=float(depth_range[1]) ** 2) if render_depth else cfg
        ts = config.tile_size
        tile_shape = (-(-h // ts), -(-w // ts))
        num_tiles = tile_shape[0] * tile_shape[1]
        T = T_camera_world.contiguous()
        proj = projection.contiguous()
        f32 = dict(dtype=torch.float32, device=dev)

        points = torch.empty((n, 7), **f32)
        depth = torch.empty((n, 1), **f32)
        ndc = torch.empty((n, 1), **f32)
        feats = torch.empty((n, F), **f32)
        indexes = torch.empty((n,), dtype=torch.int64, device=dev)
        slot_of = torch.empty((n,), dtype=torch.int32, device=dev)
        # [0] = V (projection) ; [4:8] = K, fullest tile, overflow flag, heavy tiles (mapper scan): every word that is
        # read is written by a kernel first, so no fill launch
        counts = torch.empty((8,), dtype=torch.int32, device=dev)
        cam_pos = torch.empty((3,), **f32)
        pbytes = lib.gs_project_scratch_bytes(n)
        pscratch = torch.empty((max(pbytes, 1),), dtype=torch.uint8, device=dev)
        s = nv.stream()
        nv.check(lib.gs_project_fwd(n, nv.ptr(position), nv.ptr(log_scaling), nv.ptr(rotation), nv.ptr(alpha_logit),
                                    nv.ptr(T), nv.ptr(proj), w, full_h, float(depth_range[0]), float(depth_range[1]),
                                    cfg, nv.ptr(points), nv.ptr(depth), nv.ptr(ndc), nv.ptr(indexes),
                                    nv.ptr(slot_of), nv.ptr(counts), nv.ptr(feats) if render_depth else None, F,
                                    nv.ptr(cam_pos), nv.ptr(pscratch), pbytes, s), "gs_project_fwd")
        v_dev = nv.ptr(counts)
        if degree >= 0 and shard is not None:
            # a rank evaluates the colours of the splats that can reach its rows only (the replicated per-Gaussian
            # stages are what bounds the scaling of a sharded frame); the other rows get the neutral 0.5
            nv.check(lib.gs_sh_fwd_shard(n, v_dev, C, degree, nv.ptr(feature), nv.ptr(position), nv.ptr(indexes),
                                         nv.ptr(cam_pos), nv.ptr(points), full_h, cfg, sh,
                                         _off(feats, col0), F, s), "gs_sh_fwd_shard")
        elif degree >= 0:
            nv.check(lib.gs_sh_fwd(n, v_dev, C, degree, nv.ptr(feature), nv.ptr(position), nv.ptr(indexes),
                                   nv.ptr(cam_pos), _off(feats, col0), F, s), "gs_sh_fwd")
        else:
            nv.check(lib.gs_feature_gather_fwd(n, v_dev, C, nv.ptr(feature), nv.ptr(indexes), _off(feats, col0), F, s),
                     "gs_feature_gather_fwd")

        tile_ranges = torch.empty((*tile_shape, 2), dtype=torch.int32, device=dev)
        tile_order = torch.empty((num_tiles,), dtype=torch.int32, device=dev)  # heaviest tiles first
        mbytes = lib.gs_map_scratch_bytes(n, max(num_tiles, 1))
        mscratch = torch.empty((mbytes,), dtype=torch.uint8, device=dev)
        want_vis = config.compute_visibility or config.compute_point_heuristic
        key = (n, w, full_h, shard, ts, bool(use_depth16))
        hint = _K_HINT.get(key)
        k_cap = 0 if hint is None else int(hint[0] * 1.25) + 4096
        tile_hint = 0 if hint is None else -max(int(hint[1]), 1)  # sizing hint only; fuller tiles are still sorted
        host_counts = _pinned_counts(dev)
        ready = torch.cuda.Event()

        def no_rows():
            # this rank owns no tile row (more ranks than rows): nothing to map or rasterize, only V is needed
            counts[4:8] = 0
            host_counts[:4].copy_(counts[4:8], non_blocking=True)
            host_counts[4:5].copy_(counts[0:1], non_blocking=True)
            host_counts[5:6].copy_(counts[5:6], non_blocking=True)  # touched splats: none
            ready.record()
            return (torch.empty((0,), dtype=torch.int32, device=dev), torch.empty((0, w, F), **f32),
                    torch.empty((0, w), **f32), torch.zeros((n,), **f32) if want_vis else None)

        def map_and_raster(k_cap):
            if num_tiles == 0:
                return no_rows()
            nv.check(lib.gs_map_prepare(n, v_dev, nv.ptr(points), w, full_h, cfg, k_cap, nv.ptr(tile_ranges),
                                        _off(counts, 4), nv.ptr(host_counts), nv.ptr(tile_order), sh,
                                        nv.ptr(mscratch), mbytes, s), "gs_map_prepare")
            # K, the overflow flag and V are final here and the scan kernel has stored them into the pinned host words
            # itself (no copy launch): the host waits on this event while the sort and the rasterizer are still running
            ready.record()
            if k_cap == 0:  # first frame of this shape: K has to be known to size the buffers
                ready.synchronize()
                k_cap = max(int(host_counts[0]), 1)
            o2p = torch.empty((k_cap,), dtype=torch.int32, device=dev)
            pairs = torch.empty((k_cap,), dtype=torch.int64, device=dev)
            nv.check(lib.gs_map_finish(n, v_dev, k_cap, tile_hint, nv.ptr(points), nv.ptr(ndc), w, full_h, cfg,
                                       int(use_depth16),
                                       nv.ptr(tile_ranges), nv.ptr(o2p), None, nv.ptr(pairs), sh, nv.ptr(mscratch),
                                       mbytes, s), "gs_map_finish")
            image = torch.empty((h, w, F), **f32)
            alpha = torch.empty((h, w), **f32)
            vis = torch.zeros((n,), **f32) if want_vis else None
            nv.check(lib.gs_raster_fwd(n, F, nv.ptr(points), nv.ptr(feats), nv.ptr(tile_ranges), nv.ptr(o2p), k_cap,
                                       w, full_h, rcfg, nv.ptr(tile_order), _off(counts, 7), nv.ptr(image),
                                       nv.ptr(alpha), nv.ptr(vis), sh, s),
                     "gs_raster_fwd")
            return o2p, image, alpha, vis

        o2p, image, alpha, vis = map_and_raster(k_cap)
        ready.synchronize()  # waits for the mapper's scan only, not for the rasterizer
        host = host_counts.tolist()
        K, max_tile, overflow, V = host[0], host[1], host[2], host[4]  # [3] = heavy tiles, device-side only
        if overflow:  # more overlaps than the hint allowed for: run the tail again with exact sizes
            o2p, image, alpha, vis = map_and_raster(max(K, 1))
            ready.synchronize()
        _K_HINT[key] = (max(K, hint[0]) if hint else K, max(max_tile, hint[1]) if hint else max_tile)

        # render_depth: per-pixel epilogue (depth, depth variance, feature slice) in one pass
        img_depth = img_var = None
        out_image = image
        if render_depth:
            out_image = torch.empty((h, w, C), **f32)
            img_depth, img_var = torch.empty((h, w), **f32), torch.empty((h, w), **f32)
            nv.check(lib.gs_depth_split_fwd(h * w, C, nv.ptr(image), nv.ptr(alpha), 1e-6, nv.ptr(out_image),
                                            nv.ptr(img_depth), nv.ptr(img_var), s), "gs_depth_split_fwd")
        else:
            img_depth = img_var = torch.empty((0,), **f32)

        # render_median_depth: a second, non-blended forward over the same tile lists that keeps the depth of the splat
        # taking each pixel past half opacity (reference renderer.py:203-208); no gradient
        median = torch.empty((0,), **f32)
        if render_median and num_tiles == 0:
            median = torch.empty((0, w), **f32)
        if render_median and num_tiles > 0:
            from dataclasses import replace as _replace
            pick = nv.make_config(_replace(config, use_alpha_blending=False, saturate_threshold=0.5,
                                           compute_visibility=False, compute_point_heuristic=False))
            median, covered = torch.empty((h, w), **f32), torch.empty((h, w), **f32)
            nv.check(lib.gs_raster_fwd(n, 1, nv.ptr(points), nv.ptr(depth), nv.ptr(tile_ranges), nv.ptr(o2p),
                                       o2p.shape[0], w, full_h, pick, nv.ptr(tile_order), _off(counts, 7),
                                       nv.ptr(median), nv.ptr(covered), None, sh, s), "gs_raster_fwd")

        points_v, depth_v, indexes_v = points[:V], depth[:V], indexes[:V]
        empty = torch.empty((0,), **f32)
        vis_out = vis[:V] if config.compute_visibility else empty
        heur = torch.zeros((V, 2), **f32) if config.compute_point_heuristic else torch.empty((0, 2), **f32)

        ctx.meta = dict(n=n, V=V, K=K, w=w, h=h, full_h=full_h, F=F, C=C, col0=col0, degree=degree, config=config,
                        render_depth=render_depth, group=group, shard=shard, far=float(depth_range[1]),
                        exchange=exchange, grad_mode=grad_mode, owned_range=owned_range)
        if shard is not None and exchange == "sparse":
            ctx.meta["v_dev"] = counts
            _start_sparse_exchange(ctx.meta, n, num_tiles, host[5] if num_tiles > 0 else 0, indexes, mscratch, 0)
        if holder is not None and shard is not None:
            holder["touched_count"] = int(host[5]) if num_tiles > 0 else 0
        ctx.camera_grads = (ctx.needs_input_grad[5], ctx.needs_input_grad[6])
        ctx.heur = heur
        ctx.holder = holder
        # outputs nobody differentiates through (projected splats, depths) must not cost zero-filled gradients
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(position, log_scaling, rotation, alpha_logit, feature, T, proj, points, feats, slot_of,
                              indexes, cam_pos, tile_ranges, o2p, image, alpha, img_depth, tile_order, counts)
        ctx.mark_non_differentiable(alpha, indexes_v, vis_out, heur, median)
        if not render_depth:
            ctx.mark_non_differentiable(img_depth, img_var)
        return out_image, alpha, points_v, depth_v, indexes_v, vis_out, heur, img_depth, img_var, median

    @staticmethod
    @nv.on_tensor_device
    def backward(ctx, g_image, _g_alpha, g_points, g_depth, _g_idx, _g_vis, _g_heur, g_img_depth, g_img_var,
                 _g_median=None):
        return _backward_stages(ctx, ctx.saved_tensors, g_image, g_points, g_depth, g_img_depth, g_img_var) + (None,) * 12


def _exchange_ranks(shard, exchange, group, n, owned_range):
    """(world, rank) of a sharded frame's sparse exchange, (0, 0) when there is none"""
    if shard is None or exchange != "sparse":
        return 0, 0
    import torch.distributed as dist

    from . import parallel
    if dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    world = max(parallel.EMULATED_WORLD, 1)
    rank = 0 if owned_range is None else owned_range[0] // max(-(-n // world), 1)
    return world, rank


def _start_sparse_exchange(meta, n, num_tiles, touched_count, indexes, scratch, scratch_offset, prepared=None):
    """Bookkeeping of a sharded frame's sparse exchange, done during the FORWARD: keep the mapper's list of the splats
    that can reach this rank's rows (M int32 rows, left in the mapper scratch) and start the all-gather of the list
    lengths (grad_mode "sharded": of the per-owner counts, the list sorted by Gaussian index so that the entries of one
    owner are contiguous).  The backward finds both in `meta`."""
    import torch.distributed as dist

    from . import parallel
    group = meta["group"]
    world = dist.get_world_size(group) if dist.is_initialized() else max(parallel.EMULATED_WORLD, 1)
    M = int(touched_count)
    lib = nv.lib()
    sharded = meta["grad_mode"] == "sharded"
    rank = dist.get_rank(group) if dist.is_initialized() else \
        (0 if meta.get("owned_range") is None else meta["owned_range"][0] // max(-(-n // world), 1))
    if prepared is not None:  # gs_frame_fwd did it (and evaluated the colours of exactly these rows)
        touched = prepared["touched"]
        counts = prepared["owner_counts"] if sharded else torch.full((1,), M, dtype=torch.int64, device=touched.device)
        meta["owned_rows"] = prepared["owned_rows"] if sharded else None
    else:
        dev = indexes.device
        # the mapper's list is grouped by screen region; ascending rows (= ascending Gaussian index) make the exchange
        # kernels walk memory forwards and put the rows of one owner rank next to each other
        touched = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
        counts = torch.empty((world if sharded else 1,), dtype=torch.int64, device=dev)
        owned_rows = torch.zeros((2,), dtype=torch.int32, device=dev) if sharded else None
        mbytes = lib.gs_map_scratch_bytes(n, max(num_tiles, 1))
        if num_tiles > 0 and M > 0:
            nv.check(lib.gs_map_touched_list(n, nv.ptr(meta["v_dev"]), max(num_tiles, 1),
                                             ctypes.c_void_p(scratch.data_ptr() + scratch_offset), mbytes,
                                             nv.ptr(touched), None, nv.ptr(indexes), n, world,
                                             nv.ptr(counts) if sharded else None, rank, nv.ptr(owned_rows),
                                             nv.stream()), "gs_map_touched_list")
        elif sharded:
            counts.zero_()
        if not sharded:
            counts.fill_(M)
        touched = touched[:M]
        meta["owned_rows"] = owned_rows if (sharded and num_tiles > 0 and M > 0) else None
    meta["touched"] = touched
    meta["touched_count"] = M
    meta["sizes"] = parallel.SizesFuture(counts, group) if world > 1 else None
    meta["emulated_rank"] = 0 if meta.get("owned_range") is None else meta["owned_range"][0] // max(-(-n // world), 1)


def _backward_stages(ctx, saved, g_image, g_points, g_depth, g_img_depth, g_img_var):
    """the frame's backward, one C-ABI entry point per stage; returns the seven input gradients"""
    if True:
        (position, log_scaling, rotation, alpha_logit, feature, T, proj, points, feats, slot_of, indexes, cam_pos,
         tile_ranges, o2p, image, alpha, img_depth, tile_order, counts) = saved
        m = ctx.meta
        lib = nv.lib()
        dev = position.device
        n, V, K, w, h, F, C, col0 = m["n"], m["V"], m["K"], m["w"], m["h"], m["F"], m["C"], m["col0"]
        config = m["config"]
        cfg = nv.make_config(config)
        sh = nv.make_shard(m["shard"])
        s = nv.stream()
        RS = lib.gs_grad_row_floats(F)
        rows = torch.zeros((max(V, 1), RS), dtype=torch.float32, device=dev)
        if m["render_depth"] and V > 0 and h > 0 and any(g is not None for g in (g_image, g_img_depth, g_img_var)):
            # assemble the gradient of the rasterized (H,W,2+C) image from the three upstream gradients
            gf_ = g_image.contiguous() if g_image is not None else None
            gd_ = g_img_depth.contiguous() if g_img_depth is not None else None
            gv_ = g_img_var.contiguous() if g_img_var is not None else None
            nv.require_device(gf_, gd_, gv_, what="render_gaussians backward")
            g_image = torch.empty((h, w, F), dtype=torch.float32, device=dev)
            nv.check(lib.gs_depth_split_bwd(h * w, C, nv.ptr(img_depth), nv.ptr(alpha), 1e-6, nv.ptr(gf_), nv.ptr(gd_),
                                            nv.ptr(gv_), nv.ptr(g_image), s), "gs_depth_split_bwd")
        if g_image is not None and V > 0 and h > 0 and K > 0:
            gi = g_image.contiguous()
            nv.require_device(gi, what="render_gaussians backward")
            nv.check(lib.gs_raster_bwd(V, F, nv.ptr(points), nv.ptr(feats), nv.ptr(tile_ranges), nv.ptr(o2p), K, w,
                                       m["full_h"], cfg, nv.ptr(tile_order), _off(counts, 7), nv.ptr(image),
                                       nv.ptr(gi), nv.ptr(rows), sh, s),
                     "gs_raster_bwd")
        if config.compute_point_heuristic and V > 0:
            ctx.heur.copy_(rows[:V, 7 + F:9 + F])

        def add_attached(pts_rows):
            # gradients a caller attached to the projected splats / depths themselves (e.g. a regulariser): every
            # rank of a sharded frame holds the same, complete one, so it is added AFTER the partial sums are reduced
            if g_points is not None and V > 0:
                pts_rows[:V, :7] += g_points
            if g_depth is not None and V > 0 and m["render_depth"]:
                pts_rows[:V, 7] += g_depth.reshape(-1)

        def publish(pts_rows):
            # `Rendering.gaussians2d` is an OUTPUT of this node, so autograd alone would leave its .grad without the
            # rasterizer's dL/d(splat) (which never leaves the node).  The reference feeds that very tensor to
            # rasterize, so `gaussians2d.retain_grad()` + `viewspace_gradient` (renderer.py:234-239) is the classic
            # densification signal there: add the rasterizer's part (summed over the ranks of a sharded frame) to what
            # the retain_grad hook has stored (this pass's upstream gradient, earlier backward passes) -- .grad
            # accumulates over several backward passes as it does in the reference, and an empty view gets (0, 7).
            out = ctx.holder.get("gaussians2d") if ctx.holder else None
            out = out() if out is not None else None
            if out is not None and out.retains_grad:
                part = pts_rows[:V, :7].clone() if V > 0 else pts_rows.new_zeros((0, 7))
                out.grad = part if out.grad is None else out.grad + part

        extra_depth = None
        if g_depth is not None and V > 0 and not m["render_depth"]:
            extra_depth = g_depth.contiguous()

        g_feat, g_feat_stride = _off(rows, 7 + col0), RS     # dL/d(SH colour) columns
        g_pts, g_pts_stride = rows, RS                        # dL/d(points) [+ depth feature] columns
        wait_points = None
        sparse = m["shard"] is not None and m.get("exchange") == "sparse" and m.get("sizes") is not None
        if sparse:
            # Sparse exchange (parallel.py): one entry [row, 7 + F gradient words] per splat that can reach this rank's
            # rows -- the mapper's own list -- instead of the dense rows, 7/8 of which are zeros on every rank of 8.
            from . import parallel
            rows_n = rows.shape[0]
            touched, M = m["touched"], int(m["touched"].shape[0])
            width = parallel.ENTRY_HEAD + F
            entries = torch.empty((max(M, 1), width), dtype=torch.float32, device=dev)
            nv.check(lib.gs_shard_pack_sparse(M, nv.ptr(touched), F, col0, nv.ptr(rows),
                                              nv.ptr(feats) if m["degree"] >= 0 else None, nv.ptr(entries), s),
                     "gs_shard_pack_sparse")
            table = m["sizes"].result()
            group = m["group"]
            rank = torch.distributed.get_rank(group) if torch.distributed.is_initialized() else m["emulated_rank"]
            if m["grad_mode"] == "sharded":   # table[q][r] = entries rank q holds for owner r
                lists = parallel.exchange_entries_sharded(entries, [int(x) for x in table[rank]],
                                                          [int(table[q][rank]) for q in range(len(table))], group)
            else:                             # table[q][0] = list length of rank q
                lists = parallel.exchange_entries_replicated(entries, M, [int(t[0]) for t in table], group)
            m["exchanged"] = dict(width=width, own=M, table=table)
            # every list in one pass over the dense rows (rank order inside each tile: the same sums on every rank;
            # every row is written, so no clearing)
            pf = torch.empty((rows_n, C), dtype=torch.float32, device=dev)
            pp = torch.empty((rows_n, 7 + col0), dtype=torch.float32, device=dev)
            nl = len(lists)
            ptrs = (ctypes.c_void_p * nl)(*[ent.data_ptr() if cnt else None for ent, cnt in lists])
            cnts = (ctypes.c_int64 * nl)(*[cnt for _, cnt in lists])
            tmp_bytes = 4 * nl * (-(-rows_n // 256) + 1)
            tmp = torch.empty((tmp_bytes,), dtype=torch.uint8, device=dev)
            nv.check(lib.gs_shard_merge_sparse(nl, ptrs, cnts, F, col0, rows_n, nv.ptr(pf), nv.ptr(pp),
                                               nv.ptr(m.get("owned_rows")), nv.ptr(tmp), tmp_bytes, s),
                     "gs_shard_merge_sparse")
            g_feat, g_feat_stride = nv.ptr(pf), C
            g_pts, g_pts_stride = pp, 7 + col0
        elif m["shard"] is not None:
            # Every rank rendered different rows: the per-Gaussian partial gradients are summed over the
            # ranks, 4*(7+F) bytes per visible Gaussian in all.  Two collectives, colour columns first: the SH
            # adjoint only needs those and runs while the splat columns are still in flight.
            from .parallel import _reduce_partial_gradients
            # (packed by one kernel, which also applies the SH clamp mask: only the ranks that rasterized a splat know it)
            rows_n = rows.shape[0]
            pf = torch.empty((rows_n, C), dtype=torch.float32, device=dev)
            pp = torch.empty((rows_n, 7 + col0), dtype=torch.float32, device=dev)
            nv.check(lib.gs_shard_pack_grads(rows_n, F, col0, nv.ptr(rows), nv.ptr(feats) if m["degree"] >= 0 else None,
                                             nv.ptr(pf), nv.ptr(pp), s), "gs_shard_pack_grads")
            wait_points = _reduce_partial_gradients(pf, pp, m["group"])
            g_feat, g_feat_stride = nv.ptr(pf), C
            g_pts, g_pts_stride = pp, 7 + col0
        else:
            publish(rows)
            add_attached(rows)

        # grad_mode "sharded": the adjoints run on this rank's index range [lo, hi) only -- the same kernels on base
        # pointers moved to row `lo` (every array they touch is indexed by the Gaussian, the gradient rows through
        # slot_of) -- and the gradients come out range-shaped
        lo, hi = m["owned_range"] if m.get("owned_range") is not None else (0, n)
        nr = hi - lo
        need_T, need_proj = ctx.camera_grads

        def at(t, row):  # pointer to row `row` of a per-Gaussian tensor
            return ctypes.c_void_p(t.data_ptr() + row * t.stride(0) * t.element_size())

        d_feature = torch.empty((nr, *feature.shape[1:]), dtype=torch.float32, device=dev)
        # camera matrix under optimisation: the SH view direction depends on the camera centre = inverse(T)[:3, 3]
        # (reference perspective/params.py:76-78), so the SH adjoint also returns dL/d(centre) and the 4x4 inverse is
        # differentiated below (pose refinement is rare: a handful of tiny torch ops, off the common path)
        d_centre = None
        if m["degree"] >= 1 and need_T:
            d_centre = torch.zeros((3,), dtype=torch.float32, device=dev)
        if m["degree"] >= 0:
            nv.check(lib.gs_sh_bwd(nr, V, C, m["degree"], at(feature, lo), at(position, lo), nv.ptr(indexes), 1,
                                   at(slot_of, lo), nv.ptr(cam_pos), g_feat, g_feat_stride, _off(feats, col0), F,
                                   nv.ptr(d_feature), None, nv.ptr(d_centre), s), "gs_sh_bwd")
        else:
            nv.check(lib.gs_feature_gather_bwd(nr, C, at(slot_of, lo), g_feat, g_feat_stride, nv.ptr(d_feature), s),
                     "gs_feature_gather_bwd")
        if wait_points is not None:
            wait_points.wait()
        if m["shard"] is not None:
            publish(g_pts)
            add_attached(g_pts)

        f32r = dict(dtype=torch.float32, device=dev)
        d_pos, d_ls = torch.empty((nr, 3), **f32r), torch.empty((nr, 3), **f32r)
        d_rot, d_al = torch.empty((nr, 4), **f32r), torch.empty((nr, 1), **f32r)
        d_T = torch.empty((4, 4), dtype=torch.float32, device=dev) if need_T else None
        d_proj = torch.empty((4,), dtype=torch.float32, device=dev) if need_proj else None
        nbytes = lib.gs_project_bwd_scratch_bytes(nr) if (need_T or need_proj) else 0
        scratch = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=dev)
        if m["render_depth"]:
            gd, gd2, gstride = _off(g_pts, 7), _off(g_pts, 8), g_pts_stride
        else:
            gd, gd2, gstride = nv.ptr(extra_depth), None, 1
        nv.check(lib.gs_project_bwd(nr, V, at(position, lo), at(log_scaling, lo), at(rotation, lo), at(alpha_logit, lo),
                                    nv.ptr(T), nv.ptr(proj), w, m["full_h"], cfg, at(slot_of, lo), nv.ptr(g_pts),
                                    g_pts_stride, gd, gd2, gstride, nv.ptr(d_pos), nv.ptr(d_ls), nv.ptr(d_rot),
                                    nv.ptr(d_al), nv.ptr(d_T), nv.ptr(d_proj), nv.ptr(scratch), nbytes, s),
                 "gs_project_bwd")
        if d_centre is not None:  # Y = T^-1, dL/dT = -Y^T (dL/dY) Y^T with dL/dY zero except the centre column
            with torch.no_grad():
                Y = torch.linalg.inv(T.detach().cpu().double())
                dY = torch.zeros((4, 4), dtype=torch.float64)
                dY[:3, 3] = d_centre.cpu().double()
                d_T = d_T + (-(Y.T @ dY @ Y.T)).to(device=dev, dtype=torch.float32)
        return (d_pos, d_ls, d_rot, d_al, d_feature, d_T, d_proj)


# ---------------------------------------------------------------------------------------------------------------
# One C-ABI call per direction (include/gsplat_hip.h gs_frame_fwd / gs_frame_bwd): the same stages as _FusedRender
# above, enqueued from a single host call into one workspace whose sub-buffers are carved by offset.  Used for every
# frame whose overlap count has been seen before (the first frame of a shape has to read K back before the pair
# buffers can be sized: that one runs _FusedRender, as does the re-run after a capacity overflow).
_FRAMES = {}   # frame key -> (GsFrame, GsFrameLayout)
_EVENTS = {}   # device index -> ring of (torch.cuda.Event, raw handle)
_EMPTY = {}    # (device, shape) -> cached empty placeholder outputs


class _Overflow(Exception):
    pass


def _frame_for(n, C, degree, w, full_h, depth_range, render_depth, use_depth16, render_median, prepare_backward,
               k_cap, tile_hint, shard, config, exchange_world=0, exchange_rank=0):
    key = (n, C, degree, w, full_h, float(depth_range[0]), float(depth_range[1]), render_depth, use_depth16,
           render_median, prepare_backward, k_cap, tile_hint, shard, config, nv.TUNING["wave_sub_blocks"],
           nv.TUNING["no_heavy_split"], exchange_world, exchange_rank)
    hit = _FRAMES.get(key)
    if hit is None:
        frame = nv.GsFrame()
        frame.n, frame.channels, frame.sh_degree, frame.width, frame.height = n, C, degree, w, full_h
        frame.near_plane, frame.far_plane = float(depth_range[0]), float(depth_range[1])
        frame.render_depth, frame.use_depth16 = int(render_depth), int(use_depth16)
        frame.render_median_depth, frame.prepare_backward = int(render_median), int(prepare_backward)
        frame.k_capacity, frame.max_tile_hint = int(k_cap), int(tile_hint)
        frame.has_shard = 0 if shard is None else 1
        if shard is not None:
            frame.shard = nv.GsRowShard(int(shard.row_begin), int(shard.row_end), int(shard.band), int(shard.period),
                                        int(shard.phase))
        frame.cfg = nv.make_config(config)
        frame.depth_forward_cut = nv.make_config(config, cut_scale=float(depth_range[1]) ** 2).forward_cut
        frame.exchange_world, frame.exchange_rank = int(exchange_world), int(exchange_rank)
        layout = nv.GsFrameLayout()
        nv.check(nv.lib().gs_frame_layout(ctypes.byref(frame), ctypes.byref(layout)), "gs_frame_layout")
        if len(_FRAMES) > 256:
            _FRAMES.clear()
        hit = _FRAMES[key] = (frame, layout)
    return hit


def _counts_event(dev: torch.device):
    ring = _EVENTS.get(dev.index)
    if ring is None:
        evs = []
        for _ in range(4):
            ev = torch.cuda.Event()
            ev.record()  # creates the underlying hipEvent_t, whose handle the library records later
            evs.append((ev, ctypes.c_void_p(ev.cuda_event)))
        ring = _EVENTS[dev.index] = dict(evs=evs, at=0)
    ring["at"] = (ring["at"] + 1) % len(ring["evs"])
    return ring["evs"][ring["at"]]


_FORKS = {}    # device index -> (side stream, events, GsFrameFork)
# GS_FORK_COLOURS=1: gs_frame_fwd runs the colour stage on a side stream underneath the tile mapper (GsFrameFork).
# Measured at C3: 1.103 vs 1.108 ms per frame -- the mapper's first kernels slow down by what the overlap hides (the
# kernel trace shows region_count at 41 us instead of 13 beside the 52-us SH kernel) -- so it is OFF by default.
FORK_COLOURS = __import__("os").environ.get("GS_FORK_COLOURS", "0") == "1"


def _frame_fork(dev: torch.device):
    hit = _FORKS.get(dev.index)
    if hit is None:
        side = torch.cuda.Stream(device=dev)
        evs = []
        for _ in range(2):
            ev = torch.cuda.Event()
            ev.record()  # materialises the hipEvent_t
            evs.append(ev)
        fork = nv.GsFrameFork(side.cuda_stream, evs[0].cuda_event, evs[1].cuda_event)
        hit = _FORKS[dev.index] = (side, evs, fork)
    return ctypes.byref(hit[2])


def _empty(dev, shape):
    key = (dev, shape)
    t = _EMPTY.get(key)
    if t is None:
        t = _EMPTY[key] = torch.empty(shape, dtype=torch.float32, device=dev)
    return t


class _FrameRender(torch.autograd.Function):
    @staticmethod
    @nv.on_tensor_device
    def forward(ctx, position, log_scaling, rotation, alpha_logit, feature, T_camera_world, projection,
                image_size, depth_range, config: RasterConfig, render_depth: bool, use_depth16: bool,
                render_median: bool, shard, group, holder, exchange: str, grad_mode: str, owned_range, key,
                k_cap: int, tile_hint: int):
        nv.require_device(position, log_scaling, rotation, alpha_logit, feature, T_camera_world, projection,
                          what="render_gaussians")
        lib = nv.lib()
        dev = position.device
        n = position.shape[0]
        w, full_h = int(image_size[0]), int(image_size[1])
        C = feature.shape[1]
        degree = check_sh_degree(feature) if feature.dim() == 3 else -1
        needs_grad = any(ctx.needs_input_grad[:7])
        # a sharded frame's backward runs stage by stage (the exchange sits in its middle) and clears its own rows
        prepare_backward = needs_grad and shard is None
        ex_world, ex_rank = _exchange_ranks(shard, exchange, group, n, owned_range)
        frame, L = _frame_for(n, C, degree, w, full_h, depth_range, render_depth, use_depth16, render_median,
                              prepare_backward, k_cap, tile_hint, shard, config, ex_world, ex_rank)
        T = T_camera_world.contiguous()
        proj = projection.contiguous()
        ws = torch.empty((L.workspace_bytes,), dtype=torch.uint8, device=dev)
        scratch = torch.empty((L.fwd_scratch_bytes,), dtype=torch.uint8, device=dev)
        host_counts = _pinned_counts(dev)
        ready, ready_handle = _counts_event(dev)
        nv.check(lib.gs_frame_fwd(ctypes.byref(frame), nv.ptr(position), nv.ptr(log_scaling), nv.ptr(rotation),
                                  nv.ptr(alpha_logit), nv.ptr(feature), nv.ptr(T), nv.ptr(proj), nv.ptr(ws),
                                  L.workspace_bytes, nv.ptr(scratch), L.fwd_scratch_bytes, nv.ptr(host_counts),
                                  ready_handle, _frame_fork(dev) if FORK_COLOURS else None,
                                  nv.stage_events(nv.FRAME_FWD_STAGES), nv.stream()), "gs_frame_fwd")
        ready.synchronize()  # waits for the mapper's scan only, not for the rasterizer
        host = host_counts.tolist()
        K, max_tile, overflow, V = host[0], host[1], host[2], host[4]
        hint = _K_HINT.get(key)
        _K_HINT[key] = (max(K, hint[0]) if hint else K, max(max_tile, hint[1]) if hint else max_tile)
        if overflow:  # more overlaps than the hint allowed for: the caller runs the frame again with exact sizes
            raise _Overflow()

        F, h = L.num_features, L.local_height
        f32 = ws.view(torch.float32)

        def view(off, shape, base=f32, esize=4):
            strides, acc = [], 1
            for s_ in reversed(shape):
                strides.append(acc)
                acc *= s_
            return base.as_strided(shape, tuple(reversed(strides)), off // esize).detach()

        image = view(L.image, (h, w, F))
        alpha = view(L.alpha, (h, w))
        points_v = view(L.points, (V, 7))
        depth_v = view(L.depth, (V, 1))
        indexes_v = view(L.indexes, (V,), ws.view(torch.int64), 8)
        empty = _empty(dev, (0,))
        vis_out = view(L.visibility, (V,)) if config.compute_visibility else empty
        heur = torch.zeros((V, 2), dtype=torch.float32, device=dev) if config.compute_point_heuristic \
            else _empty(dev, (0, 2))
        out_image = image
        img_depth = img_var = empty
        if render_depth:
            out_image = view(L.out_image, (h, w, C))
            img_depth, img_var = view(L.img_depth, (h, w)), view(L.img_var, (h, w))
        median = empty
        if render_median:
            median = view(L.median, (h, w))

        ctx.meta = dict(n=n, V=V, K=K, w=w, h=h, full_h=full_h, F=F, C=C, col0=F - C, degree=degree, config=config,
                        render_depth=render_depth, group=group, shard=shard, far=float(depth_range[1]),
                        exchange=exchange, grad_mode=grad_mode, owned_range=owned_range)
        if shard is not None and exchange == "sparse":
            T_tiles = L.tiles_x * L.tiles_y
            # the frame call has already compacted the list (ascending rows) and cut it by owner: views of the workspace
            i32 = ws.view(torch.int32)
            M = int(host[5]) if T_tiles > 0 else 0
            prepared = dict(touched=i32.as_strided((M,), (1,), L.touched // 4),
                            owner_counts=ws.view(torch.int64).as_strided((ex_world,), (1,), L.owner_counts // 8),
                            owned_rows=i32.as_strided((2,), (1,), L.counts // 4 + 2))
            _start_sparse_exchange(ctx.meta, n, T_tiles, M, None, None, 0, prepared=prepared)
        if holder is not None and shard is not None:
            holder["touched_count"] = int(host[5]) if L.tiles_x * L.tiles_y > 0 else 0
        ctx.camera_grads = (ctx.needs_input_grad[5], ctx.needs_input_grad[6])
        ctx.frame, ctx.layout = frame, L
        ctx.heur = heur
        ctx.holder = holder
        ctx.rows_clean = prepare_backward
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(position, log_scaling, rotation, alpha_logit, feature, T, proj, ws)
        ctx.mark_non_differentiable(alpha, indexes_v, vis_out, heur, median)
        if not render_depth:
            ctx.mark_non_differentiable(img_depth, img_var)
        return out_image, alpha, points_v, depth_v, indexes_v, vis_out, heur, img_depth, img_var, median

    @staticmethod
    @nv.on_tensor_device
    def backward(ctx, g_image, _g_alpha, g_points, g_depth, _g_idx, _g_vis, _g_heur, g_img_depth, g_img_var,
                 _g_median=None):
        position, log_scaling, rotation, alpha_logit, feature, T, proj, ws = ctx.saved_tensors
        m, frame, L = ctx.meta, ctx.frame, ctx.layout
        if m["shard"] is not None:
            # a sharded frame exchanges its partial gradients between the rasterizer's backward and the per-Gaussian
            # adjoints: stage by stage, on views of the workspace
            f32_, i32_ = ws.view(torch.float32), ws.view(torch.int32)

            def vw(base, off, shape, esize=4):
                strides, acc = [], 1
                for s_ in reversed(shape):
                    strides.append(acc)
                    acc *= s_
                return base.as_strided(shape, tuple(reversed(strides)), off // esize)

            n_, F_, h_, w_ = m["n"], m["F"], m["h"], m["w"]
            T_tiles = max(L.tiles_x * L.tiles_y, 0)
            saved = (position, log_scaling, rotation, alpha_logit, feature, T, proj,
                     vw(f32_, L.points, (n_, 7)), vw(f32_, L.features, (n_, F_)), vw(i32_, L.slot_of, (n_,)),
                     vw(ws.view(torch.int64), L.indexes, (n_,), 8), vw(f32_, L.camera_pos, (3,)),
                     vw(i32_, L.tile_ranges, (T_tiles, 2)), vw(i32_, L.overlap_to_point, (frame.k_capacity,)),
                     vw(f32_, L.image, (h_, w_, F_)), vw(f32_, L.alpha, (h_, w_)),
                     vw(f32_, L.img_depth, (h_, w_)) if m["render_depth"] else _empty(position.device, (0,)),
                     vw(i32_, L.tile_order, (T_tiles,)), vw(i32_, L.counts, (8,)))
            return _backward_stages(ctx, saved, g_image, g_points, g_depth, g_img_depth, g_img_var) + (None,) * 15
        lib = nv.lib()
        dev = position.device
        n, V, K, F = m["n"], m["V"], m["K"], m["F"]
        config = m["config"]
        RS = L.grad_row_floats
        f32 = ws.view(torch.float32)
        scratch = torch.empty((L.bwd_scratch_bytes,), dtype=torch.uint8, device=dev)
        rows_off = L.grad_rows if L.grad_rows >= 0 else None
        rows = (f32.as_strided((max(V, 1), RS), (RS, 1), L.grad_rows // 4) if rows_off is not None
                else scratch.view(torch.float32).as_strided((max(V, 1), RS), (RS, 1), L.b_grad_rows // 4))
        if rows_off is not None and not ctx.rows_clean:
            rows.zero_()  # a second backward through the same frame (retain_graph): the rows hold the first one's sums
        ctx.rows_clean = False
        gi = gd_ = gv_ = None
        if g_image is not None:
            gi = g_image.contiguous()
        if m["render_depth"]:
            gd_ = g_img_depth.contiguous() if g_img_depth is not None else None
            gv_ = g_img_var.contiguous() if g_img_var is not None else None
        att_p = g_points.contiguous() if (g_points is not None and V > 0) else None
        att_d = g_depth.contiguous() if (g_depth is not None and V > 0) else None
        nv.require_device(gi, gd_, gv_, att_p, att_d, what="render_gaussians backward")
        need_T, need_proj = ctx.needs_input_grad[5], ctx.needs_input_grad[6]
        # one allocation for the five parameter gradients
        sizes = (position.numel(), log_scaling.numel(), rotation.numel(), alpha_logit.numel(), feature.numel())
        flat = torch.empty((sum(sizes),), dtype=torch.float32, device=dev)
        outs, at = [], 0
        for t, sz in zip((position, log_scaling, rotation, alpha_logit, feature), sizes):
            outs.append(flat.as_strided(t.shape, t.stride(), at))
            at += sz
        d_pos, d_ls, d_rot, d_al, d_feature = outs
        d_T = torch.empty((4, 4), dtype=torch.float32, device=dev) if need_T else None
        d_proj = torch.empty((4,), dtype=torch.float32, device=dev) if need_proj else None
        d_centre = None
        if m["degree"] >= 1 and need_T:
            d_centre = torch.zeros((3,), dtype=torch.float32, device=dev)
        nv.check(lib.gs_frame_bwd(ctypes.byref(frame), nv.ptr(position), nv.ptr(log_scaling), nv.ptr(rotation),
                                  nv.ptr(alpha_logit), nv.ptr(feature), nv.ptr(T), nv.ptr(proj), nv.ptr(ws),
                                  L.workspace_bytes, nv.ptr(scratch), L.bwd_scratch_bytes, V, K, nv.ptr(gi),
                                  nv.ptr(gd_), nv.ptr(gv_), nv.ptr(att_p), nv.ptr(att_d), nv.ptr(d_pos), nv.ptr(d_ls),
                                  nv.ptr(d_rot), nv.ptr(d_al), nv.ptr(d_feature), nv.ptr(d_T), nv.ptr(d_proj),
                                  nv.ptr(d_centre), nv.stage_events(nv.FRAME_BWD_STAGES), nv.stream()), "gs_frame_bwd")
        if config.compute_point_heuristic and V > 0:
            ctx.heur.copy_(rows[:V, 7 + F:9 + F])
        # publish the rasterizer's part of dL/d(gaussians2d) (see _FusedRender.backward)
        out = ctx.holder.get("gaussians2d") if ctx.holder else None
        out = out() if out is not None else None
        if out is not None and out.retains_grad:
            part = rows[:V, :7].clone() if V > 0 else rows.new_zeros((0, 7))
            if att_p is not None:
                part -= att_p  # the rows hold the attached gradient as well; the retain_grad hook has stored that part
            out.grad = part if out.grad is None else out.grad + part
        if d_centre is not None:  # Y = T^-1, dL/dT = -Y^T (dL/dY) Y^T with dL/dY zero except the centre column
            with torch.no_grad():
                Y = torch.linalg.inv(T.detach().cpu().double())
                dY = torch.zeros((4, 4), dtype=torch.float64)
                dY[:3, 3] = d_centre.cpu().double()
                d_T = d_T + (-(Y.T @ dY @ Y.T)).to(device=dev, dtype=torch.float32)
        return (d_pos, d_ls, d_rot, d_al, d_feature, d_T, d_proj) + (None,) * 15


class _OwnedRender(torch.autograd.Function):
    """grad_mode "sharded" (parallel.py): the replicated Gaussians are DATA here; what is differentiated are this rank's
    own rows [lo, hi) of them, held as separate leaf tensors by a sharded optimizer (`owned`: their values are taken to
    be the corresponding rows of the replicated tensors).  Forward: the sharded frame as usual.  Backward: partial
    gradients go to their owners through one all-to-all, the SH / projection adjoints run on the owned range only and
    the gradients come out range-shaped -- no all-gather of gradients, no (N, ...) zero rows written."""

    @staticmethod
    @nv.on_tensor_device
    def forward(ctx, o_position, o_log_scaling, o_rotation, o_alpha_logit, o_feature, full, rest, key, k_cap, tile_hint):
        ctx.needs_full = (True,) * 5 + (False, False)
        if k_cap:
            return _FrameRender.forward(_Proxy(ctx), *full, *rest, key, k_cap, tile_hint)
        return _FusedRender.forward(_Proxy(ctx), *full, *rest)

    @staticmethod
    @nv.on_tensor_device
    def backward(ctx, g_image, _g_alpha, g_points, g_depth, _g_idx, _g_vis, _g_heur, g_img_depth, g_img_var,
                 _g_median=None):
        inner = _FrameRender if hasattr(ctx, "frame") else _FusedRender
        grads = inner.backward(_Proxy(ctx), g_image, _g_alpha, g_points, g_depth, _g_idx, _g_vis, _g_heur, g_img_depth,
                               g_img_var, _g_median)
        return tuple(grads[:5]) + (None,) * 5


class _Proxy:
    """lets the forward / backward bodies of the two Functions above run on behalf of _OwnedRender: same attributes,
    but the differentiable inputs are the five Gaussian tensors and never the camera"""

    def __init__(self, ctx):
        object.__setattr__(self, "_ctx", ctx)

    def __getattr__(self, name):
        if name == "needs_input_grad":
            return object.__getattribute__(self, "_ctx").needs_full
        return getattr(object.__getattribute__(self, "_ctx"), name)

    def __setattr__(self, name, value):
        setattr(object.__getattribute__(self, "_ctx"), name, value)


def fused_supported(gaussians, camera_params, use_sh: bool, render_median_depth: bool) -> bool:
    """The fused node covers SH colours (N, C <= 8, D) and plain features (N, C <= 30), with or without the depth
    and median-depth images, camera gradients included.  What is left -- an empty scene, wider features -- runs the
    composed operators."""
    f = gaussians.feature
    if gaussians.position.shape[0] == 0 or not f.is_cuda or f.dtype != torch.float32:
        return False
    if use_sh:
        return f.ndim == 3 and f.shape[1] <= 8
    return f.ndim == 2 and 1 <= f.shape[1] <= 30


def render_fused(gaussians, camera_params, config: RasterConfig, render_depth: bool, use_depth16: bool,
                 shard=None, group=None, render_median_depth: bool = False, exchange: str = "dense",
                 grad_mode: str = "replicated", owned=None, owned_range=None):
    """shard (parallel.RowShard): render only the tile rows this rank owns; the images then hold those pixel rows,
    everything per-Gaussian (`gaussians2d` included) stays in full-image coordinates.
    See parallel.render_gaussians_sharded."""
    import weakref

    from .renderer import Rendering
    holder = {}
    args = (gaussians.position.contiguous(), gaussians.log_scaling.contiguous(), gaussians.rotation.contiguous(),
            gaussians.alpha_logit.contiguous(), gaussians.feature.contiguous(), camera_params.T_camera_world,
            camera_params.projection, camera_params.image_size, camera_params.depth_range, config, render_depth,
            use_depth16, render_median_depth, shard, group, holder, exchange, grad_mode, owned_range)
    size = camera_params.image_size
    key = (args[0].shape[0], int(size[0]), int(size[1]), shard, config.tile_size, bool(use_depth16))
    hint = _K_HINT.get(key)
    outs = None
    if hint is None and FRAME_CALLS == "always":  # tests: size the frame by an untracked staged pass first
        with torch.no_grad():
            _FusedRender.apply(*args[:16], "dense", "replicated", None)
        hint = _K_HINT.get(key)
    if owned is not None:
        # grad_mode "sharded": gradients flow to the rank's own rows (range-shaped leaf tensors), see _OwnedRender
        k_cap = tile_hint = 0
        if hint is not None and FRAME_CALLS:
            k_cap = -(-(int(hint[0] * 1.25) + 4096) // 65536) * 65536
            tile_hint = next((c for c in (256, 512, 1024, 2048) if hint[1] <= c), 4096)
        own = (owned.position.contiguous(), owned.log_scaling.contiguous(), owned.rotation.contiguous(),
               owned.alpha_logit.contiguous(), owned.feature.contiguous())
        full = tuple(t.detach() for t in args[:7])
        try:
            outs = _OwnedRender.apply(*own, full, args[7:], key, k_cap, tile_hint)
        except _Overflow:
            outs = _OwnedRender.apply(*own, full, args[7:], key, 0, 0)
    elif hint is not None and FRAME_CALLS:
        # one C-ABI call per direction; capacities rounded up to a few classes so that the cached frame descriptors
        # and workspace layouts are reused from frame to frame
        k_cap = -(-(int(hint[0] * 1.25) + 4096) // 65536) * 65536
        tile_hint = next((c for c in (256, 512, 1024, 2048) if hint[1] <= c), 4096)
        try:
            outs = _FrameRender.apply(*args, key, k_cap, tile_hint)
        except _Overflow:
            outs = None  # more overlaps than the hint allowed for (now updated): run the stages with exact sizes
    if outs is None:
        outs = _FusedRender.apply(*args)
    image, alpha, g2d, depths, indexes, vis, heur, img_depth, img_var, median = outs
    holder["gaussians2d"] = weakref.ref(g2d)
    indexes._gs_unique = True
    if not render_depth:
        img_depth = img_var = None
    rendering = Rendering(image=image, image_weight=alpha, depth=img_depth, depth_var=img_var,
                          median_depth=median if render_median_depth else None, camera=camera_params, config=config,
                          point_visibility=vis if config.compute_visibility else None,
                          point_heuristic=heur if config.compute_point_heuristic else None,
                          points_in_view=indexes, point_depth=depths, gaussians2d=g2d)
    if shard is not None:  # how many splats can reach this rank's rows (= its list in a sparse exchange)
        object.__setattr__(rendering, "touched_count", holder.get("touched_count"))
    return rendering
