"""Multi-GPU rendering: screen tiles shard across ranks, one RCCL exchange of per-Gaussian
gradients (the reference has no distributed code at all).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Every rank holds the
full, replicated Gaussian set and for each frame:

  1. projects all N Gaussians and evaluates SH colour (replicated: 236 B/Gaussian read, cheaper than
     exchanging projected splats, and it keeps the visible set identical everywhere);
  2. owns a set of tile ROWS of the image (`RowShard`): one contiguous strip, or -- `interleave=b` -- bands of b
     tile rows dealt round-robin over the ranks (load balance on real scenes, where the splats crowd a part of
     the screen).  The mapper and the rasterizer work in full-image coordinates and simply skip the rows they do
     not own (GsRowShard in include/gsplat_hip.h), so a shard's tiles get exactly the lists and pixels of the
     unsharded frame and `Rendering.gaussians2d` is the same tensor on every rank;
  3. evaluates the loss on its rows; the rasterizer backward yields PARTIAL gradients for the
     projected splats (V,7), their features (V,C) and depths (V,1);
  4. all-reduces those partial gradients -- 4*(7+C+1)*V bytes (40 B/Gaussian, vs 236 B/Gaussian if the final
     parameter gradients were reduced instead), as two collectives, colour columns first, so that the SH adjoint
     runs while the splat columns are in flight.  EVERY rank issues the same two collectives with the same shapes,
     also one that owns no row;
  5. runs SH / projection backward redundantly, so every rank ends with identical, complete
     parameter gradients (drop-in for a replicated optimizer).

The stage operators are injectable (`ops`) so the sharding + collective logic is exercised on CPU
with the gloo backend in tests/ (stage callables backed by the CPU oracle there).  That composed path renders
band by band on exactly shifted coordinates (the operators keep the reference's signatures, which know nothing
of shards); the HIP path (`ops=None`) is the fused frame with kernel-level row ownership.
"""
from __future__ import annotations

from dataclasses import dataclass
from functools import lru_cache
from types import SimpleNamespace
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from .data_types import Gaussians3D, RasterConfig
from .perspective.params import CameraParams


def tile_rows(image_height: int, tile_size: int) -> int:
    return -(-int(image_height) // tile_size)


def strip_rows(rank: int, world: int, num_tile_rows: int) -> Tuple[int, int]:
    """Contiguous, balanced split of tile rows: rank r owns rows [start, end)."""
    base, rem = divmod(num_tile_rows, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def strip_pixels(rank: int, world: int, image_height: int, tile_size: int) -> Tuple[int, int]:
    r0, r1 = strip_rows(rank, world, tile_rows(image_height, tile_size))
    return min(r0 * tile_size, image_height), min(r1 * tile_size, image_height)


@dataclass(frozen=True)
class RowShard:
    """The tile rows of the full image one rank owns (mirrors GsRowShard, include/gsplat_hip.h).
    period == 1: the contiguous strip [row_begin, row_end); period > 1: row ty is owned iff
    (ty // band) % period == phase (row_begin = 0, row_end = all rows)."""
    row_begin: int
    row_end: int
    band: int
    period: int
    phase: int
    tile_size: int
    image_height: int

    def owns(self, ty: int) -> bool:
        if not (self.row_begin <= ty < self.row_end):
            return False
        return self.period <= 1 or (ty // self.band) % self.period == self.phase

    def rows(self) -> List[int]:
        return list(_rows_of(self))

    @property
    def bands(self) -> List[Tuple[int, int]]:
        """owned pixel-row bands [(y0, y1), ...] of the full image, ascending, maximal"""
        return list(_bands_of(self))

    @property
    def local_height(self) -> int:
        return sum(y1 - y0 for y0, y1 in self.bands)

    @property
    def local_rows(self) -> int:
        return len(self.rows())


@lru_cache(maxsize=256)
def _rows_of(shard: RowShard) -> Tuple[int, ...]:
    return tuple(ty for ty in range(shard.row_begin, shard.row_end) if shard.owns(ty))


@lru_cache(maxsize=256)
def _bands_of(shard: RowShard) -> Tuple[Tuple[int, int], ...]:
    out: List[Tuple[int, int]] = []
    for ty in _rows_of(shard):
        y0, y1 = ty * shard.tile_size, min((ty + 1) * shard.tile_size, shard.image_height)
        if out and out[-1][1] == y0:
            out[-1] = (out[-1][0], y1)
        else:
            out.append((y0, y1))
    return tuple(out)


@lru_cache(maxsize=256)
def shard_for(rank: int, world: int, image_height: int, tile_size: int, interleave: int = 0) -> RowShard:
    """interleave = 0: balanced contiguous strips; interleave = b > 0: bands of b tile rows, band j -> rank j % world."""
    rows = tile_rows(image_height, tile_size)
    if interleave and world > 1:
        return RowShard(0, rows, int(interleave), world, rank, tile_size, int(image_height))
    r0, r1 = strip_rows(rank, world, rows)
    return RowShard(r0, r1, max(rows, 1), 1, 0, tile_size, int(image_height))


def owned_pixel_rows(bands) -> torch.Tensor:
    """row indices (int64) of the full image covered by `bands`, in the order a shard's images hold them"""
    if not bands:
        return torch.empty((0,), dtype=torch.int64)
    return torch.cat([torch.arange(y0, y1, dtype=torch.int64) for y0, y1 in bands])


def _reduce_partial_gradients(colour: torch.Tensor, splat: torch.Tensor, group, wait_colour: bool = True):
    """THE exchange step of a sharded frame, shared by the fused and the composed path so that every rank issues the
    same collectives in the same order: sum over the ranks of the colour columns (V', C), then of the splat columns
    (V', 7 [+2 depth features | +1 depth]), V' = max(V, 1).  Returns the handle of the second collective (or None)."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return None
    first = dist.all_reduce(colour, op=dist.ReduceOp.SUM, group=group, async_op=True)
    second = dist.all_reduce(splat, op=dist.ReduceOp.SUM, group=group, async_op=True)
    if wait_colour:
        first.wait()
    return second


# ------------------------------------------------------------------------------------------------------------------
# Sparse exchange.  A rank's partial gradients cover only the splats that can reach its tile rows: of the rows an
# 8-rank dense all-reduce sums, 7/8 are zeros on every rank.  Each rank instead contributes a LIST of entries
# [row id (int32 bits), 7 + F gradient words] for the splats it touched:
#   grad_mode "replicated": the lists are all-gathered (padded to the longest, whose length every rank knows from an
#       all-gather of the list lengths issued during the forward) and every rank adds them into zeroed dense rows, one
#       list after the other IN RANK ORDER -- plain read-modify-write, the rows of one list are distinct -- so that all
#       ranks hold the same sums bit for bit (a replicated optimizer must not drift);
#   grad_mode "sharded": rank r is the owner of the Gaussians [lo_r, hi_r) (`owned_range`); the entries, grouped by
#       owner, go through ONE all-to-all with per-destination counts (known from the forward as well), the owner adds
#       what it received in source-rank order and runs the SH / projection adjoints on ITS index range only (SURVEY 8e's
#       alternative for a sharded optimizer: `reduce_scatter` + sharded a14; no all-gather of gradients).
# Exchanged bytes per rank (F = 3, 44-byte entries, M ~ V / R (1 + border) touched splats): replicated sends 44 M and
# receives 44 (R - 1) M, sharded sends and receives 44 M (R - 1) / R, against 2 (R - 1) / R * 40 V each way for the ring
# all-reduce of the dense rows.
# tools/exp_shard.py only: > 1 lets ONE process run a rank's share of a sparse exchange without a process group -- the
# pack and add kernels see lists of the size the real exchange would deliver (every peer's list taken to be as long as
# this rank's own), nothing is communicated
EMULATED_WORLD = 0
EXCHANGES = ("dense", "sparse")
GRAD_MODES = ("replicated", "sharded")
ENTRY_HEAD = 8  # words of an entry in front of the F feature gradients: row id + 7 splat gradients


def owned_range(rank: int, world: int, n: int) -> Tuple[int, int]:
    """the Gaussians [lo, hi) whose gradients rank `rank` reduces and owns in grad_mode "sharded" """
    chunk = -(-int(n) // max(int(world), 1))
    lo = min(rank * chunk, n)
    return lo, min(lo + chunk, n)


def split_owned(gaussians: Gaussians3D, rank: int, world: int) -> Gaussians3D:
    """this rank's rows of the replicated Gaussians as independent leaf tensors (what a sharded optimizer holds)"""
    lo, hi = owned_range(rank, world, gaussians.position.shape[0])
    return gaussians.apply(lambda t: t[lo:hi].detach().clone())


def gather_owned(owned: Gaussians3D, n: int, group=None) -> Gaussians3D:
    """all-gather the ranks' owned rows into the replicated (detached) Gaussians every rank projects -- the step a
    sharded optimizer needs after its update (236 B / Gaussian at SH degree 3; the price of not all-gathering gradients)"""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return owned.apply(lambda t: t.detach())
    chunk = -(-int(n) // world)

    def gather(t):
        pad = t.new_zeros((chunk, *t.shape[1:]))
        pad[:t.shape[0]] = t.detach()
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        return torch.cat(parts, 0)[:n].contiguous()
    return owned.apply(gather)


def exchange_sizes(counts: torch.Tensor, group=None, async_op: bool = False):
    """all-gather of every rank's count vector (k int64: its list length, or its per-destination counts).  Returns
    (matrix (world, k), handle or None)"""
    world = dist.get_world_size(group)
    parts = [torch.empty_like(counts) for _ in range(world)]
    handle = dist.all_gather(parts, counts, group=group, async_op=async_op)
    return parts, handle


_PINNED_SIZES = {}  # (world, k) -> ring of pinned int64 buffers


class SizesFuture:
    """The list lengths of a sparse exchange are known when the mapper has run, i.e. during the FORWARD: they are
    all-gathered there (asynchronously) and parked in pinned host memory, so that the backward finds them without a
    device synchronisation of its own."""

    def __init__(self, counts: torch.Tensor, group=None):
        if EMULATED_WORLD > 1 and not dist.is_initialized():
            self.host, self.event = torch.stack([counts] * EMULATED_WORLD).cpu(), None
            return
        self.world = dist.get_world_size(group)
        parts, handle = exchange_sizes(counts, group, async_op=True)
        handle.wait()  # NCCL: the current stream waits; gloo: the host does
        table = torch.stack(parts)
        if table.is_cuda:
            key = (table.device.index, self.world, int(counts.numel()))
            ring = _PINNED_SIZES.get(key)
            if ring is None:
                ring = _PINNED_SIZES[key] = dict(bufs=[torch.empty(table.shape, dtype=table.dtype).pin_memory()
                                                       for _ in range(4)], at=0)
            ring["at"] = (ring["at"] + 1) % len(ring["bufs"])
            self.host = ring["bufs"][ring["at"]]
            self.host.copy_(table, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()
        else:
            self.host, self.event = table, None

    def result(self) -> List[List[int]]:
        if self.event is not None:
            self.event.synchronize()
        return self.host.tolist()


def exchange_entries_replicated(entries: torch.Tensor, count: int, sizes: List[int], group=None):
    """entries (>= count, W): this rank's list.  Returns [(entries of rank q, count of rank q)] for every rank, in
    rank order (padded all-gather: every rank contributes max(sizes) rows)."""
    if EMULATED_WORLD > 1 and not dist.is_initialized():
        return [(entries, count)] * EMULATED_WORLD
    world = dist.get_world_size(group)
    longest = max(max(sizes), 1)
    own = entries.new_empty((longest, entries.shape[1]))
    own[:count] = entries[:count]
    parts = [torch.empty_like(own) for _ in range(world)]
    dist.all_gather(parts, own, group=group)
    return [(parts[q], int(sizes[q])) for q in range(world)]


def exchange_entries_sharded(entries: torch.Tensor, send_counts: List[int], recv_counts: List[int], group=None):
    """entries (sum(send_counts), W) grouped by destination rank.  Returns [(entries from rank q, count)] in rank order
    (one all-to-all with per-destination split sizes)."""
    if EMULATED_WORLD > 1 and not dist.is_initialized():
        at, out = 0, []
        for q in range(EMULATED_WORLD):  # what this rank would receive: about one owner's share from every peer
            out.append((entries[at:at + recv_counts[q]], int(recv_counts[q])))
            at += send_counts[q]
        return out
    world = dist.get_world_size(group)
    width = entries.shape[1]
    recv = entries.new_empty((max(sum(recv_counts), 1), width))
    dist.all_to_all_single(recv[:sum(recv_counts)], entries[:sum(send_counts)].contiguous(),
                           output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts), group=group)
    out, at = [], 0
    for q in range(world):
        out.append((recv[at:at + recv_counts[q]], int(recv_counts[q])))
        at += recv_counts[q]
    return out


def exchanged_bytes(mode: str, grad_mode: str, world: int, visible: int, touched: List[int], features: int,
                    rank: int = 0) -> dict:
    """payload bytes rank `rank` sends and receives in a frame's exchange step (exact without any hardware: list
    lengths and V are all it depends on).  dense: the ring all-reduce of (V, 7 + F) rows."""
    dense = 4 * (7 + features) * visible
    if mode == "dense" or world == 1:
        wire = int(2 * (world - 1) / max(world, 1) * dense)
        return dict(mode="dense", payload=dense, sent=wire, received=wire)
    entry = 4 * (ENTRY_HEAD + features)
    if grad_mode == "sharded":  # a uniform split over the owners is what a random index order gives
        sent = int(entry * touched[rank] * (world - 1) / world)
        recv = int(entry * sum(t for q, t in enumerate(touched) if q != rank) / world)
        return dict(mode="sparse/sharded", payload=entry * touched[rank], sent=sent, received=recv, dense_payload=dense)
    longest = max(touched)
    return dict(mode="sparse/replicated", payload=entry * touched[rank], sent=entry * longest,
                received=entry * longest * (world - 1), dense_payload=dense)


class _AllReduceGrads(torch.autograd.Function):
    """Identity in the forward; in the backward the partial gradients of (gaussians2d, features, depths) are summed
    over the process group with the two collectives of `_reduce_partial_gradients`."""

    @staticmethod
    def forward(ctx, group, depth_cols, gaussians2d, features, depths, exchange="dense", row_owner=None):
        """row_owner (V int64, grad_mode "sharded"): the rank that owns each visible Gaussian"""
        ctx.group, ctx.depth_cols, ctx.channels = group, int(depth_cols), int(features.shape[1])
        ctx.exchange, ctx.row_owner = exchange, row_owner
        return gaussians2d.view_as(gaussians2d), features.view_as(features), depths.view_as(depths)

    @staticmethod
    def backward(ctx, g_points, g_features, g_depths):
        ref = next(g for g in (g_points, g_features, g_depths) if g is not None)
        v = ref.shape[0]
        rows = max(v, 1)
        # the fused frame reduces (V', 7 + depth feature columns): [z, z^2] with render_depth, nothing otherwise.
        # Here the depth gradient is already folded to one column; pad to the same width so shapes agree.
        width = 7 + ctx.depth_cols
        c = ctx.channels
        colour = torch.zeros((rows, c), dtype=ref.dtype, device=ref.device)
        splat = torch.zeros((rows, width), dtype=ref.dtype, device=ref.device)
        if g_features is not None:
            colour[:v] = g_features
        if g_points is not None:
            splat[:v, :7] = g_points
        if g_depths is not None and ctx.depth_cols:
            splat[:v, 7] = g_depths.reshape(-1)
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(ctx.group) > 1
        if ctx.exchange == "sparse" and multi:
            # the protocol of the fused frame (module comment), on torch operators: the composed path has no mapper
            # list at hand, so "touched" = the rows that received any gradient
            packed = torch.cat([splat, colour], dim=1)
            touched = (packed != 0).any(dim=1).nonzero().reshape(-1)
            ids = touched.to(torch.int32).view(torch.float32).reshape(-1, 1)
            entries = torch.cat([ids, packed[touched]], dim=1).contiguous()
            rank, world = dist.get_rank(ctx.group), dist.get_world_size(ctx.group)
            if ctx.row_owner is not None:
                send = torch.bincount(ctx.row_owner[touched], minlength=world)  # touched ascends -> owners contiguous
                table = [t.tolist() for t in exchange_sizes(send, ctx.group)[0]]
                lists = exchange_entries_sharded(entries, table[rank], [table[q][rank] for q in range(world)], ctx.group)
            else:
                count = torch.tensor([touched.shape[0]], dtype=torch.int64, device=ref.device)
                sizes = [int(t.item()) for t in exchange_sizes(count, ctx.group)[0]]
                lists = exchange_entries_replicated(entries, int(touched.shape[0]), sizes, ctx.group)
            total = torch.zeros_like(packed)
            for ent, cnt in lists:  # rank order, distinct rows inside a list: the same sums on every rank
                if cnt:
                    total[ent[:cnt, 0].contiguous().view(torch.int32).long()] += ent[:cnt, 1:]
            splat, colour = total[:, :width].contiguous(), total[:, width:].contiguous()
        else:
            second = _reduce_partial_gradients(colour, splat, ctx.group)
            if second is not None:
                second.wait()
        g_d = splat[:v, 7:8].contiguous() if ctx.depth_cols else \
            (g_depths if g_depths is not None else None)
        return None, None, splat[:v, :7].contiguous(), colour[:v], g_d, None, None


def default_ops() -> SimpleNamespace:
    """The HIP operators of this package."""
    from .mapper.tile_mapper import map_to_tiles
    from .perspective.projection import project_with_ndc
    from .rasterizer.function import rasterize_with_tiles
    from .spherical_harmonics import evaluate_sh_at
    return SimpleNamespace(project_with_ndc=project_with_ndc, evaluate_sh_at=evaluate_sh_at,
                           map_to_tiles=map_to_tiles, rasterize_with_tiles=rasterize_with_tiles)


def render_gaussians_sharded(gaussians: Gaussians3D, camera_params: CameraParams,
                             config: RasterConfig = RasterConfig(), use_sh: bool = False, render_depth: bool = False,
                             use_depth16: bool = False, group=None, rank: Optional[int] = None,
                             world_size: Optional[int] = None, ops: Optional[SimpleNamespace] = None,
                             interleave: int = 0, exchange: str = "dense", grad_mode: str = "replicated",
                             owned: Optional[Gaussians3D] = None):
    """Render this rank's rows of the frame.  Returns a `Rendering` whose image tensors hold the owned pixel rows
    of the full image in ascending order: `rendering.bands` lists them as [(y0, y1), ...] (`rendering.strip` is the
    first band -- the whole share of a contiguous shard; index a full-size target with
    `owned_pixel_rows(rendering.bands)`).  After `.backward()` of a loss summed over the ranks, every rank holds the
    full parameter gradients.  `gaussians2d` is in full-image coordinates, identical on every rank.
    `point_visibility` / `point_heuristic` are this rank's share: `reduce_point_statistics` sums them.

    exchange: "dense" = all-reduce of the (V, 7 + F) partial gradient rows; "sparse" = lists of the touched rows only
    (module comment above `EXCHANGES`).  grad_mode "sharded" (implies the sparse exchange): for a sharded optimizer --
    `gaussians` is then the replicated, detached set every rank projects and `owned` (`split_owned`) this rank's rows
    [lo, hi) = `owned_range(rank, world, N)` of it as leaf tensors; after `.backward()` those hold the COMPLETE
    gradients of the Gaussians this rank owns, and nothing else is written (no camera gradients in this mode)."""
    from .renderer import Rendering, compute_depth_variance
    if exchange not in EXCHANGES or grad_mode not in GRAD_MODES:
        raise ValueError(f"exchange {exchange!r} / grad_mode {grad_mode!r}: expected one of {EXCHANGES} / {GRAD_MODES}")
    if (grad_mode == "sharded") != (owned is not None):
        raise ValueError('grad_mode "sharded" takes the rank\'s own rows as `owned` (parallel.split_owned), and only it does')
    if grad_mode == "sharded":
        exchange = "sparse"
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size(group) if dist.is_initialized() else 1
    w, h = (int(x) for x in camera_params.image_size)
    ts = config.tile_size
    shard = shard_for(rank, world_size, h, ts, interleave)
    bands = shard.bands

    def tag(r):
        object.__setattr__(r, "bands", bands)
        object.__setattr__(r, "strip", bands[0] if bands else (min(shard.row_begin * ts, h),) * 2)
        object.__setattr__(r, "shard", shard)
        return r

    if ops is None:
        from .fused import fused_supported, render_fused
        if fused_supported(gaussians, camera_params, use_sh, False):
            # the fused frame (fused.py) on this rank's rows; its backward carries the exchange step
            rng = owned_range(rank, world_size, gaussians.position.shape[0]) if owned is not None else None
            return tag(render_fused(gaussians, camera_params, config, render_depth, use_depth16, shard=shard,
                                    group=group, exchange=exchange, grad_mode=grad_mode, owned=owned, owned_range=rng))
        ops = default_ops()

    row_owner = None
    if owned is not None:
        # composed path: splice the owned leaf rows into the replicated data, so that autograd hands the range-shaped
        # gradients back through the concatenation
        lo, hi = owned_range(rank, world_size, gaussians.position.shape[0])
        data = gaussians.apply(lambda t: t.detach())
        gaussians = Gaussians3D(**{k: torch.cat([getattr(data, k)[:lo], getattr(owned, k), getattr(data, k)[hi:]], 0)
                                   for k in ("position", "log_scaling", "rotation", "alpha_logit", "feature")},
                                batch_size=data.batch_size)

    gaussians2d, depths, indexes, ndc_depths = ops.project_with_ndc(
        *gaussians.shape_tensors(), camera_params.T_camera_world, camera_params.projection,
        camera_params.image_size, camera_params.depth_range, config)
    if use_sh:
        features = ops.evaluate_sh_at(gaussians.feature, gaussians.position.detach(), indexes,
                                      camera_params.camera_position)
    else:
        features = gaussians.feature[indexes]

    # everything upstream of this point is replicated; gradients arriving here are partial sums
    if owned is not None:
        chunk = -(-gaussians.position.shape[0] // world_size)
        row_owner = indexes // chunk
    g2d_r, features_r, depths_r = _AllReduceGrads.apply(group, 2 if render_depth else 0, gaussians2d, features,
                                                        depths, exchange, row_owner)
    raster_features = torch.cat([depths_r, depths_r ** 2, features_r], dim=1) if render_depth else features_r
    F = raster_features.shape[1]

    # band by band on exactly shifted means (the origin is a multiple of the tile size): the reference-shaped
    # operators know nothing of shards
    images, weights, vis, heur, band_heur = [], [], None, None, []
    for y0, y1 in bands:
        shift = torch.zeros((7,), dtype=g2d_r.dtype, device=g2d_r.device)
        shift[1] = float(y0)
        local2d = g2d_r - shift
        o2p, ranges = ops.map_to_tiles(local2d, ndc_depths, image_size=(w, y1 - y0), config=config,
                                       use_depth16=use_depth16)
        raster = ops.rasterize_with_tiles(local2d, raster_features, tile_overlap_ranges=ranges.view(-1, 2),
                                          overlap_to_point=o2p, image_size=(w, y1 - y0), config=config)
        images.append(raster.image)
        weights.append(raster.image_weight)
        if config.compute_visibility:
            vis = raster.visibility if vis is None else vis + raster.visibility
        if config.compute_point_heuristic:
            band_heur.append(raster.point_heuristic)
    if images:
        image, weight = torch.cat(images, 0), torch.cat(weights, 0)
    else:  # no row owned: an empty image that still hangs on the graph, so backward() joins the collectives
        zero = (g2d_r.sum() + raster_features.sum()) * 0.0
        image = torch.zeros((0, w, F), dtype=features.dtype, device=features.device) + zero
        weight = torch.zeros((0, w), dtype=features.dtype, device=features.device)
        if config.compute_visibility:
            vis = torch.zeros((gaussians2d.shape[0],), dtype=features.dtype, device=features.device)
        if config.compute_point_heuristic:
            heur = torch.zeros((gaussians2d.shape[0], 2), dtype=features.dtype, device=features.device)
    img_depth = img_var = None
    if render_depth:
        img_depth, img_var = compute_depth_variance(image[..., :2], weight)
        image = image[..., 2:]
    if band_heur:
        # filled in by each band's backward pass (reference function.py:48-59): one band -> that tensor; several ->
        # reduce_point_statistics() sums them (after backward()), `point_heuristic` itself then holds the first band only
        heur = band_heur[0]
    r = tag(Rendering(image=image, image_weight=weight, depth=img_depth, depth_var=img_var,
                         camera=camera_params, config=config, point_visibility=vis, point_heuristic=heur,
                         points_in_view=indexes, point_depth=depths, gaussians2d=gaussians2d))
    object.__setattr__(r, "band_heuristics", band_heur)
    return r


def reduce_point_statistics(rendering, group=None):
    """`point_visibility` / `point_heuristic` of a sharded rendering cover this rank's rows only (they are sums
    over pixels).  Returns (visibility, heuristic) summed over the ranks -- one small all-reduce of (V,3) floats,
    to be called after `.backward()` (the heuristic is filled in by the backward pass); None where not computed."""
    vis, heur = rendering.point_visibility, rendering.point_heuristic
    bands = getattr(rendering, "band_heuristics", None)
    if bands and len(bands) > 1:
        heur = torch.stack([b.detach() for b in bands]).sum(0)
    parts = [t.reshape(t.shape[0], -1) for t in (vis, heur) if t is not None]
    if not parts:
        return None, None
    packed = torch.cat(parts, dim=1).contiguous()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    out_vis = packed[:, 0].contiguous() if vis is not None else None
    out_heur = packed[:, (1 if vis is not None else 0):].contiguous() if heur is not None else None
    return out_vis, out_heur


def gather_image(shard_image: torch.Tensor, image_height: int, tile_size: int, group=None,
                 interleave: int = 0) -> torch.Tensor:
    """All-gather the ranks' rows into the full (H, W, C) image on every rank (only when a caller needs it)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return shard_image
    shards = [shard_for(r, world, image_height, tile_size, interleave) for r in range(world)]
    max_h = max(s.local_height for s in shards)
    pad = torch.zeros((max_h, *shard_image.shape[1:]), dtype=shard_image.dtype, device=shard_image.device)
    pad[:shard_image.shape[0]] = shard_image
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    full = torch.empty((image_height, *shard_image.shape[1:]), dtype=shard_image.dtype, device=shard_image.device)
    for p, s in zip(parts, shards):
        rows = owned_pixel_rows(s.bands).to(full.device)
        full[rows] = p[:rows.shape[0]]
    return full
