"""Tile mapper: which Gaussians overlap which screen tiles, depth-sorted per tile (HIP).

Operator interface of the reference mapper/tile_mapper.py:202-223:
map_to_tiles(gaussians (V,7), depth (V,1), image_size (W,H), config, use_depth16=False)
  -> (overlap_to_point (K) int32, tile_ranges (Th,Tw,2) int32).

Runs the fused per-tile pipeline of csrc/mapper.hip (histogram -> scan -> bucket -> per-tile sort)
with ONE host read-back (K and the largest tile population).  `map_to_tiles_reference_stages`
runs the reference's own stage sequence (count -> cumsum -> keys -> global radix sort -> ranges)
on the reference-shaped primitives; both produce identical results.
"""
from __future__ import annotations

import math
from numbers import Integral
from typing import Tuple

import torch

from .. import _native as nv
from ..data_types import RasterConfig


def pad_to_tile(image_size: Tuple[Integral, Integral], tile_size: int):
    def pad(x):
        return int(math.ceil(x / tile_size) * tile_size)
    return tuple(pad(x) for x in image_size)


def _validate(gaussians, depth, image_size, config, use_depth16):
    if not isinstance(gaussians, torch.Tensor) or not isinstance(depth, torch.Tensor):
        raise TypeError("gaussians and depth must be torch.Tensor")
    if not (len(image_size) == 2 and all(isinstance(x, Integral) for x in image_size)):
        raise TypeError(f"image_size must be Tuple[Integral, Integral], got {image_size!r}")
    if not isinstance(config, RasterConfig):
        raise TypeError(f"config must be RasterConfig, got {type(config).__name__}")
    if not isinstance(use_depth16, bool):
        raise TypeError("use_depth16 must be bool")
    assert gaussians.ndim == 2 and gaussians.shape[1] == 7, f"gaussians must be Nx7 got {gaussians.shape}"
    assert depth.ndim == 2 and depth.shape[1] == 1, f"depths must be Nx1, got {depth.shape}"
    assert depth.shape[0] == gaussians.shape[0], f"size mismatch {gaussians.shape} vs {depth.shape}"


@torch.no_grad()
@nv.on_tensor_device
def map_to_tiles(gaussians: torch.Tensor, depth: torch.Tensor, image_size: Tuple[Integral, Integral],
                 config: RasterConfig, use_depth16: bool = False, return_keys: bool = False):
    """maps gaussians to tiles, sorted by depth (front to back).

    The reference limits the tile count to < 65535 (16-bit tile id in a 48-bit key,
    tile_mapper.py:29,175); the per-tile sort here has no such limit (up to 2^20 tiles).
    """
    _validate(gaussians, depth, image_size, config, use_depth16)
    g = gaussians.detach().contiguous()
    d = depth.detach().contiguous()
    nv.require_device(g, d, what="map_to_tiles")
    lib = nv.lib()
    dev = g.device
    v = g.shape[0]
    ts = config.tile_size
    wp, hp = pad_to_tile(image_size, ts)
    tile_shape = (hp // ts, wp // ts)
    num_tiles = tile_shape[0] * tile_shape[1]
    cfg = nv.make_config(config)
    w, h = int(image_size[0]), int(image_size[1])

    tile_ranges = torch.empty((*tile_shape, 2), dtype=torch.int32, device=dev)
    counts = torch.empty((4,), dtype=torch.int32, device=dev)
    nbytes = lib.gs_map_scratch_bytes(v, num_tiles)
    scratch = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    nv.check(lib.gs_map_prepare(v, None, nv.ptr(g), w, h, cfg, 0, nv.ptr(tile_ranges), nv.ptr(counts), None,
                                None, None, nv.ptr(scratch), nbytes, nv.stream()), "gs_map_prepare")
    k, max_tile = (int(x) for x in counts[:2].tolist())  # host sync (reference: full_cumsum.cu:45)
    overlap_to_point = torch.empty((k,), dtype=torch.int32, device=dev)
    keys = torch.empty((k,), dtype=torch.int64, device=dev) if return_keys else None
    if k > 0:
        pairs = torch.empty((k,), dtype=torch.int64, device=dev)
        nv.check(lib.gs_map_finish(v, None, k, max_tile, nv.ptr(g), nv.ptr(d), w, h, cfg, int(use_depth16),
                                   nv.ptr(tile_ranges), nv.ptr(overlap_to_point), nv.ptr(keys), nv.ptr(pairs),
                                   None, nv.ptr(scratch), nbytes, nv.stream()), "gs_map_finish")
    if return_keys:
        return overlap_to_point, tile_ranges, keys
    return overlap_to_point, tile_ranges


@torch.no_grad()
@nv.on_tensor_device
def map_to_tiles_reference_stages(gaussians: torch.Tensor, depth: torch.Tensor,
                                  image_size: Tuple[Integral, Integral], config: RasterConfig,
                                  use_depth16: bool = False, return_keys: bool = False):
    """The reference's stage sequence (tile_mapper.py:146-196) on the reference-shaped HIP primitives."""
    from ..hip_lib import full_cumsum, radix_sort_pairs
    _validate(gaussians, depth, image_size, config, use_depth16)
    g = gaussians.detach().contiguous()
    d = depth.detach().contiguous()
    nv.require_device(g, d, what="map_to_tiles")
    lib = nv.lib()
    dev = g.device
    v = g.shape[0]
    ts = config.tile_size
    wp, hp = pad_to_tile(image_size, ts)
    tile_shape = (hp // ts, wp // ts)
    num_tiles = tile_shape[0] * tile_shape[1]
    cfg = nv.make_config(config)
    w, h = int(image_size[0]), int(image_size[1])
    tile_ranges = torch.zeros((*tile_shape, 2), dtype=torch.int32, device=dev)
    empty = torch.empty((0,), dtype=torch.int32, device=dev)
    if v == 0:
        return (empty, tile_ranges, torch.empty((0,), dtype=torch.int64, device=dev)) if return_keys \
            else (empty, tile_ranges)
    counts = torch.empty((v,), dtype=torch.int32, device=dev)
    nv.check(lib.gs_tile_count(v, nv.ptr(g), w, h, cfg, nv.ptr(counts), nv.stream()), "gs_tile_count")
    cum, total = full_cumsum(counts)
    if total == 0:
        return (empty, tile_ranges, torch.empty((0,), dtype=torch.int64, device=dev)) if return_keys \
            else (empty, tile_ranges)
    keys = torch.empty((total,), dtype=torch.int64, device=dev)
    values = torch.empty((total,), dtype=torch.int32, device=dev)
    nv.check(lib.gs_tile_emit_keys(v, nv.ptr(g), nv.ptr(d), nv.ptr(cum), w, h, cfg, int(use_depth16), nv.ptr(keys),
                                   nv.ptr(values), nv.stream()), "gs_tile_emit_keys")
    tile_bits = max(1, (num_tiles - 1).bit_length())
    end_bit = (16 if use_depth16 else 32) + tile_bits
    keys, values = radix_sort_pairs(keys, values, 0, end_bit)
    nv.check(lib.gs_find_ranges(total, nv.ptr(keys), int(use_depth16), num_tiles, nv.ptr(tile_ranges), nv.stream()),
             "gs_find_ranges")
    return (values, tile_ranges, keys) if return_keys else (values, tile_ranges)
