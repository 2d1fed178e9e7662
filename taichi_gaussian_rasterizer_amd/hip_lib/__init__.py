"""Scan and radix-sort primitives (HIP) with the interface of the reference's only native
module, taichi_splatting/cuda_lib (cuda_lib/__init__.py:16-41; CUB wrappers full_cumsum.cu,
radix_sort_pairs.cu): full_cumsum, radix_sort_pairs, radix_argsort.  `cuda_lib` is kept as an
alias of this module so `from taichi_splatting import cuda_lib` call sites keep working.
"""
from __future__ import annotations

from typing import Tuple

import torch

from .. import _native as nv

_KEY_BYTES = {torch.int32: 4, torch.uint32: 4, torch.int64: 8, torch.uint64: 8}


@nv.on_tensor_device
def full_cumsum(x: torch.Tensor) -> Tuple[torch.Tensor, int]:
    """Exclusive scan with the total appended: out has x.shape[0]+1 entries; returns (out, total)."""
    assert x.is_cuda, f"full_cumsum: device must be a cuda device, got {x.device}"
    if x.dtype != torch.int32:
        raise RuntimeError("Not yet implemented for data type.")  # reference full_cumsum.cu:64
    if x.shape[0] == 0:
        return x.new_zeros((1,)), 0
    lib = nv.lib()
    x = x.contiguous()
    n = x.shape[0]
    out = x.new_empty((n + 1,))
    nbytes = lib.gs_cumsum_scratch_bytes(n)
    scratch = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
    nv.check(lib.gs_full_cumsum_i32(n, nv.ptr(x), nv.ptr(out), nv.ptr(scratch), nbytes, nv.stream()),
             "gs_full_cumsum_i32")
    return out, int(out[n].item())


@nv.on_tensor_device
def radix_sort_pairs(keys: torch.Tensor, values: torch.Tensor, start_bit=0, end_bit=None):
    """Stable ascending sort of (key, value) pairs on key bits [start_bit, end_bit); returns new tensors.
    Keys are compared as UNSIGNED integers of their width (the mapper's keys are non-negative)."""
    assert keys.is_cuda, f"keys: device must be a cuda device, got {keys.device}"
    assert values.is_cuda, f"values: device must be a cuda device, got {values.device}"
    if keys.dtype not in _KEY_BYTES or values.dtype != torch.int32:
        raise RuntimeError("Not yet implemented for data type(s).")  # reference radix_sort_pairs.cu:66
    assert keys.dim() == 1 and values.dim() == 1 and keys.shape[0] == values.shape[0], \
        "keys and values must be 1D and have the same size"
    if end_bit is None:
        end_bit = -1
    lib = nv.lib()
    keys, values = keys.contiguous(), values.contiguous()
    k = keys.shape[0]
    kb = _KEY_BYTES[keys.dtype]
    keys_out, values_out = torch.empty_like(keys), torch.empty_like(values)
    if k == 0:
        return keys_out, values_out
    nbytes = lib.gs_sort_scratch_bytes(k, kb)
    scratch = torch.empty((nbytes,), dtype=torch.uint8, device=keys.device)
    nv.check(lib.gs_radix_sort_pairs(k, kb, nv.ptr(keys), nv.ptr(values), nv.ptr(keys_out), nv.ptr(values_out),
                                     int(start_bit), int(end_bit), nv.ptr(scratch), nbytes, nv.stream()),
             "gs_radix_sort_pairs")
    return keys_out, values_out


@nv.on_tensor_device
def segmented_sort_pairs(keys: torch.Tensor, values: torch.Tensor, start_offset: torch.Tensor,
                         end_offset: torch.Tensor):
    """Ascending sort of (key, value) pairs inside each segment [start_offset[s], end_offset[s]); returns new
    tensors (reference cuda_lib/segmented_sort_pairs.cu:36-78: int32 or int16 keys, int32 values, int64 offsets).
    Stable; positions outside every segment keep the input pair."""
    for name, t in (("keys", keys), ("values", values), ("start_offset", start_offset), ("end_offset", end_offset)):
        assert t.is_cuda, f"{name}: device must be a cuda device, got {t.device}"
    assert keys.dim() == 1 and values.dim() == 1 and keys.shape[0] == values.shape[0], \
        "keys and values must be 1D and have the same size"
    assert start_offset.dim() == 1 and end_offset.dim() == 1 and start_offset.shape[0] == end_offset.shape[0], \
        "start_offset and end_offset must be 1D and have the same size"
    assert start_offset.dtype == torch.int64 and end_offset.dtype == torch.int64, \
        "start_offset/end_offset must be int64"
    if keys.dtype not in (torch.int32, torch.int16) or values.dtype != torch.int32:
        raise RuntimeError("Not yet implemented for data type.")  # reference segmented_sort_pairs.cu:76
    lib = nv.lib()
    keys, values = keys.contiguous(), values.contiguous()
    keys_out, values_out = keys.clone(), values.clone()
    n, segs = keys.shape[0], start_offset.shape[0]
    if n == 0 or segs == 0:
        return keys_out, values_out
    scratch = torch.empty((n * 8,), dtype=torch.uint8, device=keys.device)
    starts, ends = start_offset.contiguous(), end_offset.contiguous()  # named: a temporary's block could be handed out again
    nv.check(lib.gs_segmented_sort_pairs(n, keys.element_size(), nv.ptr(keys), nv.ptr(values), nv.ptr(keys_out),
                                         nv.ptr(values_out), segs, nv.ptr(starts), nv.ptr(ends), nv.ptr(scratch), n * 8,
                                         nv.stream()), "gs_segmented_sort_pairs")
    return keys_out, values_out


@nv.on_tensor_device
def radix_argsort(keys: torch.Tensor):
    idx = torch.arange(keys.shape[0], dtype=torch.int32, device=keys.device)
    _, idx = radix_sort_pairs(keys, idx)
    return idx


__all__ = ["full_cumsum", "radix_sort_pairs", "segmented_sort_pairs", "radix_argsort"]
