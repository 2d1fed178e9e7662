"""Finite-value checks over tensors and containers of tensors (reference torch_lib/util.py:5-47: `check_finite`,
`count_nonfinite`).  Containers: sequences, mappings and this package's tensor records (the reference walks
tensorclasses the same way)."""
from __future__ import annotations

from typing import Mapping, Sequence

import torch


def _record_items(t):
    items = getattr(t, "items", None)
    return list(items()) if callable(items) and not isinstance(t, Mapping) else None


def count_nonfinite(t, name: str, warn: bool = False) -> dict:
    """{path: number of non-finite entries}, only for the tensors (and their .grad) that have any"""
    found = {}

    def visit(value, path):
        if isinstance(value, torch.Tensor):
            for label, tensor in ((path, value), (f"{path}.grad", value.grad)):
                if tensor is not None:
                    bad = int((~torch.isfinite(tensor)).sum())
                    if bad:
                        found[label] = bad
        elif isinstance(value, Mapping):
            for key, item in value.items():
                visit(item, f"{path}.{key}")
        elif isinstance(value, Sequence) and not isinstance(value, str):
            for i, item in enumerate(value):
                visit(item, f"{path}[{i}]")
        else:
            fields = _record_items(value)
            if fields is not None:
                for key, item in fields:
                    visit(item, f"{path}.{key}")

    visit(t, name)
    return found


def check_finite(t, name: str, warn: bool = False) -> None:
    """raise (or, with warn=True, print) when anything under `t` holds NaN / inf"""
    found = count_nonfinite(t, name, warn)
    if found:
        if not warn:
            raise ValueError(f"Non-finite entries: {found}")
        print(f"Non-finite entries: {found}")
