"""Per-group optimizer state accessors (reference optim/util.py:5-38).  Note the reference's return order: the FIRST
value is what its kernels receive as the first moment (N, D), the second the second moment -- (N) for vector groups --
although it keeps them under the keys 'v' and 'm'.  This package stores first/second moment as 'm'/'v'
(fractional.py `_Rows.moments`); these helpers return the same pair in the same order."""
from __future__ import annotations

import torch


def _rows(param: torch.Tensor) -> torch.Tensor:
    return param.view(param.shape[0], -1)


def get_vector_state(state: dict, param: torch.Tensor):
    if "m" not in state:
        state["m"] = torch.zeros_like(_rows(param))
        state["v"] = torch.zeros(param.shape[0], dtype=param.dtype, device=param.device)
    return state["m"], state["v"]


def get_scalar_state(state: dict, param: torch.Tensor):
    if "m" not in state:
        state["m"] = torch.zeros_like(_rows(param))
        state["v"] = torch.zeros_like(_rows(param))
    return state["m"], state["v"]


def _float_counter(state: dict, key: str, shape, device) -> torch.Tensor:
    if key not in state:
        state[key] = torch.zeros(shape, device=device, dtype=torch.float32)
    return state[key]


def get_total_weight(state: dict, n: int, device: torch.device) -> torch.Tensor:
    return _float_counter(state, "total_weight", n, device)


def get_running_vis(state: dict, shape: tuple, device: torch.device) -> torch.Tensor:
    return _float_counter(state, "running_vis", shape, device)


def flatten_param(param: torch.Tensor):
    """(parameter, gradient) as (N, D) matrices"""
    return _rows(param), param.grad.view(param.shape[0], -1)
