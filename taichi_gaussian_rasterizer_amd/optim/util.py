"""Per-group optimizer state accessors (reference optim/util.py:5-38).  CHECKPOINT FORMAT, kept exactly: the
reference stores the FIRST moment (N, D) under the key 'v' and the SECOND moment -- (N) for vector groups, (N, D) for
scalar ones -- under the key 'm' (its `m, v = get_vector_state(...)` unpacks ('v', 'm') in that order,
optim/fractional.py:117-120).  A state dict written by the reference therefore loads into this package unchanged
(`ParameterClass.load_state_dict`), and vice versa."""
from __future__ import annotations

import torch


def _rows(param: torch.Tensor) -> torch.Tensor:
    return param.view(param.shape[0], -1)


def get_vector_state(state: dict, param: torch.Tensor):
    """(first moment (N, D), second moment (N))"""
    if "v" not in state:
        state["v"] = torch.zeros_like(_rows(param))
        state["m"] = torch.zeros(param.shape[0], dtype=param.dtype, device=param.device)
    return state["v"], state["m"]


def get_scalar_state(state: dict, param: torch.Tensor):
    """(first moment (N, D), second moment (N, D))"""
    if "v" not in state:
        state["v"] = torch.zeros_like(_rows(param))
        state["m"] = torch.zeros_like(_rows(param))
    return state["v"], state["m"]


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
