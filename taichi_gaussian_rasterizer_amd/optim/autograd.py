"""`restore_grad` (reference optim/autograd.py:5-17): run a block with fresh zero .grad buffers on the given tensors
and put the previous .grad back afterwards.  The reference needs it around Taichi's kernel.grad calls; here the HIP
adjoints return their gradients, so nothing in the package uses it -- it is kept for callers that do."""
from __future__ import annotations

from contextlib import contextmanager

import torch


@contextmanager
def restore_grad(*tensors: torch.Tensor):
    saved = [t.grad for t in tensors]
    try:
        for t in tensors:
            if t.requires_grad:
                t.grad = torch.zeros_like(t)
        yield
    finally:
        for t, grad in zip(tensors, saved):
            t.grad = grad
