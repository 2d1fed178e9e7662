"""Adam moment kernels under the reference's module name (optim/fractional_adam.py:7-85); see _kernels.py."""
from . import _kernels

ALGORITHM = 0


def scalar_kernel(betas=(0.9, 0.999), eps=1e-16, bias_correction=True):
    """second moment per element: m, v both (N, D)"""
    return _kernels.make(ALGORITHM, False, tuple(betas), eps, None, bias_correction)


def vector_kernel(betas=(0.9, 0.999), eps=1e-16, dims=3, bias_correction=True):
    """one second moment per row (running squared norm of the row's gradient): m (N, dims), v (N)"""
    return _kernels.make(ALGORITHM, True, tuple(betas), eps, int(dims), bias_correction)
