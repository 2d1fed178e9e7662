"""Sparse / visibility-weighted optimizers that consume the renderer's gradients (reference optim/).

FractionalAdam, FractionalLaProp, SparseAdam, SparseLaProp (optim/fractional.py:152-222) and
VisibilityAwareAdam, VisibilityAwareLaProp (optim/visibility_aware.py:53-124): same constructor
arguments, param-group keys (`name`, `type` in {"scalar","vector","local_vector"}, `mask_lr`, `point_lr`)
and `step(indexes, weight|visibility, basis=None)` signatures; the per-row moment update runs in a HIP
kernel (gs_optim_step).  `ParameterClass` (optim/parameter_class.py: parameters + optimizer whose per-row state
follows pruning and densification) is provided on a plain tensor table instead of tensordict.
"""
from .autograd import restore_grad
from .fractional import FractionalAdam, FractionalLaProp, SparseAdam, SparseLaProp
from .parameter_class import ParameterClass, TensorTable
from .visibility_aware import VisibilityAwareAdam, VisibilityAwareLaProp

__all__ = ['FractionalAdam', 'FractionalLaProp', 'SparseAdam', 'SparseLaProp', 'VisibilityAwareAdam',
           'VisibilityAwareLaProp', 'ParameterClass', 'TensorTable', 'restore_grad']
