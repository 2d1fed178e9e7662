"""Input types of the render path: RasterConfig, Gaussians3D, Gaussians2D.

Mirrors the reference taichi_splatting/data_types.py:13-121 (same field names, defaults, derived
properties and methods).  The reference builds Gaussians3D/2D with tensordict's @tensorclass;
tensordict is not a dependency here, so a small tensor-record base class supplies the part of
that interface the render path and its callers use (batch_size, indexing, to/cuda/cpu,
apply, requires_grad_, detach, clone, items, from_tensordict/to_tensordict on plain dicts).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

import torch


@dataclass(frozen=True, eq=True, kw_only=True)
class RasterConfig:
    """Reference data_types.py:13-39.  Frozen and hashable: every field selects a kernel
    specialisation or is passed to the kernels as a scalar."""
    tile_size: int = 16

    # pixel tiling per thread in the backward pass.  The HIP backward always runs one wave64 per
    # tile (4 pixels per lane at tile_size 16); results do not depend on this field
    # (reference rasterizer/tiling.py:35-65 only remaps threads to pixels).
    pixel_stride: Tuple[int, int] = (2, 2)

    # clamp position to within this margin of the image for the affine jacobian
    clamp_margin: float = 0.15

    antialias: bool = False
    blur_cov: float = 0.3

    clamp_max_alpha: float = 0.99
    alpha_threshold: float = 1.0 / 255.0

    # stop alpha blending at this point (backward pass; quantile level when not blending)
    saturate_threshold: float = 0.9999

    use_alpha_blending: bool = True

    compute_point_heuristic: bool = False  # implies compute_visibility
    compute_visibility: bool = False

    # NOT a reference field (the last one, with a default: reference call sites, which pass none of it, are
    # unaffected).  The reference's forward blends down a tile's whole list (rasterizer/forward.py:84-128); this
    # forward stops a 16x16 region once every pixel of it has less than forward_cut of its transmittance left, which
    # changes a pixel by < forward_cut * max|feature|.  0.0 is the closest setting to the reference, not a literal
    # reproduction: the kernel then stops at 2^-25, where the reference's own f32 W += w stops changing W, while the
    # reference goes on adding alpha * (1 - W) * feature with 1 - W stuck at ~2^-24 -- a difference below
    # (remaining splats) * 2^-24 * max|feature| that no reference fixture pins.  With render_depth the cut is divided
    # by far^2 (the z^2 feature).
    forward_cut: float = 2.0 ** -20

    def __post_init__(self):
        if not isinstance(self.tile_size, int) or isinstance(self.tile_size, bool):
            raise TypeError(f"tile_size must be int, got {type(self.tile_size).__name__}")
        if not (isinstance(self.pixel_stride, tuple) and len(self.pixel_stride) == 2
                and all(isinstance(x, int) for x in self.pixel_stride)):
            raise TypeError(f"pixel_stride must be Tuple[int, int], got {self.pixel_stride!r}")
        for name in ("clamp_margin", "blur_cov", "clamp_max_alpha", "alpha_threshold", "saturate_threshold",
                     "forward_cut"):
            if not isinstance(getattr(self, name), float):
                raise TypeError(f"{name} must be float, got {type(getattr(self, name)).__name__}")
        for name in ("antialias", "use_alpha_blending", "compute_point_heuristic", "compute_visibility"):
            if not isinstance(getattr(self, name), bool):
                raise TypeError(f"{name} must be bool, got {type(getattr(self, name)).__name__}")


def check_packed3d(packed_gaussians: torch.Tensor):
    assert len(packed_gaussians.shape) == 2 and packed_gaussians.shape[1] == 11, \
        f"Expected shape (N, 11), got {packed_gaussians.shape}"


def check_packed2d(packed_gaussians: torch.Tensor):
    # the packed 2D gaussian is 7 floats [mean.xy, axis.xy, sigma.xy, alpha]
    # (reference taichi_lib/generic.py:31-41; data_types.py:48-49 still says 6)
    assert len(packed_gaussians.shape) == 2 and packed_gaussians.shape[1] == 7, \
        f"Expected shape (N, 7), got {packed_gaussians.shape}"


class _TensorRecord:
    """Minimal stand-in for tensordict.tensorclass: a record of tensors sharing batch_size."""

    _tensor_fields: Tuple[str, ...] = ()

    def _init_record(self, batch_size, kwargs):
        for name in self._tensor_fields:
            if name not in kwargs:
                raise TypeError(f"{type(self).__name__}: missing field '{name}'")
            value = kwargs[name]
            if not isinstance(value, torch.Tensor):
                raise TypeError(f"{type(self).__name__}.{name} must be a torch.Tensor, got {type(value).__name__}")
            object.__setattr__(self, name, value)
        extra = set(kwargs) - set(self._tensor_fields)
        if extra:
            raise TypeError(f"{type(self).__name__}: unexpected fields {sorted(extra)}")
        first = getattr(self, self._tensor_fields[0])
        if batch_size is None:
            batch_size = (first.shape[0],)
        batch_size = torch.Size(tuple(int(b) for b in batch_size))
        for name in self._tensor_fields:
            t = getattr(self, name)
            if tuple(t.shape[:len(batch_size)]) != tuple(batch_size):
                raise RuntimeError(f"{type(self).__name__}.{name}: shape {tuple(t.shape)} does not start with "
                                   f"batch_size {tuple(batch_size)}")
        object.__setattr__(self, "batch_size", batch_size)

    # --- tensorclass-like interface ---
    def items(self):
        return [(name, getattr(self, name)) for name in self._tensor_fields]

    def keys(self):
        return list(self._tensor_fields)

    def values(self):
        return [getattr(self, name) for name in self._tensor_fields]

    def to_tensordict(self):
        return dict(self.items())

    def to_dict(self):
        return dict(self.items())

    @classmethod
    def from_tensordict(cls, td):
        d = {k: td[k] for k in cls._tensor_fields}
        return cls(**d, batch_size=tuple(d[cls._tensor_fields[0]].shape[:1]))

    def _map(self, fn, batch_size=None):
        out = {name: fn(t) for name, t in self.items()}
        if batch_size is None:
            batch_size = (out[self._tensor_fields[0]].shape[0],) if out[self._tensor_fields[0]].ndim > 0 else ()
        return type(self)(**out, batch_size=batch_size)

    def apply(self, fn, batch_size=None):
        return self._map(fn, batch_size)

    def to(self, *args, **kwargs):
        return self._map(lambda t: t.to(*args, **kwargs), self.batch_size)

    def cuda(self, device=None):
        return self._map(lambda t: t.cuda(device), self.batch_size)

    def cpu(self):
        return self._map(lambda t: t.cpu(), self.batch_size)

    def detach(self):
        return self._map(lambda t: t.detach(), self.batch_size)

    def clone(self):
        return self._map(lambda t: t.clone(), self.batch_size)

    def contiguous(self):
        return self._map(lambda t: t.contiguous(), self.batch_size)

    def requires_grad_(self, requires_grad: bool = True):
        for _, t in self.items():
            t.requires_grad_(requires_grad)
        return self

    @property
    def device(self):
        return getattr(self, self._tensor_fields[0]).device

    @property
    def dtype(self):
        return getattr(self, self._tensor_fields[0]).dtype

    @property
    def shape(self):
        return self.batch_size

    def __len__(self):
        return int(self.batch_size[0])

    def __getitem__(self, index):
        return self._map(lambda t: t[index])

    def __repr__(self):
        body = ", ".join(f"{k}={tuple(v.shape)}" for k, v in self.items())
        return f"{type(self).__name__}({body}, batch_size={tuple(self.batch_size)}, device={self.device})"


class Gaussians3D(_TensorRecord):
    """Reference data_types.py:53-94.

    position (N,3), log_scaling (N,3), rotation (N,4) quaternion **xyzw** (the reference comment
    says wxyz but taichi_lib/generic.py:420 unpacks x,y,z,w), alpha_logit (N,1),
    feature (N,C) or (N,C,(deg+1)^2).
    """
    _tensor_fields = ("position", "log_scaling", "rotation", "alpha_logit", "feature")

    def __init__(self, *, position, log_scaling, rotation, alpha_logit, feature, batch_size=None):
        self._init_record(batch_size, dict(position=position, log_scaling=log_scaling, rotation=rotation,
                                           alpha_logit=alpha_logit, feature=feature))
        assert self.position.shape[1] == 3, f"Expected shape (N, 3), got {self.position.shape}"
        assert self.log_scaling.shape[1] == 3, f"Expected shape (N, 3), got {self.log_scaling.shape}"
        assert self.rotation.shape[1] == 4, f"Expected shape (N, 4), got {self.rotation.shape}"
        assert self.alpha_logit.shape[1] == 1, f"Expected shape (N, 1), got {self.alpha_logit.shape}"

    def packed(self):
        return torch.cat([self.position, self.log_scaling, self.rotation, self.alpha_logit], dim=-1)

    def shape_tensors(self):
        return (self.position, self.log_scaling, self.rotation, self.alpha_logit)

    @property
    def scale(self):
        return torch.exp(self.log_scaling)

    @property
    def alpha(self):
        return torch.sigmoid(self.alpha_logit)

    def replace(self, **kwargs):
        d = dict(self.items())
        d.update(kwargs)
        return Gaussians3D(**d, batch_size=self.batch_size)

    def concat(self, other):
        return Gaussians3D(
            position=torch.cat([self.position, other.position], dim=0),
            log_scaling=torch.cat([self.log_scaling, other.log_scaling], dim=0),
            rotation=torch.cat([self.rotation, other.rotation], dim=0),
            alpha_logit=torch.cat([self.alpha_logit, other.alpha_logit], dim=0),
            feature=torch.cat([self.feature, other.feature], dim=0),
            batch_size=(self.batch_size[0] + other.batch_size[0],))


def inverse_sigmoid(x: torch.Tensor):
    return torch.log(x / (1 - x))


class Gaussians2D(_TensorRecord):
    """Reference data_types.py:101-121."""
    _tensor_fields = ("position", "z_depth", "log_scaling", "rotation", "alpha_logit", "feature")

    def __init__(self, *, position, z_depth, log_scaling, rotation, alpha_logit, feature, batch_size=None):
        self._init_record(batch_size, dict(position=position, z_depth=z_depth, log_scaling=log_scaling,
                                           rotation=rotation, alpha_logit=alpha_logit, feature=feature))

    @property
    def opacity(self):
        return self.alpha_logit.sigmoid()

    @property
    def scaling(self):
        return torch.exp(self.log_scaling)

    def set_scaling(self, scaling) -> "Gaussians2D":
        return self.replace(log_scaling=torch.log(scaling))

    def replace(self, **kwargs):
        d = dict(self.items())
        d.update(kwargs)
        return Gaussians2D(**d, batch_size=self.batch_size)


__all__ = ["RasterConfig", "Gaussians3D", "Gaussians2D", "check_packed2d", "check_packed3d", "inverse_sigmoid"]
