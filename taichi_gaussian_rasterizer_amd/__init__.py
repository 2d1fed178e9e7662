"""taichi_gaussian_rasterizer_amd -- MI355X-native tile-based Gaussian-splat rasterizer behind the
operator API of taichi_splatting (reference taichi_splatting/__init__.py:1-33).

Compute backend: hand-written HIP kernels for gfx950 in libgsplat_hip.so (C-ABI:
include/gsplat_hip.h), called through ctypes.  There is no other backend.
"""
from . import hip_lib, perspective
from . import hip_lib as cuda_lib  # drop-in alias for `taichi_splatting.cuda_lib`
from .data_types import Gaussians2D, Gaussians3D, RasterConfig
from .mapper.tile_mapper import map_to_tiles, pad_to_tile
from .perspective import CameraParams
from .rasterizer import RasterOut, rasterize, rasterize_with_tiles
from .renderer import Rendering, render_gaussians
from .spherical_harmonics import evaluate_sh_at
from .taichi_queue import TaichiQueue, taichi_queue

__version__ = "0.1.0"


def install_as_taichi_splatting():
    """Register this package under the name `taichi_splatting` so that callers written against
    the reference (`from taichi_splatting import render_gaussians`, splat-trainer) import it unchanged."""
    import importlib
    import pkgutil
    import sys
    pkg = sys.modules[__name__]
    for info in pkgutil.walk_packages(pkg.__path__, prefix=__name__ + "."):  # every sub-module under both names
        if not info.name.rsplit(".", 1)[-1].startswith("lib"):  # libgsplat_hip.so is a C library, not a module
            importlib.import_module(info.name)
    sys.modules.setdefault("taichi_splatting", pkg)
    for name, mod in list(sys.modules.items()):
        if name.startswith(__name__ + "."):
            sys.modules.setdefault("taichi_splatting" + name[len(__name__):], mod)
    return pkg


__all__ = [
    'render_gaussians', 'Rendering',
    'map_to_tiles', 'pad_to_tile',
    'Gaussians2D', 'Gaussians3D',
    'RasterConfig', 'CameraParams', 'RasterOut',
    'evaluate_sh_at',
    'rasterize', 'rasterize_with_tiles',
    'perspective', 'hip_lib', 'cuda_lib',
    'TaichiQueue', 'taichi_queue',
    'install_as_taichi_splatting',
]
