"""Dev-only loader for the reference's pure-torch oracles (torch_lib).

TEST INFRASTRUCTURE -- only oracle/make_golden.py uses this, and only in the
authoring container where /root/reference exists.  Nothing from the reference
is copied: its modules are imported from where they lie, by file path, with
three tiny in-memory stubs for packages that are not installed here
(beartype: identity decorator; taichi_splatting.{data_types,taichi_queue}:
empty shells so the package __init__, which needs taichi, is never executed).
"""
import importlib.util
import os
import sys
import types

REF = os.environ.get("GS_REFERENCE_ROOT", "/root/reference")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load(modname, relpath):
    path = os.path.join(REF, relpath)
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference_torch_lib():
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference tree not found at {REF}")
    os.environ.setdefault("TORCH_COMPILE_DISABLE", "1")
    import typing

    bt = _stub("beartype", beartype=lambda f: f)
    _stub("beartype.typing", **{k: getattr(typing, k) for k in
                                ("Tuple", "Optional", "NamedTuple", "Sequence", "Callable", "List")})
    bt.typing = sys.modules["beartype.typing"]

    pkg = _stub("taichi_splatting")
    pkg.__path__ = []
    _stub("taichi_splatting.taichi_queue", queued=lambda f: f)
    _stub("taichi_splatting.data_types", Gaussians3D=object, RasterConfig=object)
    tl = _stub("taichi_splatting.torch_lib")
    tl.__path__ = []
    persp = _stub("taichi_splatting.perspective")
    persp.__path__ = []

    params = _load("taichi_splatting.perspective.params", "taichi_splatting/perspective/params.py")
    persp.CameraParams = params.CameraParams
    transforms = _load("taichi_splatting.torch_lib.transforms", "taichi_splatting/torch_lib/transforms.py")
    projection = _load("taichi_splatting.torch_lib.projection", "taichi_splatting/torch_lib/projection.py")
    rsh = _load("taichi_splatting.torch_lib.rsh", "taichi_splatting/torch_lib/rsh.py")
    tl.rsh = rsh
    sh = _load("taichi_splatting.torch_lib.spherical_harmonics", "taichi_splatting/torch_lib/spherical_harmonics.py")
    return types.SimpleNamespace(params=params, transforms=transforms, projection=projection, rsh=rsh, sh=sh)
