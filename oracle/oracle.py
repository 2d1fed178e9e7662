"""ctypes/numpy front end of the CPU oracle (oracle/gsplat_oracle.cpp).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py.  The product package never imports this module.

All functions take and return numpy arrays (or torch CPU tensors, converted), f32 or f64.
Function-by-function reference citations are in the C++ source.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libgsplat_oracle.so")
# GS_ORACLE_LIB: load another build of the same source instead (the `make -C oracle asan` sanitizer build)
_LIB_OVERRIDE = os.environ.get("GS_ORACLE_LIB")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gsplat_oracle.cpp")
    hdr = os.path.join(_HERE, "..", "include", "gs_detmath.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in (src, hdr))
    if force or stale:
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(os.path.abspath(_LIB_OVERRIDE) if _LIB_OVERRIDE else _LIB_PATH)
        _lib.orc_full_cumsum_i32.restype = ctypes.c_int64
        _lib.orc_num_threads.restype = ctypes.c_int
        _lib.orc_det_logf.restype = ctypes.c_float
        _lib.orc_det_logf.argtypes = [ctypes.c_float]
    return _lib


class _RasterCfg(ctypes.Structure):
    _fields_ = [("tile_size", ctypes.c_int), ("antialias", ctypes.c_int), ("use_alpha_blending", ctypes.c_int),
                ("compute_visibility", ctypes.c_int), ("compute_point_heuristic", ctypes.c_int),
                ("clamp_max_alpha", ctypes.c_double), ("alpha_threshold", ctypes.c_double),
                ("saturate_threshold", ctypes.c_double)]


@dataclass(frozen=True)
class OracleConfig:
    """Field-for-field the reference RasterConfig (data_types.py:13-39)."""
    tile_size: int = 16
    pixel_stride: tuple = (2, 2)
    clamp_margin: float = 0.15
    antialias: bool = False
    blur_cov: float = 0.3
    clamp_max_alpha: float = 0.99
    alpha_threshold: float = 1.0 / 255.0
    saturate_threshold: float = 0.9999
    use_alpha_blending: bool = True
    compute_point_heuristic: bool = False
    compute_visibility: bool = False

    @staticmethod
    def of(cfg) -> "OracleConfig":
        if isinstance(cfg, OracleConfig):
            return cfg
        return OracleConfig(**{k: getattr(cfg, k) for k in OracleConfig.__dataclass_fields__})

    def c_struct(self) -> _RasterCfg:
        return _RasterCfg(self.tile_size, int(self.antialias), int(self.use_alpha_blending),
                          int(self.compute_visibility or self.compute_point_heuristic),
                          int(self.compute_point_heuristic), self.clamp_max_alpha, self.alpha_threshold,
                          self.saturate_threshold)


def _np(x, dtype=None):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    a = np.ascontiguousarray(x)
    if dtype is not None and a.dtype != dtype:
        a = np.ascontiguousarray(a.astype(dtype))
    return a


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _suf(dtype):
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError(f"oracle supports float32/float64, got {dtype}")


def _i64(x):
    return ctypes.c_int64(int(x))


def pad_to_tile(image_size, tile_size):
    return tuple(int(-(-int(x) // tile_size) * tile_size) for x in image_size)


# ------------------------------------------------------------------------------- projection
def project_dense(position, log_scaling, rotation, alpha_logit, T_camera_world, projection, image_size, depth_range,
                  blur_cov=0.0, clamp_margin=0.15, alpha_threshold=1.0 / 255.0):
    """Dense (N-row) projection + visibility mask."""
    pos = _np(position)
    dt = pos.dtype
    s = _suf(dt)
    n = pos.shape[0]
    ls, q, al = _np(log_scaling, dt), _np(rotation, dt), _np(alpha_logit, dt).reshape(-1)
    T, pr = _np(T_camera_world, dt).reshape(16), _np(projection, dt)
    points = np.empty((n, 7), dt)
    depth = np.empty((n,), dt)
    vis = np.empty((n,), np.uint8)
    getattr(lib(), f"orc_project_fwd_{s}")(
        _i64(n), _p(pos), _p(ls), _p(q), _p(al), _p(T), _p(pr), int(image_size[0]), int(image_size[1]),
        ctypes.c_double(depth_range[0]), ctypes.c_double(depth_range[1]), ctypes.c_double(blur_cov),
        ctypes.c_double(clamp_margin), ctypes.c_double(alpha_threshold), _p(points), _p(depth), _p(vis))
    return points, depth, vis.astype(bool)


def project(position, log_scaling, rotation, alpha_logit, T_camera_world, projection, image_size, depth_range,
            blur_cov=0.0, clamp_margin=0.15, alpha_threshold=1.0 / 255.0):
    """Same contract as perspective/projection.py:190-215 apply(): (points (V,7), depth (V,1), indexes (V) i64)."""
    points, depth, vis = project_dense(position, log_scaling, rotation, alpha_logit, T_camera_world, projection,
                                       image_size, depth_range, blur_cov, clamp_margin, alpha_threshold)
    idx = np.nonzero(vis)[0].astype(np.int64)
    return points[idx], depth[idx].reshape(-1, 1), idx


def project_backward(position, log_scaling, rotation, alpha_logit, T_camera_world, projection, image_size, indexes,
                     grad_points, grad_depth, blur_cov=0.0, clamp_margin=0.15):
    pos = _np(position)
    dt = pos.dtype
    s = _suf(dt)
    n = pos.shape[0]
    ls, q, al = _np(log_scaling, dt), _np(rotation, dt), _np(alpha_logit, dt).reshape(-1)
    T, pr = _np(T_camera_world, dt).reshape(16), _np(projection, dt)
    idx = _np(indexes, np.int64)
    gp, gd = _np(grad_points, dt), _np(grad_depth, dt).reshape(-1)
    v = idx.shape[0]
    dpos, dls, dq, dal = np.empty((n, 3), dt), np.empty((n, 3), dt), np.empty((n, 4), dt), np.empty((n, 1), dt)
    dT, dpr = np.empty((4, 4), dt), np.empty((4,), dt)
    getattr(lib(), f"orc_project_bwd_{s}")(
        _i64(n), _i64(v), _p(pos), _p(ls), _p(q), _p(al), _p(T), _p(pr), int(image_size[0]), int(image_size[1]),
        ctypes.c_double(blur_cov), ctypes.c_double(clamp_margin), _p(idx), _p(gp), _p(gd), _p(dpos), _p(dls), _p(dq),
        _p(dal), _p(dT), _p(dpr))
    return dpos, dls, dq, dal, dT, dpr


def ndc_depth(depth, near, far):
    d = _np(depth, np.float32)
    out = np.empty_like(d)
    lib().orc_ndc_depth_f32(_i64(d.size), _p(d), ctypes.c_double(near), ctypes.c_double(far), _p(out))
    return out


# --------------------------------------------------------------------------------------- SH
def _degree(params):
    d = params.shape[2]
    n = int(round(d ** 0.5))
    assert n * n == d, f"SH feature count must be square, got {d}"
    return n - 1


def evaluate_sh_at(params, points, indexes, camera_pos):
    par = _np(params)
    dt = par.dtype
    s = _suf(dt)
    pts, idx, cam = _np(points, dt), _np(indexes, np.int64), _np(camera_pos, dt)
    v, C = idx.shape[0], par.shape[1]
    out = np.empty((v, C), dt)
    getattr(lib(), f"orc_sh_fwd_{s}")(_i64(v), C, _degree(par), _p(par), _p(pts), _p(idx), _p(cam), _p(out))
    return out


def evaluate_sh_at_backward(params, points, indexes, camera_pos, grad_out):
    par = _np(params)
    dt = par.dtype
    s = _suf(dt)
    pts, idx, cam, go = _np(points, dt), _np(indexes, np.int64), _np(camera_pos, dt), _np(grad_out, dt)
    n, C = par.shape[0], par.shape[1]
    dpar, dpts, dcam = np.empty_like(par), np.empty_like(pts), np.empty((3,), dt)
    getattr(lib(), f"orc_sh_bwd_{s}")(_i64(n), _i64(idx.shape[0]), C, _degree(par), _p(par), _p(pts), _p(idx), _p(cam),
                                      _p(go), _p(dpar), _p(dpts), _p(dcam))
    return dpar, dpts, dcam


# ----------------------------------------------------------------------------------- mapper
def tile_counts(gaussians2d, image_size_padded, tile_size, alpha_threshold):
    g = _np(gaussians2d, np.float32)
    counts = np.empty((g.shape[0],), np.int32)
    lib().orc_tile_counts(_i64(g.shape[0]), _p(g), int(image_size_padded[0]), int(image_size_padded[1]),
                          int(tile_size), ctypes.c_double(alpha_threshold), _p(counts))
    return counts


def full_cumsum(counts):
    c = _np(counts, np.int32)
    out = np.empty((c.shape[0] + 1,), np.int32)
    total = lib().orc_full_cumsum_i32(_i64(c.shape[0]), _p(c), _p(out))
    return out, int(total)


def radix_sort_pairs(keys, values, begin_bit=0, end_bit=-1):
    k = _np(keys, np.uint64)
    v = _np(values, np.int32)
    ko, vo = np.empty_like(k), np.empty_like(v)
    lib().orc_sort_pairs_u64(_i64(k.shape[0]), _p(k), _p(v), int(begin_bit), int(end_bit), _p(ko), _p(vo))
    return ko, vo


def map_to_tiles(gaussians2d, depth, image_size, config=OracleConfig(), use_depth16=False, return_keys=False):
    """mapper/tile_mapper.py:169-223: (overlap_to_point (K) i32, tile_ranges (Th,Tw,2) i32)."""
    cfg = OracleConfig.of(config)
    g = _np(gaussians2d, np.float32)
    d = _np(depth, np.float32).reshape(-1)
    assert g.ndim == 2 and g.shape[1] == 7, f"gaussians must be Nx7 got {g.shape}"
    ts = cfg.tile_size
    Wp, Hp = pad_to_tile(image_size, ts)
    tile_shape = (Hp // ts, Wp // ts)
    ranges = np.zeros((*tile_shape, 2), np.int32)
    if g.shape[0] == 0:
        out = (np.empty((0,), np.int32), ranges)
        return (*out, np.empty((0,), np.uint64)) if return_keys else out
    counts = tile_counts(g, (Wp, Hp), ts, cfg.alpha_threshold)
    cum, total = full_cumsum(counts)
    if total == 0:
        out = (np.empty((0,), np.int32), ranges)
        return (*out, np.empty((0,), np.uint64)) if return_keys else out
    keys = np.empty((total,), np.uint64)
    vals = np.empty((total,), np.int32)
    lib().orc_tile_emit(_i64(g.shape[0]), _p(g), _p(d), _p(cum), Wp, Hp, ts, ctypes.c_double(cfg.alpha_threshold),
                        int(use_depth16), _p(keys), _p(vals))
    # reference: 48 bits (32 of depth + 16 of tile id), or 32 with 16-bit depth codes, and at most 65534 tiles
    # (tile_mapper.py:29,154,175).  Past that limit -- which the build lifts -- the tile id simply takes the bits it
    # needs above the depth field: key = tile << (16 | 32) | depth code, as a u64.
    num_tiles = tile_shape[0] * tile_shape[1]
    tile_bits = 16 if num_tiles < 65535 else max(1, (num_tiles - 1).bit_length())
    keys, vals = radix_sort_pairs(keys, vals, 0, (16 if use_depth16 else 32) + tile_bits)
    lib().orc_tile_ranges(_i64(total), _p(keys), int(use_depth16), _i64(tile_shape[0] * tile_shape[1]), _p(ranges))
    return (vals, ranges, keys) if return_keys else (vals, ranges)


# ------------------------------------------------------------------------------- rasterizer
def rasterize_with_tiles(gaussians2d, features, overlap_to_point, tile_overlap_ranges, image_size,
                         config=OracleConfig()):
    """rasterizer/function.py:96-127 forward: (image (H,W,F), alpha (H,W), visibility (V) or None)."""
    cfg = OracleConfig.of(config)
    g = _np(gaussians2d)
    dt = g.dtype
    s = _suf(dt)
    f = _np(features, dt)
    o2p, rng = _np(overlap_to_point, np.int32), _np(tile_overlap_ranges, np.int32).reshape(-1, 2)
    W, H = int(image_size[0]), int(image_size[1])
    V, F = g.shape[0], f.shape[1]
    ts = cfg.tile_size
    assert rng.shape[0] == (-(-W // ts)) * (-(-H // ts)), "tile range count does not match image size"
    image, alpha = np.empty((H, W, F), dt), np.empty((H, W), dt)
    c = cfg.c_struct()
    vis = np.zeros((V,), dt)
    getattr(lib(), f"orc_raster_fwd_{s}")(_i64(V), F, _p(g), _p(f), _p(rng), _p(o2p), W, H, ctypes.byref(c),
                                          _p(image), _p(alpha), _p(vis))
    return image, alpha, (vis if c.compute_visibility else None)


def rasterize_backward(gaussians2d, features, overlap_to_point, tile_overlap_ranges, image_size, image, grad_image,
                       config=OracleConfig()):
    """rasterizer/function.py:79-91: (grad_gaussians2d (V,7), grad_features (V,F), heuristic (V,2) or None)."""
    cfg = OracleConfig.of(config)
    g = _np(gaussians2d)
    dt = g.dtype
    s = _suf(dt)
    f = _np(features, dt)
    o2p, rng = _np(overlap_to_point, np.int32), _np(tile_overlap_ranges, np.int32).reshape(-1, 2)
    im, gi = _np(image, dt), _np(grad_image, dt)
    W, H = int(image_size[0]), int(image_size[1])
    V, F = g.shape[0], f.shape[1]
    gg, gf, heur = np.empty((V, 7), dt), np.empty((V, F), dt), np.zeros((V, 2), dt)
    c = cfg.c_struct()
    getattr(lib(), f"orc_raster_bwd_{s}")(_i64(V), F, _p(g), _p(f), _p(rng), _p(o2p), _i64(o2p.shape[0]), W, H,
                                          ctypes.byref(c), _p(im), _p(gi), _p(gg), _p(gf), _p(heur))
    return gg, gf, (heur if cfg.compute_point_heuristic else None)


def raster_flip_margin(gaussians2d, overlap_to_point, tile_overlap_ranges, image_size, config=OracleConfig(),
                       features=None):
    """(H, W) f32: per pixel, min over its tile's splats of |alpha - alpha_threshold| / alpha_threshold (diagnostic
    used by the parity tests to prove that an out-of-tolerance pixel is an `alpha > threshold` decision two f32
    implementations round to different sides, forward.py:100)."""
    cfg = OracleConfig.of(config)
    g = _np(gaussians2d, np.float32)
    o2p, rng = _np(overlap_to_point, np.int32), _np(tile_overlap_ranges, np.int32).reshape(-1, 2)
    W, H = int(image_size[0]), int(image_size[1])
    margin = np.empty((H, W), np.float32)
    c = cfg.c_struct()
    if features is None:
        lib().orc_raster_flip_margin_f32(_p(g), None, 0, _p(rng), _p(o2p), W, H, ctypes.byref(c), _p(margin), None)
        return margin
    # with features: also (H, W, F), per pixel and channel the largest |feature| among the splats it blends
    f = _np(features, np.float32)
    feat_max = np.empty((H, W, f.shape[1]), np.float32)
    lib().orc_raster_flip_margin_f32(_p(g), _p(f), int(f.shape[1]), _p(rng), _p(o2p), W, H, ctypes.byref(c),
                                     _p(margin), _p(feat_max))
    return margin, feat_max


def rasterize(gaussians2d, depth, features, image_size, config=OracleConfig(), use_depth16=False):
    o2p, ranges = map_to_tiles(_np(gaussians2d, np.float32), depth, image_size, config, use_depth16)
    return rasterize_with_tiles(gaussians2d, features, o2p, ranges.reshape(-1, 2), image_size, config), (o2p, ranges)


def num_threads() -> int:
    return int(lib().orc_num_threads())


def set_num_threads(n: int) -> None:
    lib().orc_set_num_threads(int(n))
