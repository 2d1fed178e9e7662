"""ctypes binding of libgsplat_hip.so (include/gsplat_hip.h).

The HIP library is the ONLY compute backend of this package: there is no CPU fallback and no
torch re-implementation behind these calls.  If the shared library is missing, or a tensor is not
a float32 CUDA(HIP) tensor, the call raises -- it never silently routes elsewhere.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_double, c_float, c_int32, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# GS_LIB_PATH: developer override to A/B two builds of the library in one session (tools/ab_lib.sh)
LIB_PATH = os.environ.get("GS_LIB_PATH") or os.path.join(_HERE, "libgsplat_hip.so")
CSRC = os.path.join(_HERE, "csrc")


class GsRasterConfig(ctypes.Structure):
    """include/gsplat_hip.h GsRasterConfig (mirrors reference data_types.py:13-39)."""
    _fields_ = [("tile_size", c_int32), ("pixel_stride_x", c_int32), ("pixel_stride_y", c_int32),
                ("antialias", c_int32), ("use_alpha_blending", c_int32), ("compute_point_heuristic", c_int32),
                ("compute_visibility", c_int32), ("clamp_margin", c_float), ("blur_cov", c_float),
                ("clamp_max_alpha", c_float), ("alpha_threshold", c_float), ("saturate_threshold", c_float),
                ("forward_cut", c_float), ("tune_wave_sub_blocks", c_int32), ("tune_no_heavy_split", c_int32)]


class GsRowShard(ctypes.Structure):
    """include/gsplat_hip.h GsRowShard: the tile rows of the full image one call covers."""
    _fields_ = [("row_begin", c_int32), ("row_end", c_int32), ("band", c_int32), ("period", c_int32),
                ("phase", c_int32)]


class GsFrame(ctypes.Structure):
    """include/gsplat_hip.h GsFrame: one render_gaussians frame for gs_frame_fwd / gs_frame_bwd"""
    _fields_ = [("n", c_int64), ("channels", c_int32), ("sh_degree", c_int32), ("width", c_int32), ("height", c_int32),
                ("near_plane", c_double), ("far_plane", c_double), ("render_depth", c_int32), ("use_depth16", c_int32),
                ("render_median_depth", c_int32), ("prepare_backward", c_int32), ("k_capacity", c_int64),
                ("max_tile_hint", c_int32), ("has_shard", c_int32), ("shard", GsRowShard), ("cfg", GsRasterConfig),
                ("depth_forward_cut", c_float), ("exchange_world", c_int32), ("exchange_rank", c_int32)]


class GsFrameFork(ctypes.Structure):
    """include/gsplat_hip.h GsFrameFork: side stream + two events for the colour stage of gs_frame_fwd"""
    _fields_ = [("side_stream", c_void_p), ("fork_event", c_void_p), ("join_event", c_void_p)]


class GsFrameLayout(ctypes.Structure):
    """include/gsplat_hip.h GsFrameLayout: byte offsets of the frame's sub-buffers (-1 = absent)"""
    _OFFSETS = ("workspace_bytes", "fwd_scratch_bytes", "bwd_scratch_bytes", "stage_bytes",
                "counts", "camera_pos", "points", "depth", "features", "indexes", "slot_of", "tile_ranges", "tile_order",
                "overlap_to_point", "image", "alpha", "visibility", "out_image", "img_depth", "img_var", "median",
                "grad_rows", "touched", "owner_counts", "s_ndc_depth", "s_pairs", "s_median_cover", "s_stage", "b_grad_image", "b_camera",
                "b_grad_rows")
    _fields_ = [(name, c_int64) for name in _OFFSETS] + [
        ("num_features", c_int32), ("grad_row_floats", c_int32), ("tiles_x", c_int32), ("tiles_y", c_int32),
        ("local_height", c_int32)]


# Developer tuning aids (tools/exp_*.py, the wave-region tests): passed per call inside GsRasterConfig; the
# library itself reads no environment variable.  GS_RASTER_NB / GS_RASTER_HEAVY seed them at import.
TUNING = {"wave_sub_blocks": int(os.environ.get("GS_RASTER_NB", "0") or 0),
          "no_heavy_split": 1 if os.environ.get("GS_RASTER_HEAVY", "1") == "0" else 0}


_CFG = POINTER(GsRasterConfig)
_FRAME = POINTER(GsFrame)
_SHARD = POINTER(GsRowShard)
_P = c_void_p
_I32, _I64, _F64 = c_int32, c_int64, c_double

# name -> (restype, argtypes); every symbol declared in include/gsplat_hip.h
SIGNATURES = {
    "gs_last_error": (ctypes.c_char_p, []),
    "gs_version": (ctypes.c_int, []),
    "gs_project_scratch_bytes": (_I64, [_I64]),
    "gs_project_fwd": (ctypes.c_int, [_I64, _P, _P, _P, _P, _P, _P, _I32, _I32, _F64, _F64, _CFG, _P, _P, _P, _P, _P,
                                       _P, _P, _I32, _P, _P, _I64, _P]),
    "gs_project_bwd_scratch_bytes": (_I64, [_I64]),
    "gs_project_bwd": (ctypes.c_int, [_I64, _I64, _P, _P, _P, _P, _P, _P, _I32, _I32, _CFG, _P, _P, _I32, _P, _P,
                                       _I32, _P, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "gs_camera_position": (ctypes.c_int, [_P, _P, _P]),
    "gs_sh_fwd": (ctypes.c_int, [_I64, _P, _I32, _I32, _P, _P, _P, _P, _P, _I32, _P]),
    "gs_sh_fwd_rows": (ctypes.c_int, [_I64, _P, _P, _I32, _I32, _P, _P, _P, _P, _P, _I32, _P]),
    "gs_sh_fwd_shard": (ctypes.c_int, [_I64, _P, _I32, _I32, _P, _P, _P, _P, _P, _I32, _CFG, _SHARD, _P, _I32, _P]),
    "gs_shard_pack_grads": (ctypes.c_int, [_I64, _I32, _I32, _P, _P, _P, _P, _P]),
    "gs_sh_bwd": (ctypes.c_int, [_I64, _I64, _I32, _I32, _P, _P, _P, _I32, _P, _P, _P, _I32, _P, _I32, _P, _P, _P,
                                  _P]),
    "gs_map_scratch_bytes": (_I64, [_I64, _I64]),
    "gs_map_touched_offset": (_I64, [_I64, _I64]),
    "gs_shard_pack_sparse": (ctypes.c_int, [_I64, _P, _I32, _I32, _P, _P, _P, _P]),
    "gs_shard_add_sparse": (ctypes.c_int, [_I64, _P, _I32, _I32, _I64, _P, _P, _P]),
    "gs_shard_merge_sparse": (ctypes.c_int, [_I32, _P, _P, _I32, _I32, _I64, _P, _P, _P, _P, _I64, _P]),
    "gs_map_touched_list": (ctypes.c_int, [_I64, _P, _I64, _P, _I64, _P, _P, _P, _I64, _I32, _P, _I32, _P, _P]),
    "gs_map_prepare": (ctypes.c_int, [_I64, _P, _P, _I32, _I32, _CFG, _I64, _P, _P, _P, _P, _SHARD, _P, _I64, _P]),
    "gs_map_finish": (ctypes.c_int, [_I64, _P, _I64, _I32, _P, _P, _I32, _I32, _CFG, _I32, _P, _P, _P, _P, _SHARD,
                                      _P, _I64, _P]),
    "gs_selftest_detmath": (ctypes.c_int, [_I64, _P, _P, _P, _P]),
    "gs_tile_count": (ctypes.c_int, [_I64, _P, _I32, _I32, _CFG, _P, _P]),
    "gs_cumsum_scratch_bytes": (_I64, [_I64]),
    "gs_full_cumsum_i32": (ctypes.c_int, [_I64, _P, _P, _P, _I64, _P]),
    "gs_tile_emit_keys": (ctypes.c_int, [_I64, _P, _P, _P, _I32, _I32, _CFG, _I32, _P, _P, _P]),
    "gs_sort_scratch_bytes": (_I64, [_I64, _I32]),
    "gs_radix_sort_pairs": (ctypes.c_int, [_I64, _I32, _P, _P, _P, _P, _I32, _I32, _P, _I64, _P]),
    "gs_find_ranges": (ctypes.c_int, [_I64, _P, _I32, _I64, _P, _P]),
    "gs_raster_fwd": (ctypes.c_int, [_I64, _I32, _P, _P, _P, _P, _I64, _I32, _I32, _CFG, _P, _P, _P, _P, _P, _SHARD,
                                      _P]),
    "gs_grad_row_floats": (_I32, [_I32]),
    "gs_raster_bwd": (ctypes.c_int, [_I64, _I32, _P, _P, _P, _P, _I64, _I32, _I32, _CFG, _P, _P, _P, _P, _P, _SHARD,
                                      _P]),
    "gs_raster_bwd_unpack": (ctypes.c_int, [_I64, _I32, _P, _P, _P, _P, _P]),
    "gs_segmented_sort_pairs": (ctypes.c_int, [_I64, _I32, _P, _P, _P, _P, _I64, _P, _P, _P, _I64, _P]),
    "gs_optim_visibility_weights": (ctypes.c_int, [_I64, _P, _P, _P, _P, c_float, c_float, c_float, _P, _P, _P]),
    "gs_optim_step": (ctypes.c_int, [_I32, _I32, _I64, _I32, _P, _P, _P, _P, _P, _P, c_float, c_float, c_float, c_float,
                                      _I32, _P, _P, _P, _P, _P, _P]),
    "gs_morton_codes64": (ctypes.c_int, [_I64, _P, _P, c_float, _I32, _P, _P]),
    "gs_feature_gather_fwd": (ctypes.c_int, [_I64, _P, _I32, _P, _P, _P, _I32, _P]),
    "gs_feature_gather_bwd": (ctypes.c_int, [_I64, _I32, _P, _P, _I32, _P, _P]),
    "gs_depth_split_fwd": (ctypes.c_int, [_I64, _I32, _P, _P, c_float, _P, _P, _P, _P]),
    "gs_depth_split_bwd": (ctypes.c_int, [_I64, _I32, _P, _P, c_float, _P, _P, _P, _P, _P]),
    "gs_frame_layout": (ctypes.c_int, [_FRAME, POINTER(GsFrameLayout)]),
    "gs_frame_fwd": (ctypes.c_int, [_FRAME, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _I64, _P, _P,
                                     POINTER(GsFrameFork), _P, _P]),
    "gs_frame_bwd": (ctypes.c_int, [_FRAME, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _I64, _I64, _I64, _P, _P, _P,
                                     _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
}

_lib = None


class KernelTimer:
    """Optional per-entry-point GPU timing with HIP events recorded on the launch stream
    (bench.py uses it to measure the dominant kernel live inside the timed region)."""

    def __init__(self):
        self.enabled = False
        self.only = None   # optional set of entry-point names to time (None = all)
        self.records = {}  # name -> list of (start_event, end_event)

    def reset(self):
        self.records = {}

    def summary(self):
        """name -> (calls, total_ms); call after torch.cuda.synchronize()."""
        return {name: (len(ev), sum(a.elapsed_time(b) for a, b in ev)) for name, ev in self.records.items()}


timer = KernelTimer()

# stages of gs_frame_fwd / gs_frame_bwd (include/gsplat_hip.h GS_FWD_* / GS_BWD_*), under the names of the entry points
# they stand for, so that bench.py's per-stage table and byte formulas read the same either way
FRAME_FWD_STAGES = ("gs_project_fwd", "gs_sh_fwd", "gs_map_prepare", "gs_map_finish", "gs_raster_fwd")
FRAME_BWD_STAGES = ("gs_raster_bwd", "gs_sh_bwd", "gs_project_bwd")


def stage_events(names):
    """None, or a ctypes array of hipEvent_t pairs for the stages of a frame call that `timer` is asked to time; the
    (start, stop) torch events are filed under the stage names in `timer.records`."""
    if not timer.enabled:
        return None
    arr = (c_void_p * (2 * len(names)))()
    used = False
    for k, name in enumerate(names):
        if timer.only is not None and name not in timer.only:
            continue
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        b.record()  # creates the hipEvent_t handles the library records again around the stage
        arr[2 * k], arr[2 * k + 1] = a.cuda_event, b.cuda_event
        timer.records.setdefault(name, []).append((a, b))
        used = True
    return arr if used else None


class _TimedLib:
    """Attribute access returns the ctypes function; launches are bracketed by events when
    `timer.enabled` (size queries and gs_last_error are never timed)."""

    def __init__(self, handle):
        self._h = handle

    def __getattr__(self, name):  # first lookup only: the result is stored on the instance
        fn = getattr(self._h, name)
        if name.endswith("_bytes") or name in ("gs_last_error", "gs_version", "gs_grad_row_floats", "gs_frame_layout",
                                               "gs_frame_fwd", "gs_frame_bwd", "gs_map_touched_offset"):
            setattr(self, name, fn)
            return fn

        def call(*args):
            if not timer.enabled or (timer.only is not None and name not in timer.only):
                return fn(*args)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = fn(*args)
            b.record()
            timer.records.setdefault(name, []).append((a, b))
            return rc
        setattr(self, name, call)
        return call


def build(force: bool = False, jobs: int = 6) -> str:
    """Compile libgsplat_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, f"-j{jobs}"]
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=True)
    res = subprocess.run(args, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"building libgsplat_hip.so failed:\n{res.stdout[-4000:]}\n{res.stderr[-4000:]}")
    return LIB_PATH


def lib():
    """Load the HIP library; raise loudly if it is not there (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is the only backend of taichi_gaussian_rasterizer_amd. "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C taichi_gaussian_rasterizer_amd/csrc`).")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = _TimedLib(handle)
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().gs_last_error().decode("utf-8", "replace")
        if rc == -2:
            raise NotImplementedError(f"{what}: {msg}")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise RuntimeError(f"{what} failed ({rc}): {msg}")


_CONFIGS = {}  # (RasterConfig (frozen, hashable), cut scale, tuning) -> its C struct


def make_config(config, cut_scale: float = 1.0) -> GsRasterConfig:
    """cut_scale divides RasterConfig.forward_cut: the bound on what the forward's early stop drops is
    forward_cut * max|feature|, so callers blending large features (z^2 up to far^2) pass that magnitude."""
    key = (config, float(cut_scale), TUNING["wave_sub_blocks"], TUNING["no_heavy_split"])
    cached = _CONFIGS.get(key)
    if cached is None:
        cached = _CONFIGS[key] = _build_config(config, cut_scale)
    return cached


def make_shard(shard):
    """None, or a parallel.RowShard -> byref(GsRowShard) for the C-ABI."""
    if shard is None:
        return None
    return ctypes.byref(GsRowShard(int(shard.row_begin), int(shard.row_end), int(shard.band), int(shard.period),
                                   int(shard.phase)))


def _build_config(config, cut_scale: float = 1.0) -> GsRasterConfig:
    return GsRasterConfig(
        int(config.tile_size), int(config.pixel_stride[0]), int(config.pixel_stride[1]), int(config.antialias),
        int(config.use_alpha_blending), int(config.compute_point_heuristic),
        int(config.compute_visibility or config.compute_point_heuristic), float(config.clamp_margin),
        float(config.blur_cov), float(config.clamp_max_alpha), float(config.alpha_threshold),
        float(config.saturate_threshold), float(getattr(config, "forward_cut", 0.0)) / max(float(cut_scale), 1.0),
        int(TUNING["wave_sub_blocks"]), int(TUNING["no_heavy_split"]))


def on_tensor_device(fn):
    """Run `fn` with the CUDA(HIP) current device set to the device of its first GPU tensor argument.  Every kernel
    of the library launches on the CURRENT device's stream, and torch allocates scratch where the inputs live: with
    tensors on cuda:1 while cuda:0 is current, device-1 pointers would go to a launch on device 0.  Wrapped around
    every autograd forward / backward and every plain operator of the package."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        for a in args:
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if a.device.index == torch.cuda.current_device():
                    return fn(*args, **kwargs)
                with torch.cuda.device(a.device):
                    return fn(*args, **kwargs)
        return fn(*args, **kwargs)
    return wrapper


def check_same_device(tensors) -> None:
    """raises if GPU tensors of one call live on different devices (pure logic: testable without a GPU)"""
    seen = None
    for t in tensors:
        if t is None or not getattr(t, "is_cuda", False):
            continue
        if seen is None:
            seen = t.device
        elif t.device != seen:
            raise RuntimeError(f"tensors of one call on different devices: {seen} and {t.device}")


def require_device(*tensors: torch.Tensor, dtype=torch.float32, what="tensor") -> None:
    """The operators run on MI355X only.  Anything else is an error, not a fallback."""
    check_same_device(tensors)
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                f"{what}: got a {t.device.type} tensor; taichi_gaussian_rasterizer_amd runs on a HIP device only "
                "(there is no CPU path in the product; the CPU oracle lives in oracle/ for tests)")
        if dtype is not None and t.dtype != dtype:
            raise TypeError(f"{what}: expected {dtype}, got {t.dtype} (the HIP kernels are float32; the reference's "
                            "float64 instantiations exist for gradcheck only)")


def ptr(t) -> c_void_p:
    return c_void_p(0) if t is None else c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream() -> c_void_p:
    """the current device's current stream as a hipStream_t (torch.cuda.current_stream() costs ~25 us of Python per
    call, the raw accessor a fraction of one)"""
    if _raw_stream is not None:
        return c_void_p(_raw_stream(torch.cuda.current_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)
