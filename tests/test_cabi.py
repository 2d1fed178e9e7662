"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol that
include/gsplat_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

from taichi_gaussian_rasterizer_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gsplat_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for required in ("gs_project_fwd", "gs_project_bwd", "gs_sh_fwd", "gs_sh_bwd", "gs_map_prepare", "gs_map_finish",
                     "gs_tile_count", "gs_full_cumsum_i32", "gs_tile_emit_keys", "gs_radix_sort_pairs",
                     "gs_segmented_sort_pairs", "gs_optim_step", "gs_optim_visibility_weights", "gs_feature_gather_fwd",
                     "gs_feature_gather_bwd", "gs_morton_codes64",
                     "gs_find_ranges", "gs_raster_fwd", "gs_raster_bwd", "gs_raster_bwd_unpack", "gs_last_error"):
        assert required in names


def test_library_exports_every_declared_symbol():
    _native.build()
    handle = ctypes.CDLL(_native.LIB_PATH)
    for name in declared_functions():
        assert hasattr(handle, name), f"libgsplat_hip.so does not export {name}"


def test_binding_table_matches_header():
    assert sorted(_native.SIGNATURES) == declared_functions()
    lib = _native.lib()
    assert lib.gs_version() >= 1
    assert lib.gs_grad_row_floats(3) == 16 and lib.gs_grad_row_floats(5) == 16 and lib.gs_grad_row_floats(8) == 32
    assert lib.gs_map_scratch_bytes(10, 16384) >= 2 * 16384 * 4
    assert lib.gs_sort_scratch_bytes(1000, 8) > 12000


def test_argument_errors_are_reported_not_crashed():
    """host-side validation runs before any launch, so it can be exercised without a GPU"""
    lib = _native.lib()
    cfg = _native.GsRasterConfig(tile_size=7, alpha_threshold=1 / 255.)
    rc = lib.gs_raster_fwd(0, 3, None, None, None, None, 0, 16, 16, cfg, None, None, None, None, None, None, None)
    assert rc == -2 and b"tile_size" in lib.gs_last_error()
    with pytest.raises(NotImplementedError):
        _native.check(rc, "gs_raster_fwd")
    cfg = _native.GsRasterConfig(tile_size=16, alpha_threshold=1 / 255.)
    rc = lib.gs_raster_fwd(0, 99, None, None, None, None, 0, 16, 16, cfg, None, None, None, None, None, None, None)
    assert rc == -2 and b"feature width" in lib.gs_last_error()
    rc = lib.gs_radix_sort_pairs(10, 3, None, None, None, None, 0, 8, None, 0, None)
    assert rc == -2
    rc = lib.gs_radix_sort_pairs(10, 8, None, None, None, None, 0, 8, None, 0, None)
    assert rc == -1 and b"NULL" in lib.gs_last_error()
    with pytest.raises(ValueError):
        _native.check(rc, "gs_radix_sort_pairs")


def test_row_shard_argument_is_validated():
    """GsRowShard (screen-tile sharding): rows outside the image, a bad interleave, a shard that owns no row -- all
    refused on the host, before any launch; the struct layout seen through ctypes is the header's"""
    lib = _native.lib()
    cfg = _native.GsRasterConfig(tile_size=16, alpha_threshold=1 / 255., forward_cut=2.0 ** -20)
    assert [f[0] for f in _native.GsRowShard._fields_] == ["row_begin", "row_end", "band", "period", "phase"]
    assert ctypes.sizeof(_native.GsRowShard) == 20 and ctypes.sizeof(_native.GsRasterConfig) == 60

    def prepare(shard):
        return lib.gs_map_prepare(0, None, None, 64, 64, cfg, 0, None, None, None, None, ctypes.byref(shard), None, 0, None)

    assert prepare(_native.GsRowShard(0, 9, 4, 1, 0)) == -1 and b"tile rows" in lib.gs_last_error()     # 64 px = 4 rows
    assert prepare(_native.GsRowShard(1, 3, 1, 2, 0)) == -1 and b"interleaved" in lib.gs_last_error()   # must span all
    assert prepare(_native.GsRowShard(0, 4, 1, 2, 2)) == -1 and b"phase" in lib.gs_last_error()
    assert prepare(_native.GsRowShard(2, 2, 4, 1, 0)) == -1 and b"owns no tile row" in lib.gs_last_error()
    # a valid shard gets past the shard checks (and stops at the NULL buffers)
    assert prepare(_native.GsRowShard(1, 3, 4, 1, 0)) == -1 and b"NULL buffer" in lib.gs_last_error()
    # an interleaved shard that deals every band to other owners: rows 0-3 in bands of 2 over 4 owners, owner 3
    assert prepare(_native.GsRowShard(0, 4, 2, 4, 3)) == -1 and b"owns no tile row" in lib.gs_last_error()


def test_shard_only_entry_points_validate_on_the_host():
    """gs_sh_fwd_shard / gs_shard_pack_grads (the sharded frame's own entry points): bad arguments are refused before
    any launch, an empty call is a no-op"""
    lib = _native.lib()
    cfg = _native.GsRasterConfig(tile_size=16, alpha_threshold=1 / 255., forward_cut=2.0 ** -20)
    shard = _native.GsRowShard(0, 9, 4, 1, 0)  # 64 px = 4 tile rows: row_end 9 is outside
    rc = lib.gs_sh_fwd_shard(10, None, 3, 3, None, None, None, None, ctypes.c_void_p(16), 64, cfg,
                             ctypes.byref(shard), None, 3, None)
    assert rc == -1 and b"tile rows" in lib.gs_last_error()
    rc = lib.gs_sh_fwd_shard(10, None, 3, 3, None, None, None, None, None, 64, cfg, None, None, 3, None)
    assert rc == -1 and b"points2d" in lib.gs_last_error()
    rc = lib.gs_sh_fwd_shard(10, None, 3, 7, None, None, None, None, ctypes.c_void_p(16), 64, cfg, None, None, 3, None)
    assert rc == -2 and b"degree" in lib.gs_last_error()
    assert lib.gs_sh_fwd_shard(0, None, 3, 3, None, None, None, None, None, 64, cfg, None, None, 3, None) == 0
    assert lib.gs_shard_pack_grads(10, 3, 3, None, None, None, None, None) == -1     # colours must start inside the row
    assert lib.gs_shard_pack_grads(10, 5, 2, None, None, None, None, None) == -1 and b"NULL" in lib.gs_last_error()
    assert lib.gs_shard_pack_grads(0, 5, 2, None, None, None, None, None) == 0


def test_frame_layout_is_computed_on_the_host():
    """gs_frame_layout (the workspace map of gs_frame_fwd / gs_frame_bwd) is host-only: sub-buffers are 256-byte
    aligned, disjoint and inside the workspace; optional buffers are -1 unless the frame asks for them; the struct
    layouts seen through ctypes are the header's; bad frames are refused before anything is launched"""
    lib = _native.lib()
    cfg = _native.GsRasterConfig(tile_size=16, alpha_threshold=1 / 255., forward_cut=2.0 ** -20)
    assert ctypes.sizeof(_native.GsFrame) == 168 and ctypes.sizeof(_native.GsFrameLayout) == 31 * 8 + 5 * 4 + 4

    def frame(**kw):
        f = _native.GsFrame(n=1000, channels=3, sh_degree=3, width=100, height=70, near_plane=0.1, far_plane=100.0,
                            k_capacity=5000, cfg=cfg)
        for k, v in kw.items():
            setattr(f, k, v)
        return f

    lay = _native.GsFrameLayout()
    assert lib.gs_frame_layout(ctypes.byref(frame()), ctypes.byref(lay)) == 0
    assert (lay.num_features, lay.grad_row_floats, lay.tiles_x, lay.tiles_y, lay.local_height) == (3, 16, 7, 5, 70)
    ws = ["counts", "camera_pos", "points", "depth", "features", "indexes", "slot_of", "tile_ranges", "tile_order",
          "overlap_to_point", "image", "alpha"]
    offs = [getattr(lay, k) for k in ws]
    assert offs == sorted(offs) and offs[0] == 0 and all(o % 256 == 0 for o in offs)
    assert lay.alpha + 100 * 70 * 4 <= lay.workspace_bytes
    assert lay.points - lay.camera_pos >= 12 and lay.depth - lay.points >= 1000 * 28
    for k in ("visibility", "out_image", "img_depth", "img_var", "median", "grad_rows", "touched", "owner_counts",
              "s_median_cover", "b_grad_image"):
        assert getattr(lay, k) == -1, k
    assert lay.b_grad_rows >= 0 and lay.bwd_scratch_bytes >= 1000 * 64
    assert lay.stage_bytes >= max(lib.gs_project_scratch_bytes(1000), lib.gs_map_scratch_bytes(1000, 35))

    full = frame(render_depth=1, render_median_depth=1, prepare_backward=1)
    full.cfg.compute_visibility = 1
    assert lib.gs_frame_layout(ctypes.byref(full), ctypes.byref(lay)) == 0
    assert lay.num_features == 5 and min(lay.visibility, lay.out_image, lay.img_depth, lay.img_var, lay.median,
                                         lay.grad_rows, lay.s_median_cover) > 0 and lay.b_grad_image >= 0
    assert lay.b_grad_rows == -1 and lay.grad_rows + 1000 * 64 <= lay.workspace_bytes

    # a shard: the images hold the owned pixel rows only (tile rows 1-2 of 5, 32 pixel rows; the last tile row is cut)
    sh = frame(has_shard=1, shard=_native.GsRowShard(1, 3, 5, 1, 0))
    assert lib.gs_frame_layout(ctypes.byref(sh), ctypes.byref(lay)) == 0
    assert (lay.tiles_y, lay.local_height) == (2, 32)
    sh = frame(has_shard=1, shard=_native.GsRowShard(3, 5, 5, 1, 0))
    assert lib.gs_frame_layout(ctypes.byref(sh), ctypes.byref(lay)) == 0 and lay.local_height == 70 - 48
    assert lay.touched == -1
    sh = frame(has_shard=1, shard=_native.GsRowShard(3, 5, 5, 1, 0), exchange_world=8, exchange_rank=3)
    assert lib.gs_frame_layout(ctypes.byref(sh), ctypes.byref(lay)) == 0
    assert lay.touched > 0 and lay.owner_counts >= lay.touched + 1000 * 4 and lay.owner_counts + 512 <= lay.workspace_bytes

    assert lib.gs_frame_layout(ctypes.byref(frame(k_capacity=0)), ctypes.byref(lay)) == -1
    assert lib.gs_frame_layout(ctypes.byref(frame(sh_degree=4)), ctypes.byref(lay)) == -2
    assert lib.gs_frame_layout(ctypes.byref(frame(channels=9)), ctypes.byref(lay)) == -2 and b"channels" in lib.gs_last_error()
    assert lib.gs_frame_layout(None, ctypes.byref(lay)) == -1
    # the calls themselves check the buffers before any launch
    f = frame()
    rc = lib.gs_frame_fwd(ctypes.byref(f), None, None, None, None, None, None, None, None, 0, None, 0, None, None, None,
                          None, None)
    assert rc == -4 and b"workspace" in lib.gs_last_error()
    rc = lib.gs_frame_bwd(ctypes.byref(sh), *([None] * 7), None, 0, None, 0, 0, 0, *([None] * 13), None, None)
    assert rc == -2 and b"sharded" in lib.gs_last_error()


def test_sparse_exchange_entry_points_validate_on_the_host():
    """the sparse exchange's entry points (round 3) refuse bad arguments before any launch; empty calls are no-ops"""
    lib = _native.lib()
    assert lib.gs_shard_pack_sparse(10, None, 3, 3, None, None, None, None) == -1          # colours start inside the row
    assert lib.gs_shard_pack_sparse(10, None, 3, 0, None, None, None, None) == -1 and b"NULL" in lib.gs_last_error()
    assert lib.gs_shard_pack_sparse(0, None, 3, 0, None, None, None, None) == 0
    assert lib.gs_shard_add_sparse(10, None, 40, 0, 100, None, None, None) == -1
    assert lib.gs_shard_add_sparse(0, None, 3, 0, 100, None, None, None) == 0
    ptrs, cnts = (ctypes.c_void_p * 2)(), (ctypes.c_int64 * 2)(0, 0)
    assert lib.gs_shard_merge_sparse(65, ptrs, cnts, 3, 0, 100, None, None, None, None, 0, None) == -1
    assert b"lists" in lib.gs_last_error()
    assert lib.gs_shard_merge_sparse(2, ptrs, cnts, 3, 0, 100, ctypes.c_void_p(16), ctypes.c_void_p(16), None, None, 0,
                                     None) == -4 and b"tmp" in lib.gs_last_error()
    assert lib.gs_shard_merge_sparse(2, ptrs, cnts, 3, 0, 0, None, None, None, None, 0, None) == 0
    assert lib.gs_map_touched_list(10, None, 4, None, 0, None, None, None, 10, 2, None, 0, None, None) == -4
    assert lib.gs_map_touched_list(10, None, 4, None, 0, None, None, None, 10, 65, ctypes.c_void_p(16), 0, None,
                                   None) == -1 and b"owners" in lib.gs_last_error()
    assert lib.gs_map_touched_offset(1000, 64) > 0 and lib.gs_map_touched_offset(1000, 64) % 256 == 0
    assert lib.gs_sh_fwd_rows(10, None, None, 3, 3, None, None, None, None, None, 3, None) == -1
    assert b"row list" in lib.gs_last_error()
