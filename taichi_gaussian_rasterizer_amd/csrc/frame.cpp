// frame.cpp -- one C-ABI call per direction for a whole render_gaussians frame (reference renderer.py:134-231).
//
// gs_frame_fwd / gs_frame_bwd enqueue every stage of the fused frame -- the same entry points, in the same order and
// with the same arguments as the stage-by-stage composition (taichi_gaussian_rasterizer_amd/fused.py: results are
// bit-identical) -- from ONE host call into ONE caller-provided workspace whose sub-buffers are carved by offset
// (gs_frame_layout).  What this removes is host time: ~25 ctypes crossings and ~20 tensor allocations per frame
// (0.40 ms of Python per frame, the binding term for training-size images and for the ranks of a sharded frame).
// Host only: no kernels here, no allocation, no synchronisation; the one event record (counts_event) lets the caller
// wait for the mapper's counts while the sort and the rasterizer are still running.

#include <string.h>

#include "gs_common.h"

namespace {

int64_t a256(int64_t bytes) { return gs_align_up(bytes > 0 ? bytes : 1, 256); }

struct Dims {
  int64_t n, P, T;
  int C, F, col0, RS, tiles_x, tiles_y, local_h, D;
  bool want_vis;
};

int frame_dims(const GsFrame* f, Dims* d) {
  GS_REQUIRE(f != nullptr, GS_ERR_INVALID_ARGUMENT, "frame: descriptor is NULL");
  if (int rc = gs_check_cfg(&f->cfg)) return rc;
  GS_REQUIRE(f->n >= 0 && f->n < (int64_t(1) << 31), GS_ERR_INVALID_ARGUMENT, "frame: %lld gaussians", (long long)f->n);
  GS_REQUIRE(f->width > 0 && f->height > 0, GS_ERR_INVALID_ARGUMENT, "frame: image size %dx%d", f->width, f->height);
  GS_REQUIRE(f->sh_degree >= -1 && f->sh_degree <= 3, GS_ERR_UNSUPPORTED, "frame: SH degree %d", f->sh_degree);
  GS_REQUIRE(f->channels >= 1 && f->channels <= (f->sh_degree >= 0 ? GS_MAX_SH_CHANNELS : GS_MAX_FEATURES - 2),
             GS_ERR_UNSUPPORTED, "frame: %d feature channels", f->channels);
  const int ts = f->cfg.tile_size;
  GsShard sh;
  if (int rc = gs_make_shard(f->has_shard ? &f->shard : nullptr, int(gs_div_up(f->height, ts)), &sh)) return rc;
  d->n = f->n;
  d->C = f->channels;
  d->F = f->channels + (f->render_depth ? 2 : 0);
  d->col0 = d->F - d->C;
  d->RS = gs_grad_row_floats(d->F);
  d->D = f->sh_degree >= 0 ? (f->sh_degree + 1) * (f->sh_degree + 1) : 1;
  d->tiles_x = int(gs_div_up(f->width, ts));
  d->tiles_y = sh.local_rows;
  // pixel rows this call produces: a shard's owned tile rows, the last one cut at the image's bottom edge
  int64_t rows = 0;
  if (!f->has_shard) rows = f->height;
  else
    for (int l = 0; l < sh.local_rows; ++l) {
      const int gy = gs_shard_global_row(sh, l);
      const int64_t y0 = int64_t(gy) * ts, y1 = y0 + ts < f->height ? y0 + ts : f->height;
      rows += y1 - y0;
    }
  d->local_h = int(rows);
  d->T = int64_t(d->tiles_x) * d->tiles_y;
  d->P = rows * f->width;
  d->want_vis = f->cfg.compute_visibility || f->cfg.compute_point_heuristic;
  return GS_OK;
}

template <typename T>
T* at(void* base, int64_t off) { return reinterpret_cast<T*>(static_cast<char*>(base) + off); }

// optional per-stage timing: events[2 k] / events[2 k + 1] are recorded on `stream` in front of / behind stage k
struct StageTimer {
  void* const* events;
  int rc = GS_OK;
  void mark(int stage, int end, void* stream) {
    if (!events || !events[2 * stage + end]) return;
    if (hipEventRecord(static_cast<hipEvent_t>(events[2 * stage + end]), static_cast<hipStream_t>(stream)) != hipSuccess) {
      gs_set_error("frame: hipEventRecord of a stage timer failed");
      rc = GS_ERR_LAUNCH;
    }
  }
};

}  // namespace

int gs_project_fwd_ex(int64_t n, const float* position, const float* log_scaling, const float* rotation,
                      const float* alpha_logit, const float* T_camera_world, const float* projection, int32_t width,
                      int32_t height, double near_plane, double far_plane, const GsRasterConfig* cfg, float* points,
                      float* depth, float* ndc_depth, int64_t* indexes, int32_t* slot_of, int32_t* num_visible,
                      float* depth_features, int32_t depth_features_stride, float* camera_pos, void* scratch,
                      int64_t scratch_bytes, float* zero_rows, int32_t zero_row_floats, const GsMapBinPlan* bin,
                      void* stream);

extern "C" int gs_frame_layout(const GsFrame* f, GsFrameLayout* out) {
  Dims d;
  if (int rc = frame_dims(f, &d)) return rc;
  GS_REQUIRE(out != nullptr, GS_ERR_INVALID_ARGUMENT, "gs_frame_layout: layout is NULL");
  GS_REQUIRE(f->k_capacity >= 1, GS_ERR_INVALID_ARGUMENT, "gs_frame_layout: k_capacity must be >= 1");
  memset(out, 0, sizeof(*out));
  int64_t p = 0;
  auto take = [&](int64_t bytes) { const int64_t o = p; p += a256(bytes); return o; };
  const int64_t n = d.n > 0 ? d.n : 1;
  // ---- workspace: everything the outputs and the backward refer to
  out->counts = take(8 * 4);
  out->camera_pos = take(3 * 4);
  out->points = take(n * 7 * 4);
  out->depth = take(n * 4);
  out->features = take(n * d.F * 4);
  out->indexes = take(n * 8);
  out->slot_of = take(n * 4);
  out->tile_ranges = take(d.T * 8);
  out->tile_order = take(d.T * 4);
  out->overlap_to_point = take(f->k_capacity * 4);
  out->image = take(d.P * d.F * 4);
  out->alpha = take(d.P * 4);
  out->visibility = d.want_vis ? take(n * 4) : -1;
  out->out_image = f->render_depth ? take(d.P * d.C * 4) : -1;
  out->img_depth = f->render_depth ? take(d.P * 4) : -1;
  out->img_var = f->render_depth ? take(d.P * 4) : -1;
  out->median = f->render_median_depth ? take(d.P * 4) : -1;
  out->grad_rows = f->prepare_backward ? take(n * d.RS * 4) : -1;
  const bool lists = f->has_shard && f->exchange_world > 0;
  out->touched = lists ? take(n * 4) : -1;
  out->owner_counts = lists ? take(64 * 8) : -1;
  out->workspace_bytes = p;
  // ---- forward scratch (dead when gs_frame_fwd's work has run).  s_stage = [mapper scratch | projection scratch]: the
  // projection's compaction pass reads its staged rows while it fills the mapper's region arrays (GsMapBinPlan)
  p = 0;
  out->s_ndc_depth = take(n * 4);
  out->s_pairs = take(f->k_capacity * 8);
  out->s_median_cover = f->render_median_depth ? take(d.P * 4) : -1;
  const int64_t pb = gs_project_scratch_bytes(d.n), mb = gs_map_scratch_bytes(d.n, d.T > 0 ? d.T : 1);
  out->s_stage = take(a256(mb) + pb);
  out->stage_bytes = a256(mb) + pb;
  out->fwd_scratch_bytes = p;
  // ---- backward scratch
  p = 0;
  out->b_grad_image = f->render_depth ? take(d.P * d.F * 4) : -1;
  out->b_camera = take(gs_project_bwd_scratch_bytes(d.n));
  out->b_grad_rows = f->prepare_backward ? -1 : take(n * d.RS * 4);
  out->bwd_scratch_bytes = p;
  out->num_features = d.F;
  out->grad_row_floats = d.RS;
  out->tiles_x = d.tiles_x;
  out->tiles_y = d.tiles_y;
  out->local_height = d.local_h;
  return GS_OK;
}

extern "C" int gs_frame_fwd(const GsFrame* f, const float* position, const float* log_scaling, const float* rotation,
                            const float* alpha_logit, const float* feature, const float* T_camera_world,
                            const float* projection, void* workspace, int64_t workspace_bytes, void* scratch,
                            int64_t scratch_bytes, int32_t* counts_host, void* counts_event, const GsFrameFork* fork,
                            void* const* stage_events, void* stream) {
  StageTimer tm{stage_events};
  Dims d;
  if (int rc = frame_dims(f, &d)) return rc;
  GsFrameLayout L;
  if (int rc = gs_frame_layout(f, &L)) return rc;
  GS_REQUIRE(workspace && workspace_bytes >= L.workspace_bytes, GS_ERR_SCRATCH_TOO_SMALL,
             "gs_frame_fwd: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)L.workspace_bytes);
  GS_REQUIRE(scratch && scratch_bytes >= L.fwd_scratch_bytes, GS_ERR_SCRATCH_TOO_SMALL,
             "gs_frame_fwd: scratch %lld < %lld bytes", (long long)scratch_bytes, (long long)L.fwd_scratch_bytes);
  GS_REQUIRE(d.n > 0, GS_ERR_INVALID_ARGUMENT, "gs_frame_fwd: no gaussians (run the composed operators)");
  GS_REQUIRE(feature != nullptr, GS_ERR_INVALID_ARGUMENT, "gs_frame_fwd: feature is NULL");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const GsRowShard* shard = f->has_shard ? &f->shard : nullptr;
  const GsRasterConfig* cfg = &f->cfg;
  // what the forward's early stop may drop is bounded by forward_cut * max|feature|: z^2 reaches far^2
  GsRasterConfig rcfg = f->cfg;
  if (f->render_depth) rcfg.forward_cut = f->depth_forward_cut;
  int32_t* counts = at<int32_t>(workspace, L.counts);
  float* cam_pos = at<float>(workspace, L.camera_pos);
  float* points = at<float>(workspace, L.points);
  float* depth = at<float>(workspace, L.depth);
  float* feats = at<float>(workspace, L.features);
  int64_t* indexes = at<int64_t>(workspace, L.indexes);
  int32_t* slot_of = at<int32_t>(workspace, L.slot_of);
  int32_t* tile_ranges = at<int32_t>(workspace, L.tile_ranges);
  int32_t* tile_order = at<int32_t>(workspace, L.tile_order);
  int32_t* o2p = at<int32_t>(workspace, L.overlap_to_point);
  float* image = at<float>(workspace, L.image);
  float* alpha = at<float>(workspace, L.alpha);
  float* vis = L.visibility >= 0 ? at<float>(workspace, L.visibility) : nullptr;
  float* ndc = at<float>(scratch, L.s_ndc_depth);
  void* stage = at<char>(scratch, L.s_stage);  // the mapper's part: the first map_bytes of it
  const int64_t map_bytes = gs_map_scratch_bytes(d.n, d.T > 0 ? d.T : 1);
  void* proj_stage = at<char>(stage, a256(map_bytes));
  const int64_t proj_bytes = L.stage_bytes - a256(map_bytes);

  const bool lists = shard && f->exchange_world > 0 && d.T > 0;
  GS_REQUIRE(f->exchange_world <= 64 && (f->exchange_world == 0 || (f->exchange_rank >= 0 && f->exchange_rank < f->exchange_world)),
             GS_ERR_INVALID_ARGUMENT, "gs_frame_fwd: exchange rank %d of %d", f->exchange_rank, f->exchange_world);
  if (lists && f->sh_degree >= 0 &&
      hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(feats), 0x3F000000, size_t(d.n) * d.F, s) != hipSuccess) {
    // 0.5 = "not clamped" for the colours of the rows this rank never evaluates (see gs_sh_fwd_shard); before the
    // projection, whose compaction pass writes the depth feature columns
    gs_set_error("gs_frame_fwd: hipMemsetD32Async failed");
    return GS_ERR_LAUNCH;
  }
  // the projection's compaction pass bins the visible rows by screen region for the mapper (one launch less)
  GsMapBinPlan bin;
  bin.width = f->width; bin.height = f->height; bin.cfg = cfg; bin.shard = shard; bin.scratch = stage;
  bin.scratch_bytes = map_bytes;
  tm.mark(GS_FWD_PROJECT, 0, stream);
  if (int rc = gs_project_fwd_ex(d.n, position, log_scaling, rotation, alpha_logit, T_camera_world, projection,
                                 f->width, f->height, f->near_plane, f->far_plane, cfg, points, depth, ndc, indexes,
                                 slot_of, counts, f->render_depth ? feats : nullptr, d.F, cam_pos, proj_stage,
                                 proj_bytes, L.grad_rows >= 0 ? at<float>(workspace, L.grad_rows) : nullptr, d.RS,
                                 d.T > 0 ? &bin : nullptr, stream))
    return rc;
  tm.mark(GS_FWD_PROJECT, 1, stream);
  const int32_t* v_dev = counts;
  float* colours = feats + d.col0;
  int rc;
  // The colours are needed by the rasterizer only, and the tile mapper -- a chain of short, latency-bound launches -- does
  // not need them: with a fork the colour kernel (HBM-bound) runs on the caller's side stream underneath the mapper and
  // joins in front of the rasterizer.  Two event records and two stream waits, all issued from this call.
  const bool forked = fork && fork->side_stream && fork->fork_event && fork->join_event && d.T > 0;
  void* colour_stream = stream;
  if (forked) {
    if (hipEventRecord(static_cast<hipEvent_t>(fork->fork_event), s) != hipSuccess ||
        hipStreamWaitEvent(static_cast<hipStream_t>(fork->side_stream), static_cast<hipEvent_t>(fork->fork_event), 0) !=
            hipSuccess) {
      gs_set_error("gs_frame_fwd: fork onto the side stream failed");
      return GS_ERR_LAUNCH;
    }
    colour_stream = fork->side_stream;
  }
  tm.mark(GS_FWD_COLOURS, 0, colour_stream);
  if (f->sh_degree >= 0 && lists)
    rc = GS_OK;  // after the mapper's first half, on its list of touched rows (below)
  else if (f->sh_degree >= 0 && shard)
    rc = gs_sh_fwd_shard(d.n, v_dev, d.C, f->sh_degree, feature, position, indexes, cam_pos, points, f->height, cfg,
                         shard, colours, d.F, colour_stream);
  else if (f->sh_degree >= 0)
    rc = gs_sh_fwd(d.n, v_dev, d.C, f->sh_degree, feature, position, indexes, cam_pos, colours, d.F, colour_stream);
  else
    rc = gs_feature_gather_fwd(d.n, v_dev, d.C, feature, indexes, colours, d.F, colour_stream);
  tm.mark(GS_FWD_COLOURS, 1, colour_stream);
  if (forked && hipEventRecord(static_cast<hipEvent_t>(fork->join_event),
                               static_cast<hipStream_t>(fork->side_stream)) != hipSuccess) {
    gs_set_error("gs_frame_fwd: join event record failed");
    return GS_ERR_LAUNCH;
  }
  if (rc) return rc;
  if (vis && hipMemsetAsync(vis, 0, size_t(d.n) * 4, s) != hipSuccess) {
    gs_set_error("gs_frame_fwd: hipMemsetAsync failed");
    return GS_ERR_LAUNCH;
  }
  if (d.T == 0) {
    // this rank owns no tile row (more ranks than rows): nothing to map or rasterize, only V is needed
    if (hipMemsetAsync(counts + 1, 0, 28, s) != hipSuccess) { gs_set_error("gs_frame_fwd: memset failed"); return GS_ERR_LAUNCH; }
    if (L.owner_counts >= 0 && hipMemsetAsync(at<char>(workspace, L.owner_counts), 0, 64 * 8, s) != hipSuccess) {
      gs_set_error("gs_frame_fwd: memset failed");
      return GS_ERR_LAUNCH;
    }
    if (counts_host) {
      if (hipMemcpyAsync(counts_host, counts + 4, 16, hipMemcpyDeviceToHost, s) != hipSuccess ||
          hipMemcpyAsync(counts_host + 4, counts, 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
          hipMemcpyAsync(counts_host + 5, counts + 5, 4, hipMemcpyDeviceToHost, s) != hipSuccess) {
        gs_set_error("gs_frame_fwd: count read-back failed");
        return GS_ERR_LAUNCH;
      }
    }
    if (counts_event && hipEventRecord(static_cast<hipEvent_t>(counts_event), s) != hipSuccess) {
      gs_set_error("gs_frame_fwd: hipEventRecord failed");
      return GS_ERR_LAUNCH;
    }
    return GS_OK;
  }
  tm.mark(GS_FWD_MAP_PREPARE, 0, stream);
  if ((rc = gs_map_prepare_ex(d.n, v_dev, points, f->width, f->height, cfg, f->k_capacity, tile_ranges, counts + 4,
                              counts_host, tile_order, shard, stage, map_bytes, 1, stream)))
    return rc;
  tm.mark(GS_FWD_MAP_PREPARE, 1, stream);
  // K, the overflow flag and V are final here and the scan kernel has stored them into the pinned host words itself:
  // the host waits on this event while the sort and the rasterizer are still running
  if (counts_event && hipEventRecord(static_cast<hipEvent_t>(counts_event), s) != hipSuccess) {
    gs_set_error("gs_frame_fwd: hipEventRecord failed");
    return GS_ERR_LAUNCH;
  }
  if (lists) {
    int32_t* touched = at<int32_t>(workspace, L.touched);
    if ((rc = gs_map_touched_list(d.n, v_dev, d.T, stage, map_bytes, touched, counts + 1, indexes, d.n,
                                  f->exchange_world, at<int64_t>(workspace, L.owner_counts), f->exchange_rank,
                                  counts + 2, stream)))
      return rc;
    if (f->sh_degree >= 0) {
      tm.mark(GS_FWD_COLOURS, 0, stream);
      if ((rc = gs_sh_fwd_rows(d.n, touched, counts + 1, d.C, f->sh_degree, feature, position, indexes, cam_pos, colours,
                               d.F, stream)))
        return rc;
      tm.mark(GS_FWD_COLOURS, 1, stream);
    }
  }
  const int32_t tile_hint = f->max_tile_hint > 0 ? -f->max_tile_hint : 0;  // a sizing hint: fuller tiles are still sorted
  tm.mark(GS_FWD_MAP_FINISH, 0, stream);
  if ((rc = gs_map_finish(d.n, v_dev, f->k_capacity, tile_hint, points, ndc, f->width, f->height, cfg, f->use_depth16,
                          tile_ranges, o2p, nullptr, at<char>(scratch, L.s_pairs), shard, stage, map_bytes,
                          stream)))
    return rc;
  tm.mark(GS_FWD_MAP_FINISH, 1, stream);
  if (forked && hipStreamWaitEvent(s, static_cast<hipEvent_t>(fork->join_event), 0) != hipSuccess) {
    gs_set_error("gs_frame_fwd: join onto the main stream failed");
    return GS_ERR_LAUNCH;
  }
  tm.mark(GS_FWD_RASTER, 0, stream);
  if ((rc = gs_raster_fwd(d.n, d.F, points, feats, tile_ranges, o2p, f->k_capacity, f->width, f->height, &rcfg,
                          tile_order, counts + 7, image, alpha, vis, shard, stream)))
    return rc;
  tm.mark(GS_FWD_RASTER, 1, stream);
  if (f->render_depth &&
      (rc = gs_depth_split_fwd(d.P, d.C, image, alpha, 1e-6f, at<float>(workspace, L.out_image),
                               at<float>(workspace, L.img_depth), at<float>(workspace, L.img_var), stream)))
    return rc;
  if (f->render_median_depth) {
    // a second, non-blended forward over the same tile lists that keeps the depth of the splat taking each pixel past
    // half opacity (reference renderer.py:203-208); no gradient
    GsRasterConfig pick = f->cfg;
    pick.use_alpha_blending = 0;
    pick.saturate_threshold = 0.5f;
    pick.compute_visibility = 0;
    pick.compute_point_heuristic = 0;
    if ((rc = gs_raster_fwd(d.n, 1, points, depth, tile_ranges, o2p, f->k_capacity, f->width, f->height, &pick,
                            tile_order, counts + 7, at<float>(workspace, L.median),
                            at<float>(scratch, L.s_median_cover), nullptr, shard, stream)))
      return rc;
  }
  return tm.rc;
}

int gs_rows_add(int64_t v, int32_t row_floats, float* rows, const float* add_points, const float* add_depth,
                int32_t depth_col, void* stream);

extern "C" int gs_frame_bwd(const GsFrame* f, const float* position, const float* log_scaling, const float* rotation,
                            const float* alpha_logit, const float* feature, const float* T_camera_world,
                            const float* projection, void* workspace, int64_t workspace_bytes, void* scratch,
                            int64_t scratch_bytes, int64_t v, int64_t k, const float* grad_image,
                            const float* grad_img_depth, const float* grad_img_var, const float* attached_points,
                            const float* attached_depth, float* d_position, float* d_log_scaling, float* d_rotation,
                            float* d_alpha_logit, float* d_feature, float* d_T_camera_world, float* d_projection,
                            float* d_camera_centre, void* const* stage_events, void* stream) {
  StageTimer tm{stage_events};
  Dims d;
  if (int rc = frame_dims(f, &d)) return rc;
  GsFrameLayout L;
  if (int rc = gs_frame_layout(f, &L)) return rc;
  GS_REQUIRE(!f->has_shard, GS_ERR_UNSUPPORTED,
             "gs_frame_bwd: a sharded frame exchanges its partial gradients between the two halves of the backward; "
             "run the stages (gs_raster_bwd, gs_shard_pack_grads, gs_sh_bwd, gs_project_bwd)");
  GS_REQUIRE(workspace && workspace_bytes >= L.workspace_bytes, GS_ERR_SCRATCH_TOO_SMALL,
             "gs_frame_bwd: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)L.workspace_bytes);
  GS_REQUIRE(scratch && scratch_bytes >= L.bwd_scratch_bytes, GS_ERR_SCRATCH_TOO_SMALL,
             "gs_frame_bwd: scratch %lld < %lld bytes", (long long)scratch_bytes, (long long)L.bwd_scratch_bytes);
  GS_REQUIRE(v >= 0 && v <= d.n && k >= 0 && k <= f->k_capacity, GS_ERR_INVALID_ARGUMENT,
             "gs_frame_bwd: v = %lld, k = %lld", (long long)v, (long long)k);
  GS_REQUIRE(d_position && d_log_scaling && d_rotation && d_alpha_logit && d_feature, GS_ERR_INVALID_ARGUMENT,
             "gs_frame_bwd: NULL gradient output");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const GsRasterConfig* cfg = &f->cfg;
  const int32_t* counts = at<int32_t>(workspace, L.counts);
  const float* cam_pos = at<float>(workspace, L.camera_pos);
  const float* points = at<float>(workspace, L.points);
  const float* feats = at<float>(workspace, L.features);
  const int64_t* indexes = at<int64_t>(workspace, L.indexes);
  const int32_t* slot_of = at<int32_t>(workspace, L.slot_of);
  float* rows = L.grad_rows >= 0 ? at<float>(workspace, L.grad_rows) : at<float>(scratch, L.b_grad_rows);
  if (L.grad_rows < 0 && v > 0 && hipMemsetAsync(rows, 0, size_t(v) * d.RS * 4, s) != hipSuccess) {
    gs_set_error("gs_frame_bwd: hipMemsetAsync failed");
    return GS_ERR_LAUNCH;
  }
  int rc;
  const float* g_img = grad_image;
  if (f->render_depth && v > 0 && d.P > 0 && (grad_image || grad_img_depth || grad_img_var)) {
    // assemble the gradient of the rasterized (H, W, 2 + C) image from the three upstream gradients
    float* assembled = at<float>(scratch, L.b_grad_image);
    if ((rc = gs_depth_split_bwd(d.P, d.C, at<float>(workspace, L.img_depth), at<float>(workspace, L.alpha), 1e-6f,
                                 grad_image, grad_img_depth, grad_img_var, assembled, stream)))
      return rc;
    g_img = assembled;
  }
  tm.mark(GS_BWD_RASTER, 0, stream);
  if (g_img && v > 0 && d.P > 0 && k > 0 &&
      (rc = gs_raster_bwd(v, d.F, points, feats, at<int32_t>(workspace, L.tile_ranges),
                          at<int32_t>(workspace, L.overlap_to_point), k, f->width, f->height, cfg,
                          at<int32_t>(workspace, L.tile_order), counts + 7, at<float>(workspace, L.image), g_img, rows,
                          nullptr, stream)))
    return rc;
  tm.mark(GS_BWD_RASTER, 1, stream);
  // gradients a caller attached to the projected splats / depths themselves (e.g. a regulariser)
  const float* extra_depth = nullptr;
  if (v > 0 && (attached_points || (attached_depth && f->render_depth))) {
    if ((rc = gs_rows_add(v, d.RS, rows, attached_points, f->render_depth ? attached_depth : nullptr, 7, stream)))
      return rc;
  }
  if (attached_depth && v > 0 && !f->render_depth) extra_depth = attached_depth;
  tm.mark(GS_BWD_COLOURS, 0, stream);
  if (f->sh_degree >= 0)
    rc = gs_sh_bwd(d.n, v, d.C, f->sh_degree, feature, position, indexes, 1, slot_of, cam_pos, rows + 7 + d.col0, d.RS,
                   feats + d.col0, d.F, d_feature, nullptr, d_camera_centre, stream);
  else
    rc = gs_feature_gather_bwd(d.n, d.C, slot_of, rows + 7 + d.col0, d.RS, d_feature, stream);
  tm.mark(GS_BWD_COLOURS, 1, stream);
  if (rc) return rc;
  const bool camera = d_T_camera_world || d_projection;
  const float* gd = f->render_depth ? rows + 7 : extra_depth;
  const float* gd2 = f->render_depth ? rows + 8 : nullptr;
  tm.mark(GS_BWD_PROJECT, 0, stream);
  rc = gs_project_bwd(d.n, v, position, log_scaling, rotation, alpha_logit, T_camera_world, projection, f->width,
                      f->height, cfg, slot_of, rows, d.RS, gd, gd2, f->render_depth ? d.RS : 1, d_position,
                      d_log_scaling, d_rotation, d_alpha_logit, d_T_camera_world, d_projection,
                      camera ? at<char>(scratch, L.b_camera) : nullptr,
                      camera ? gs_project_bwd_scratch_bytes(d.n) : 0, stream);
  tm.mark(GS_BWD_PROJECT, 1, stream);
  return rc ? rc : tm.rc;
}
