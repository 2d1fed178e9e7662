/* gsplat_hip.h -- C-ABI of libgsplat_hip.so: the MI355X (gfx950) render path behind the
 * taichi_splatting operator API.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - every pointer is a caller-owned DEVICE buffer unless the name ends in _host; nothing is
 *     allocated or freed inside the library; scratch comes from the caller (sizes below).
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     all work is enqueued on it and no entry point synchronises the device.  Host-visible
 *     counts (V, K) are written to device int32 words that the caller reads back itself.
 *   - return value: 0 on success, negative GsStatus on failure; gs_last_error() returns a
 *     thread-local message for the last failure on the calling thread.
 *   - re-entrant, no global mutable state.
 *   - f32 only (the reference's f64 instantiations exist for gradcheck only,
 *     taichi_lib/__init__.py:8-14; f64 checks run against the CPU oracle in tests/).
 *
 * "replaces" lines cite the reference interface each entry point stands in for (paths relative
 * to /root/reference/taichi_splatting/).  The reference-side binding is in INTEGRATION.md.
 */
#ifndef GSPLAT_HIP_H
#define GSPLAT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum GsStatus {
  GS_OK = 0,
  GS_ERR_INVALID_ARGUMENT = -1,
  GS_ERR_UNSUPPORTED = -2,   /* e.g. tile_size not in {8,16,32}, feature width > GS_MAX_FEATURES */
  GS_ERR_LAUNCH = -3,        /* hipGetLastError() after a launch */
  GS_ERR_SCRATCH_TOO_SMALL = -4
} GsStatus;

#define GS_MAX_FEATURES 32
#define GS_MAX_SH_CHANNELS 8

/* Mirrors RasterConfig (data_types.py:13-39) field for field. */
typedef struct GsRasterConfig {
  int32_t tile_size;            /* 8, 16 or 32 */
  int32_t pixel_stride_x;       /* accepted, ignored: thread<->pixel mapping only (rasterizer/tiling.py:35-65) */
  int32_t pixel_stride_y;
  int32_t antialias;
  int32_t use_alpha_blending;
  int32_t compute_point_heuristic;
  int32_t compute_visibility;
  float clamp_margin;
  float blur_cov;
  float clamp_max_alpha;
  float alpha_threshold;
  float saturate_threshold;
  /* Not a reference field.  The reference's forward keeps blending down a tile's whole list (forward.py:84-128);
   * this library stops a 16x16 region once EVERY pixel of it has less than forward_cut of its transmittance left.
   * What is dropped changes a pixel by less than forward_cut * max|feature|.  0 is the setting closest to the
   * reference: values below 2^-25 act as 2^-25, half an ulp of 1, below which the reference's own f32 accumulation
   * W += w no longer changes W (this library carries the transmittance T = 1 - W itself, which would keep
   * shrinking); the reference still adds alpha * (1 - W) * feature there with 1 - W stuck at ~2^-24, so the two differ
   * by less than N * 2^-24 * max|feature| for N remaining splats (no reference fixture pins this).
   * The Python layer passes RasterConfig.forward_cut (default 2^-20), divided by the squared far plane when the
   * z / z^2 depth features are blended, so the bound is relative to the feature magnitude. */
  float forward_cut;
  /* Developer tuning aids, per call (the library reads no environment variable and keeps no global state); results
   * never depend on them.  tune_wave_sub_blocks: 0 = choose the rasterizer's wave region from the grid size, 1 | 2 | 4
   * = force 8x8 / 16x8 / 16x16 pixels per wave.  tune_no_heavy_split: 1 = never give the fullest tiles four
   * workgroups. */
  int32_t tune_wave_sub_blocks;
  int32_t tune_no_heavy_split;
} GsRasterConfig;

/* Screen-tile sharding (the reference has no counterpart): which tile ROWS of the full image one call
 * covers.  Row ty is owned iff row_begin <= ty < row_end and (period <= 1 or (ty / band) % period == phase);
 * owned rows keep their order and are numbered 0, 1, ... ("local rows").  period <= 1: one contiguous strip;
 * period > 1 (with row_begin = 0, row_end = all rows): bands of `band` rows dealt round-robin to `period` owners
 * (interleaved strips, load balance on real scenes).  With a shard, `height` stays the FULL image height, splat
 * coordinates stay full-image coordinates, tile ids / tile_ranges / tile_order are local (local row * tiles_wide +
 * column) and image buffers hold the owned pixel rows only, in order.  NULL = the whole image. */
typedef struct GsRowShard {
  int32_t row_begin, row_end;
  int32_t band, period, phase;
} GsRowShard;

const char* gs_last_error(void);
int gs_version(void);

/* ------------------------------------------------------------------ projection (a2, a3) --
 * replaces: perspective/projection.py:32-80 project_kernel + :146-149 (nonzero + gathers) and
 * torch_lib/projection.py:120-123 ndc_depth (called renderer.py:189).
 *
 * Step 1 projects all n Gaussians into n-row staging buffers and counts the visible ones per
 * 256-thread block; step 2 scans the block counts and compacts, in ascending index order, into
 * buffers the caller sized for n rows (only the first V rows are written).  *num_visible
 * (device int32) receives V.
 *   T_camera_world: 16 floats row-major 4x4 (device), projection: [fx,fy,cx,cy] (device).
 *   points: (V,7) [mean.xy, axis.xy, sigma.xy, alpha]; depth: (V) camera z; ndc_depth: (V)
 *   = 1 - (1/z - (float)(1/far)) / (float)(1/near - 1/far)  (fixed f32 op order);
 *   indexes: (V) int64 ascending; slot_of: (n) int32, compact row of Gaussian i or -1.
 *   depth_features (optional): row i of a (V, depth_features_stride) raster-feature buffer receives
 *   [z, z^2] in its first two columns (renderer.py:191-193, render_depth=True).
 *   camera_pos (optional, 3 floats): receives the camera centre, as gs_camera_position would (saves its launch).
 * scratch: gs_project_scratch_bytes(n).
 */
int64_t gs_project_scratch_bytes(int64_t n);
int gs_project_fwd(int64_t n, const float* position, const float* log_scaling, const float* rotation,
                   const float* alpha_logit, const float* T_camera_world, const float* projection, int32_t width,
                   int32_t height, double near_plane, double far_plane, const GsRasterConfig* cfg, float* points,
                   float* depth, float* ndc_depth, int64_t* indexes, int32_t* slot_of, int32_t* num_visible,
                   float* depth_features, int32_t depth_features_stride, float* camera_pos,
                   void* scratch, int64_t scratch_bytes, void* stream);

/* replaces: perspective/projection.py:84-118 indexed_project_kernel.grad (Taichi autodiff,
 * :166-185).  Hand-derived adjoint.  Dense gradients (rows of culled Gaussians are zero);
 * d_T_camera_world is 16 floats (last row zero) and d_projection 4 floats, summed over the
 * visible set (the reference sums its per-point expanded copies, :212-213).  Either may be NULL.
 * grad_points (row stride grad_points_stride, <= 0 means 7), grad_depth and grad_depth_sq (stride
 * grad_depth_stride, <= 0 means 1) may be NULL (treated as zero); grad_depth_sq is the gradient of a
 * z^2 feature and contributes 2 z g (so the rasterizer's gradient rows can be consumed in place).
 * scratch: gs_project_bwd_scratch_bytes(n) (per-block camera partials; unused when both camera
 * outputs are NULL).
 */
int64_t gs_project_bwd_scratch_bytes(int64_t n);
int gs_project_bwd(int64_t n, int64_t v, const float* position, const float* log_scaling, const float* rotation,
                   const float* alpha_logit, const float* T_camera_world, const float* projection, int32_t width,
                   int32_t height, const GsRasterConfig* cfg, const int32_t* slot_of, const float* grad_points,
                   int32_t grad_points_stride, const float* grad_depth, const float* grad_depth_sq,
                   int32_t grad_depth_stride, float* d_position, float* d_log_scaling, float* d_rotation,
                   float* d_alpha_logit, float* d_T_camera_world, float* d_projection, void* scratch,
                   int64_t scratch_bytes, void* stream);

/* replaces: CameraParams.camera_position (perspective/params.py:76-78, torch.inverse(T)[0:3,3]) for an
 * affine camera matrix (last row 0 0 0 1), computed on the device without a host round trip. */
int gs_camera_position(const float* T_camera_world, float* camera_pos, void* stream);

/* ------------------------------------------------------------ spherical harmonics (a4) --
 * replaces: spherical_harmonics.py:118-134 evaluate_sh_at_kernel (+ .grad, :154-161).
 * params (n,C,D) D=(degree+1)^2, degree 0..3, C <= GS_MAX_SH_CHANNELS; positions (n,3);
 * indexes (v) int64 (may repeat); camera_pos 3 floats (device); out (v,C), row stride out_stride
 * (<= 0 means C).  v_dev (optional, device int32): the live row count when v is only a capacity.
 * Backward: d_params (n,C,D) and d_positions (n,3) are zero-filled inside, then written.  When
 * indexes_unique != 0 (the list came from gs_project_fwd) rows are written with plain stores;
 * otherwise with float atomics (the reference test passes repeated indexes,
 * tests/test_spherical_harmonics.py:27).  d_camera_pos 3 floats.  d_positions / d_camera_pos may
 * be NULL.  If slot_of (n int32: row of Gaussian i in the index list, or -1; from gs_project_fwd)
 * is given, the dense gradient is written in one pass over all n Gaussians (no memset, no atomics) and
 * grad_out may be strided: row i of the upstream gradient is grad_out + i*grad_out_stride
 * (grad_out_stride <= 0 means `channels`).  fwd_out (optional, same striding rule) is the forward's
 * output: when given and neither d_positions nor d_camera_pos is requested, the clamp mask is taken
 * from it (0 < out < 1) and the coefficients are not re-read.
 */
int gs_sh_fwd(int64_t v, const int32_t* v_dev, int32_t channels, int32_t degree, const float* params,
              const float* positions, const int64_t* indexes, const float* camera_pos, float* out,
              int32_t out_stride, void* stream);
/* Sharded frame (no reference counterpart): gs_sh_fwd for the rows whose splat can reach a tile row of
 * `shard` (points2d (v,7) from gs_project_fwd; a superset of what gs_map_prepare lists for that shard); the other rows
 * of `out` receive 0.5 ("not clamped").  shard == NULL evaluates the rows of the whole image's splats. */
/* Sharded frame, the colours of a LIST of rows (the mapper's touched list, gs_map_touched_list): row ids rows[0 ..
 * *rows_count) (both on the device; v = capacity of the list), out row = the listed row, nothing else is written. */
int gs_sh_fwd_rows(int64_t v, const int32_t* rows, const int32_t* rows_count, int32_t channels, int32_t degree,
                   const float* params, const float* positions, const int64_t* indexes, const float* camera_pos,
                   float* out, int32_t out_stride, void* stream);
int gs_sh_fwd_shard(int64_t v, const int32_t* v_dev, int32_t channels, int32_t degree, const float* params,
                    const float* positions, const int64_t* indexes, const float* camera_pos, const float* points2d,
                    int32_t height, const GsRasterConfig* cfg, const GsRowShard* shard, float* out,
                    int32_t out_stride, void* stream);
/* Sharded frame: split gs_raster_bwd's gradient rows (v, gs_grad_row_floats(F)) into the two packed arrays the ranks
 * sum: splat_out (v, 7 + colour_col0) = columns [0, 7 + colour_col0) and colour_out (v, F - colour_col0) = the feature
 * columns from colour_col0 on.  features (optional, (v,F): the forward's SH colours of THIS rank) zeroes the colour
 * gradient of clamped channels before the sum -- required with gs_sh_fwd_shard, whose other rows are 0.5. */
int gs_shard_pack_grads(int64_t v, int32_t num_features, int32_t colour_col0, const float* grad_rows,
                        const float* features, float* colour_out, float* splat_out, void* stream);
/* Sharded frame, sparse exchange: a rank's partial gradients cover the splats that can reach its rows only (7/8 of
 * the rows an 8-rank all-reduce sums are zeros on every rank).  gs_shard_pack_sparse writes one entry of 8 + F words
 * per touched splat: [row id (int32 bits), d splat (7), d features (F)] with the clamp mask of gs_shard_pack_grads
 * applied (features (v,F) optional); `touched` (m int32 rows of the gradient-row buffer) is the mapper's list
 * (gs_map_touched_offset) or any list of distinct rows.  gs_shard_add_sparse adds m entries into the two packed
 * arrays gs_shard_pack_grads would have filled, splat_out (v, 7 + colour_col0) and colour_out (v, F - colour_col0):
 * plain read-modify-write, the rows of ONE call are distinct, so calling it once per source rank in rank order
 * gives every rank the same sums bit for bit. */
int gs_shard_pack_sparse(int64_t m, const int32_t* touched, int32_t num_features, int32_t colour_col0,
                         const float* grad_rows, const float* features, float* entries, void* stream);
int gs_shard_add_sparse(int64_t m, const float* entries, int32_t num_features, int32_t colour_col0, int64_t v,
                        float* colour_out, float* splat_out, void* stream);
/* All `world` lists at once, for lists whose row ids ASCEND (gs_map_touched_list): entries_host / counts_host are HOST
 * arrays of `world` device pointers / entry counts (padding beyond a list's count is not read).  One pass: every row of
 * colour_out / splat_out is WRITTEN (zeros where no list has it -- the buffers need not be cleared), the lists are
 * summed in rank order inside each 256-row tile, so every rank gets the same bits, as with gs_shard_add_sparse list by
 * list.  row_range (optional, device int32[2], from gs_map_touched_list's owned_rows): only the 256-row tiles that
 * meet [first, last) are written (sharded gradients: nothing else is read).  tmp: 4 * world * (ceil(v / 256) + 1) bytes. */
int gs_shard_merge_sparse(int32_t world, const float* const* entries_host, const int64_t* counts_host,
                          int32_t num_features, int32_t colour_col0, int64_t v, float* colour_out, float* splat_out,
                          const int32_t* row_range, void* tmp, int64_t tmp_bytes, void* stream);
int gs_sh_bwd(int64_t n, int64_t v, int32_t channels, int32_t degree, const float* params, const float* positions,
              const int64_t* indexes, int32_t indexes_unique, const int32_t* slot_of, const float* camera_pos,
              const float* grad_out, int32_t grad_out_stride, const float* fwd_out, int32_t fwd_out_stride,
              float* d_params, float* d_positions, float* d_camera_pos, void* stream);

/* --------------------------------------------------------------- tile mapper (a5 - a10) --
 * Fused path (what map_to_tiles runs).  replaces mapper/tile_mapper.py:169-196 as a whole:
 *   gs_map_prepare: OBB tile query per Gaussian (taichi_lib/grid_query.py:10-91) -> per-tile
 *     histogram -> exclusive scan -> tile_ranges (T,2) (empty tiles (0,0), :186) and
 *     counts_out[0] = K, counts_out[1] = largest tile population (device int32[4], see below).
 *   gs_map_finish: re-runs the query, buckets (depth key, index) pairs by tile, then sorts each
 *     tile's bucket on the composite (depth bits, Gaussian index): exactly the order of the
 *     reference's stable 48-bit radix sort of (tile<<32 | depth bits) in generation order
 *     (:34-40, :113-155).  overlap_to_point (K) int32; sorted_keys (K) uint64, optional (NULL),
 *     receives the reference's key layout for verification.
 * image size is the UNPADDED (width,height); padding to a tile multiple is internal (:18-22).
 * depth: (v) f32, non-negative (ndc depth).  use_depth16: key layout of :53-59.
 * scratch: gs_map_scratch_bytes(v, num_tiles), the SAME buffer for both calls (prepare leaves the
 * region ordering in it); pair_scratch: K * 8 bytes.
 * Asynchronous use (no host read-back between the calls): v may be a capacity with the live count in
 * v_dev (device int32); k_capacity > 0 bounds the pair / overlap buffers -- tiles that would run
 * past it are dropped and counts_out[2] is set to 1 so that the caller can retry with more room;
 * counts_out is int32[4] = {K, fullest tile, overflow flag, heavy tiles (see gs_raster_fwd; 0 without tile_order)};
 * max_tile_count <= 0 in
 * gs_map_finish means "not read back": its magnitude is only a hint for sizing the per-tile sort
 * (0 = no hint); fuller tiles are still sorted.  tile_order (optional, T int32) receives the tiles by
 * descending population: a launch order for gs_raster_fwd / gs_raster_bwd (heaviest tiles first).
 * counts_host (optional): device-accessible pinned HOST memory, int32[6]; the scan kernel stores the four
 * counts_out words, then *v_dev (0 without v_dev) and then M, the number of "touched" Gaussians (those whose
 * candidate tile span reaches an owned tile row: the whole list the mapper works on), there as well, so a caller that
 * has to size buffers from K waits for an event recorded behind gs_map_prepare and reads them -- no device-to-host
 * copy launch.  The touched list itself -- M int32 rows of `points`, grouped by screen region -- is left in `scratch` at
 * byte offset gs_map_touched_offset(v, num_tiles): a sharded frame exchanges the gradient rows of exactly these
 * splats (gs_shard_pack_sparse).
 * shard (optional, host pointer, read during the call): only the owned tile rows are mapped; num_tiles in
 * gs_map_scratch_bytes and the T of tile_ranges / tile_order are then the LOCAL tile count, sorted_keys carry local
 * tile ids.  Tile decisions are computed in full-image coordinates: the tiles of a shard get exactly the lists the
 * unsharded call gives them.
 */
int64_t gs_map_scratch_bytes(int64_t v, int64_t num_tiles);
int64_t gs_map_touched_offset(int64_t v, int64_t num_tiles);
/* The touched rows in ASCENDING order (the list at gs_map_touched_offset is grouped by screen region): read from the
 * scratch gs_map_prepare has filled, written to touched_out (room for v int32; M of them are written, M = counts_host[5]).
 * Ascending rows make the gather of gs_shard_pack_sparse and the read-modify-write of gs_shard_add_sparse walk memory
 * forwards.  With owner_counts (world int64, optional): rank r owns the Gaussians [r chunk, (r + 1) chunk), chunk =
 * ceil(n / world), and since rows ascend with the Gaussian index (indexes (v) int64 from gs_project_fwd) the rows of one
 * owner are contiguous in the list -- owner_counts[r] = how many belong to owner r: the send counts of the all-to-all of
 * grad_mode "sharded".  count_out (optional, device int32) receives M.  owned_rows (optional, device int32[2], with
 * owner_counts): the row range [first, last) of the visible list whose Gaussians rank `owner` owns. */
int gs_map_touched_list(int64_t v, const int32_t* v_dev, int64_t num_tiles, const void* scratch, int64_t scratch_bytes,
                        int32_t* touched_out, int32_t* count_out, const int64_t* indexes, int64_t n, int32_t world,
                        int64_t* owner_counts, int32_t owner, int32_t* owned_rows, void* stream);
int gs_map_prepare(int64_t v, const int32_t* v_dev, const float* points, int32_t width, int32_t height,
                   const GsRasterConfig* cfg, int64_t k_capacity, int32_t* tile_ranges, int32_t* counts_out,
                   int32_t* counts_host, int32_t* tile_order, const GsRowShard* shard, void* scratch,
                   int64_t scratch_bytes, void* stream);
int gs_map_finish(int64_t v, const int32_t* v_dev, int64_t k, int32_t max_tile_count, const float* points,
                  const float* depth, int32_t width, int32_t height, const GsRasterConfig* cfg, int32_t use_depth16,
                  const int32_t* tile_ranges, int32_t* overlap_to_point, uint64_t* sorted_keys, void* pair_scratch,
                  const GsRowShard* shard, void* scratch, int64_t scratch_bytes, void* stream);

/* Diagnostic: the device forms of the two functions of include/gs_detmath.h the mapper's integer decisions rest on
 * (correctly rounded sqrt, deterministic ln), element-wise, so that a test can hold them bit for bit against the
 * host forms the CPU oracle is built from.  Either output may be NULL. */
int gs_selftest_detmath(int64_t n, const float* x, float* sqrt_out, float* log_out, void* stream);

/* Reference-shaped primitives (the same pipeline stage by stage, as the reference runs it). */

/* replaces: mapper/tile_mapper.py:74-84 tile_overlaps_kernel.  counts (v) int32. */
int gs_tile_count(int64_t v, const float* points, int32_t width, int32_t height, const GsRasterConfig* cfg,
                  int32_t* counts, void* stream);
/* replaces: cuda_lib.full_cumsum (cuda_lib/__init__.py:16-25, full_cumsum.cu:17-67).
 * out has n+1 entries; out[n] is the total (read it back from there).  scratch: gs_cumsum_scratch_bytes(n). */
int64_t gs_cumsum_scratch_bytes(int64_t n);
int gs_full_cumsum_i32(int64_t n, const int32_t* in, int32_t* out, void* scratch, int64_t scratch_bytes,
                       void* stream);
/* replaces: mapper/tile_mapper.py:113-144 generate_sort_keys_kernel.  keys (K) uint64 (the u32
 * depth16 key is zero-extended), values (K) int32, in generation order. */
int gs_tile_emit_keys(int64_t v, const float* points, const float* depth, const int32_t* offsets, int32_t width,
                      int32_t height, const GsRasterConfig* cfg, int32_t use_depth16, uint64_t* keys,
                      int32_t* values, void* stream);
/* replaces: cuda_lib.radix_sort_pairs (cuda_lib/__init__.py:28-35, radix_sort_pairs.cu:8-70):
 * stable ascending LSD radix sort of (key,value) pairs on key bits [begin_bit,end_bit)
 * (end_bit <= 0 means all bits).  key_bytes 4 or 8.  Outputs in keys_out / values_out; the inputs
 * are preserved.  scratch: gs_sort_scratch_bytes(k, key_bytes). */
int64_t gs_sort_scratch_bytes(int64_t k, int32_t key_bytes);
int gs_radix_sort_pairs(int64_t k, int32_t key_bytes, const void* keys_in, const int32_t* values_in, void* keys_out,
                        int32_t* values_out, int32_t begin_bit, int32_t end_bit, void* scratch,
                        int64_t scratch_bytes, void* stream);
/* replaces: cuda_lib.segmented_sort_pairs (cuda_lib/segmented_sort_pairs.cu:8-78; exported by the reference's
 * native module, not called on the render path): ascending sort of the pairs inside each segment
 * [start_offsets[s], end_offsets[s]) (int64, device); key_bytes 2 or 4 (signed keys), int32 values; stable.
 * Positions outside every segment are not written.  scratch: 8 bytes per item. */
int gs_segmented_sort_pairs(int64_t num_items, int32_t key_bytes, const void* keys, const int32_t* values,
                            void* keys_out, int32_t* values_out, int64_t num_segments, const int64_t* start_offsets,
                            const int64_t* end_offsets, void* scratch, int64_t scratch_bytes, void* stream);
/* replaces: mapper/tile_mapper.py:91-110 find_ranges_kernel (+ zero-init :186). */
int gs_find_ranges(int64_t k, const uint64_t* sorted_keys, int32_t use_depth16, int64_t num_tiles,
                   int32_t* tile_ranges, void* stream);

/* ------------------------------------------------------------- rasterizer (a11 - a13) --
 * replaces: rasterizer/forward.py:25-137 _forward_kernel (allocation contract of
 * rasterizer/function.py:43-76).  points (V,7), features (V,F), tile_ranges (T,2) int32 with
 * T = ceil(W/ts)*ceil(H/ts), overlap_to_point (K) int32.  image (H,W,F), alpha (H,W).
 * visibility (V) must be zero-filled by the caller when cfg->compute_visibility, else may be NULL.
 * tile_order (optional, T int32, a permutation of the tile ids, from gs_map_prepare): launch order;
 * NULL = XCD-contiguous bands.  heavy_tiles (optional, device int32, from gs_map_prepare's counts_out[3]): the
 * first *heavy_tiles tiles of the order are rasterized by four workgroups each (one per 8x8 quadrant; tile_size
 * 16 only, ignored otherwise) so that no single wave walks a very full tile alone.  Results depend on neither.
 * shard (optional, see GsRowShard): tile ids are local, image / alpha hold the owned pixel rows only.
 */
int gs_raster_fwd(int64_t v, int32_t num_features, const float* points, const float* features,
                  const int32_t* tile_ranges, const int32_t* overlap_to_point, int64_t k, int32_t width,
                  int32_t height, const GsRasterConfig* cfg, const int32_t* tile_order, const int32_t* heavy_tiles,
                  float* image, float* alpha, float* visibility, const GsRowShard* shard, void* stream);

/* replaces: rasterizer/backward.py:53-228 _backward_kernel.
 * Per-Gaussian gradients are accumulated with float atomics into ONE row per Gaussian,
 * grad_rows (V, gs_grad_row_floats(F)), 64-byte aligned rows:
 *   [0..7) d(mean.xy, axis.xy, sigma.xy, alpha), [7..7+F) d(features), [7+F, 8+F) point heuristics.
 * (One 64-B segment per (tile, splat) is the shape the memory-side float atomics run fastest at.)
 * The caller zero-fills grad_rows.  gs_raster_bwd_unpack splits rows into the reference's
 * separate tensors (rasterizer/function.py:84-91); any output may be NULL.
 */
int32_t gs_grad_row_floats(int32_t num_features);
int gs_raster_bwd(int64_t v, int32_t num_features, const float* points, const float* features,
                  const int32_t* tile_ranges, const int32_t* overlap_to_point, int64_t k, int32_t width,
                  int32_t height, const GsRasterConfig* cfg, const int32_t* tile_order, const int32_t* heavy_tiles,
                  const float* image, const float* grad_image, float* grad_rows, const GsRowShard* shard,
                  void* stream);
int gs_raster_bwd_unpack(int64_t v, int32_t num_features, const float* grad_rows, float* grad_points,
                         float* grad_features, float* point_heuristic, void* stream);

/* ------------------------------------------------- plain features (render_gaussians(use_sh=False)) --
 * replaces: `features = gaussians.feature[indexes]` (renderer.py:166) and its index backward, inside the fused frame:
 * gather rows of the (N, C) features into the rasterizer's feature buffer (row stride out_stride, live count
 * optionally in v_dev), and the dense adjoint d_features (N, C): row i = gradient row slot_of[i] (stride
 * grad_out_stride) or zeros for a culled Gaussian.
 */
int gs_feature_gather_fwd(int64_t v, const int32_t* v_dev, int32_t channels, const float* features,
                          const int64_t* indexes, float* out, int32_t out_stride, void* stream);
int gs_feature_gather_bwd(int64_t n, int32_t channels, const int32_t* slot_of, const float* grad_out,
                          int32_t grad_out_stride, float* d_features, void* stream);

/* ------------------------------------------------------- depth / depth-variance epilogue --
 * replaces: renderer.py:174-180 compute_depth_variance (+ the feature slice at :213-215) for
 * render_depth=True.  image (P, 2+C) rasterized [z, z^2, features], alpha (P): depth = I0/(alpha+eps),
 * depth_var = I1/(alpha+eps) - depth^2, features = I[2:].  Backward assembles grad_image (P, 2+C) from
 * the three upstream gradients (any may be NULL = zero); alpha is non-differentiable.
 */
int gs_depth_split_fwd(int64_t pixels, int32_t channels, const float* image, const float* alpha, float eps,
                       float* features, float* depth, float* depth_var, void* stream);
int gs_depth_split_bwd(int64_t pixels, int32_t channels, const float* depth, const float* alpha, float eps,
                       const float* grad_features, const float* grad_depth, const float* grad_depth_var,
                       float* grad_image, void* stream);

/* --------------------------------------------------- one call per direction --
 * replaces: renderer.py:134-231 render_gaussians as a whole -- project_to_image (:154-157), evaluate_sh_at / the
 * feature gather (:160-166), map_to_tiles + rasterize_with_tiles (render_projected, :183-231), the depth / variance
 * epilogue (:174-180) and the median-depth pass (:203-208) -- and its autograd backward.
 *
 * gs_frame_fwd and gs_frame_bwd enqueue the entry points above, in the order and with the arguments of the
 * stage-by-stage composition (results are bit-identical), from ONE host call each, into ONE caller-provided workspace:
 * gs_frame_layout gives the byte offset of every sub-buffer (-1 = not present for this frame) so that the caller can
 * view them as tensors.  No allocation and no synchronisation inside.
 *   GsFrame (host struct): the frame.  sh_degree -1 = plain (n, channels) features (use_sh=False).  k_capacity (>= 1):
 *     how many (tile, splat) overlaps the workspace holds; the mapper clamps to it and raises the overflow flag
 *     (counts word [6]), in which case the caller runs the frame again with more room.  max_tile_hint: expected
 *     population of the fullest tile (sizes the per-tile sort launches; 0 = none; fuller tiles are still sorted).
 *     prepare_backward: the forward also zero-fills the gradient rows gs_frame_bwd accumulates into (inside the
 *     projection's compaction pass: no fill launch); without it gs_frame_bwd clears its own rows first.
 *   workspace (gs_frame_layout().workspace_bytes): outputs and everything the backward reads.  `counts` is int32[8]:
 *     [0] = V, [4] = K, [5] = fullest tile, [6] = overflow flag, [7] = heavy tiles.  Per-Gaussian buffers are sized
 *     for n rows; V rows are live.  scratch: fwd_scratch_bytes / bwd_scratch_bytes, dead when the call's work has run.
 *   counts_host (optional, pinned host int32[5]) as in gs_map_prepare: {K, fullest tile, overflow, heavy, V}.
 *   counts_event (optional, a hipEvent_t): recorded on `stream` right behind the mapper's scan, i.e. when counts_host is
 *     final -- the caller waits on it for V and K while the sort and the rasterizer are still running.
 *   gs_frame_bwd: v, k = the counts read back; grad_image (H, W, C) and, with render_depth, grad_img_depth /
 *     grad_img_var (H, W) -- any may be NULL (zero); attached_points (v, 7) / attached_depth (v): gradients the caller
 *     attached to the projected splats / depths themselves (optional).  Outputs: dense parameter gradients
 *     d_position (n,3), d_log_scaling (n,3), d_rotation (n,4), d_alpha_logit (n,1), d_feature (n,C[,D]);
 *     d_T_camera_world (16) / d_projection (4) optional; d_camera_centre (3, optional, ZEROED BY THE CALLER) receives
 *     the SH view direction's gradient with respect to the camera centre.  Not for sharded frames (their backward
 *     exchanges partial gradients between the rasterizer and the per-Gaussian adjoints: run the stages).
 */
typedef struct GsFrame {
  int64_t n;
  int32_t channels, sh_degree;
  int32_t width, height;          /* the FULL image, also under a shard */
  double near_plane, far_plane;
  int32_t render_depth, use_depth16, render_median_depth, prepare_backward;
  int64_t k_capacity;
  int32_t max_tile_hint;
  int32_t has_shard;
  GsRowShard shard;
  GsRasterConfig cfg;
  /* render_depth: cfg.forward_cut for the rasterizer call that blends [z, z^2, features] -- the caller divides its
   * forward_cut by far^2 (GsRasterConfig.forward_cut above); the mapper and the median-depth pass use cfg as it is */
  float depth_forward_cut;
  /* Sharded frame with a sparse exchange (has_shard; 0 = none): the number of ranks.  The forward then also leaves the
   * ascending list of the touched rows in the workspace (layout.touched; count in counts[1]; per-owner counts in
   * layout.owner_counts; the row range of the Gaussians rank exchange_rank owns in counts[2..3]) and evaluates SH
   * colours for exactly those rows (gs_sh_fwd_rows; every other colour is 0.5) instead of testing every row against
   * the rank's rows (gs_sh_fwd_shard). */
  int32_t exchange_world, exchange_rank;
} GsFrame;

typedef struct GsFrameLayout {
  int64_t workspace_bytes, fwd_scratch_bytes, bwd_scratch_bytes, stage_bytes;
  /* workspace */
  int64_t counts, camera_pos, points, depth, features, indexes, slot_of, tile_ranges, tile_order, overlap_to_point,
      image, alpha, visibility, out_image, img_depth, img_var, median, grad_rows, touched, owner_counts;
  /* forward scratch */
  int64_t s_ndc_depth, s_pairs, s_median_cover, s_stage;
  /* backward scratch */
  int64_t b_grad_image, b_camera, b_grad_rows;
  int32_t num_features, grad_row_floats, tiles_x, tiles_y, local_height;
} GsFrameLayout;

/* Optional fork of gs_frame_fwd (host struct of caller-owned handles, NULL = everything on `stream`): the colour stage
 * (SH evaluation / feature gather) is enqueued on side_stream (a hipStream_t) between fork_event and join_event (two
 * hipEvent_t), i.e. underneath the tile mapper, and the rasterizer waits for it.  Everything is ordered behind `stream`
 * again when the call's work has run: buffers need no other synchronisation than for the unforked call. */
typedef struct GsFrameFork {
  void* side_stream;
  void* fork_event;
  void* join_event;
} GsFrameFork;

/* Optional per-stage timing of the frame calls: stage_events is a HOST array of 2 * GS_FWD_STAGES (gs_frame_fwd) or
 * 2 * GS_BWD_STAGES (gs_frame_bwd) hipEvent_t handles; entry 2 k is recorded in front of stage k and 2 k + 1 behind it, on
 * the stream the stage runs on; NULL entries (or a NULL array) are skipped.  This is how a caller measures one kernel's
 * launch time with HIP events on the launch stream although the whole direction is one call (bench.py's roofline). */
enum { GS_FWD_PROJECT = 0, GS_FWD_COLOURS, GS_FWD_MAP_PREPARE, GS_FWD_MAP_FINISH, GS_FWD_RASTER, GS_FWD_STAGES };
enum { GS_BWD_RASTER = 0, GS_BWD_COLOURS, GS_BWD_PROJECT, GS_BWD_STAGES };

int gs_frame_layout(const GsFrame* frame, GsFrameLayout* layout);
int gs_frame_fwd(const GsFrame* frame, const float* position, const float* log_scaling, const float* rotation,
                 const float* alpha_logit, const float* feature, const float* T_camera_world, const float* projection,
                 void* workspace, int64_t workspace_bytes, void* scratch, int64_t scratch_bytes, int32_t* counts_host,
                 void* counts_event, const GsFrameFork* fork, void* const* stage_events, void* stream);
int gs_frame_bwd(const GsFrame* frame, const float* position, const float* log_scaling, const float* rotation,
                 const float* alpha_logit, const float* feature, const float* T_camera_world, const float* projection,
                 void* workspace, int64_t workspace_bytes, void* scratch, int64_t scratch_bytes, int64_t v, int64_t k,
                 const float* grad_image, const float* grad_img_depth, const float* grad_img_var,
                 const float* attached_points, const float* attached_depth, float* d_position, float* d_log_scaling,
                 float* d_rotation, float* d_alpha_logit, float* d_feature, float* d_T_camera_world,
                 float* d_projection, float* d_camera_centre, void* const* stage_events, void* stream);

/* ------------------------------------------------------------------- Morton ordering --
 * replaces: misc/morton_sort.py:78-88 code_points64_kernel (Grid.morton_code64, :37-66).  points (n,3);
 * lower_host: 3 floats on the HOST (the grid origin, a min-reduction the caller already has);
 * cell = clamp((p - lower)/inc, 0, size-1), size <= 2^21; codes (n) uint64.  Sorting the codes with
 * gs_radix_sort_pairs gives misc/morton_sort.py:121-126 argsort.
 */
int gs_morton_codes64(int64_t n, const float* points, const float* lower_host, float inc, int32_t size,
                      uint64_t* codes, void* stream);

/* ------------------------------------------------ sparse / fractional optimizer step (8f-3) --
 * replaces: optim/fractional_adam.py:7-85 and optim/fractional_laprop.py scalar_kernel / vector_kernel.
 * For each visible row i (idx = indexes[i], int64) with fractional weight weight[i]: update the running
 * moments m, v of row idx in place and write lr_step (rows, dims).  laprop: 0 = Adam, 1 = LaProp;
 * vector_group: 0 = per-element second moment v (N,dims), 1 = one v per row (N) from |g|^2.
 * m (N,dims), total_weight (N), grad (N,dims).
 * Optional fusions (each may be NULL): row_scale (rows) multiplies the gradient of visible row i
 * (optim/visibility_aware.py:100-103); param (N,dims) applies the step in place,
 * param[idx] -= lr_step * (1 - exp(-2 weight)) * mask_lr[j] * point_lr[idx] (optim/fractional.py:31-32, :57-63,
 * :139-146; mask_lr (dims), point_lr (N) optional), in which case lr_step itself may be NULL.
 */
/* replaces: optim/visibility_aware.py:24-31 (update_visibility) + :93-103 of VisibilityOptimizer.step for the visible
 * rows (indexes unique): running_vis[idx] <- (v^4 + (running^4 - v^4) vis_beta)^(1/4); weight[i] = v / max(running, 1e-12);
 * total_weight[idx] += weight[i]; row_scale[i] = grad_scale / (v + vis_smooth)  (the latter two feed gs_optim_step). */
int gs_optim_visibility_weights(int64_t rows, const int64_t* indexes, const float* visibility, float* running_vis,
                                float* total_weight, float vis_beta, float grad_scale, float vis_smooth, float* weight,
                                float* row_scale, void* stream);
int gs_optim_step(int32_t laprop, int32_t vector_group, int64_t rows, int32_t dims, const int64_t* indexes,
                  const float* weight, float* m, float* v, const float* total_weight, const float* grad, float lr,
                  float beta1, float beta2, float eps, int32_t bias_correction, float* lr_step,
                  const float* row_scale, float* param, const float* mask_lr, const float* point_lr, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_HIP_H */
