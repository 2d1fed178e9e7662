// gs_common.h -- shared by every translation unit of libgsplat_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gsplat_hip.h"

#define GS_WAVE 64

void gs_set_error(const char* fmt, ...);

#define GS_REQUIRE(cond, status, ...)  \
  do {                                 \
    if (!(cond)) {                     \
      gs_set_error(__VA_ARGS__);       \
      return (status);                 \
    }                                  \
  } while (0)

#define GS_CHECK_LAUNCH(name)                                                   \
  do {                                                                          \
    hipError_t e_ = hipGetLastError();                                          \
    if (e_ != hipSuccess) {                                                     \
      gs_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
      return GS_ERR_LAUNCH;                                                     \
    }                                                                           \
  } while (0)

static inline int64_t gs_div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t gs_align_up(int64_t a, int64_t b) { return gs_div_up(a, b) * b; }

static inline int gs_check_cfg(const GsRasterConfig* cfg) {
  GS_REQUIRE(cfg != nullptr, GS_ERR_INVALID_ARGUMENT, "config is NULL");
  GS_REQUIRE(cfg->tile_size == 8 || cfg->tile_size == 16 || cfg->tile_size == 32, GS_ERR_UNSUPPORTED,
             "tile_size %d not supported (8, 16 or 32)", cfg->tile_size);
  GS_REQUIRE(cfg->alpha_threshold > 0.f, GS_ERR_INVALID_ARGUMENT, "alpha_threshold must be > 0");
  return GS_OK;
}

// ---- screen-tile sharding (GsRowShard, include/gsplat_hip.h): owned tile rows <-> local rows.  Plain struct
// arithmetic, usable on host and device.  `rows` = tile rows of the full image.
struct GsShard {
  int begin, end, band, period, phase;  // period == 1: every row of [begin, end) is owned
  int local_rows;                       // number of owned rows
};

#ifdef __HIPCC__
#define GS_HD __host__ __device__ __forceinline__
#else
#define GS_HD inline
#endif

GS_HD bool gs_shard_owns(const GsShard& s, int ty) {
  if (ty < s.begin || ty >= s.end) return false;
  return s.period == 1 || (ty / s.band) % s.period == s.phase;
}
GS_HD int gs_shard_local_row(const GsShard& s, int ty) {  // ty must be owned
  return s.period == 1 ? ty - s.begin : (ty / (s.band * s.period)) * s.band + ty % s.band;
}
GS_HD int gs_shard_global_row(const GsShard& s, int local) {
  return s.period == 1 ? local + s.begin : (local / s.band) * s.band * s.period + s.phase * s.band + local % s.band;
}

// does [lo, hi) hold a tile row of the shard?
GS_HD bool gs_shard_any_row(const GsShard& s, int lo, int hi) {
  lo = lo > s.begin ? lo : s.begin;
  hi = hi < s.end ? hi : s.end;
  if (lo >= hi) return false;
  if (s.period == 1 || hi - lo >= s.band * s.period) return true;
  for (int ty = lo; ty < hi; ++ty)
    if ((ty / s.band) % s.period == s.phase) return true;
  return false;
}

// validates `shard` against the image (NULL = whole image) and fills `out`
static inline int gs_make_shard(const GsRowShard* shard, int rows, GsShard* out) {
  if (shard == nullptr) {
    out->begin = 0; out->end = rows; out->band = rows > 0 ? rows : 1; out->period = 1; out->phase = 0;
    out->local_rows = rows;
    return GS_OK;
  }
  GS_REQUIRE(shard->row_begin >= 0 && shard->row_begin <= shard->row_end && shard->row_end <= rows,
             GS_ERR_INVALID_ARGUMENT, "shard rows [%d, %d) outside the image's %d tile rows", shard->row_begin,
             shard->row_end, rows);
  out->begin = shard->row_begin; out->end = shard->row_end;
  if (shard->period <= 1) {
    out->band = rows > 0 ? rows : 1; out->period = 1; out->phase = 0;
    out->local_rows = out->end - out->begin;
    return GS_OK;
  }
  GS_REQUIRE(shard->band >= 1 && shard->phase >= 0 && shard->phase < shard->period, GS_ERR_INVALID_ARGUMENT,
             "shard band %d period %d phase %d", shard->band, shard->period, shard->phase);
  GS_REQUIRE(shard->row_begin == 0 && shard->row_end == rows, GS_ERR_INVALID_ARGUMENT,
             "an interleaved shard covers all tile rows (got [%d, %d) of %d)", shard->row_begin, shard->row_end, rows);
  out->band = shard->band; out->period = shard->period; out->phase = shard->phase;
  const int cycle = shard->band * shard->period, full = rows / cycle, rem = rows % cycle;
  int tail = rem - shard->phase * shard->band;
  tail = tail < 0 ? 0 : (tail > shard->band ? shard->band : tail);
  out->local_rows = full * shard->band + tail;
  return GS_OK;
}

// How many 8x8 pixel sub-blocks one rasterizer wave owns: 4 (a 16x16 region, 4 pixels per lane) when the
// tile grid alone fills the chip (256 CUs x 4 SIMDs x ~8 waves); 2 or 1 for small grids (training-size
// images, a tile-row strip of a sharded frame), which then run 2x / 4x as many waves at some extra
// staging work per splat.  Results do not depend on it.
int gs_raster_sub_blocks(const GsRasterConfig* cfg, int64_t num_tiles, int backward);

// library-internal entry points behind gs_frame_fwd / gs_frame_bwd (frame.cpp)
int gs_project_fwd_ex(int64_t n, const float* position, const float* log_scaling, const float* rotation,
                      const float* alpha_logit, const float* T_camera_world, const float* projection, int32_t width,
                      int32_t height, double near_plane, double far_plane, const GsRasterConfig* cfg, float* points,
                      float* depth, float* ndc_depth, int64_t* indexes, int32_t* slot_of, int32_t* num_visible,
                      float* depth_features, int32_t depth_features_stride, float* camera_pos, void* scratch,
                      int64_t scratch_bytes, float* zero_rows, int32_t zero_row_floats, const struct GsMapBinPlan* bin,
                      void* stream);
// The frame calls fold the mapper's first pass (which screen region does a visible Gaussian's centre fall in, how many of
// each region does a workgroup hold) into the projection's compaction pass, which has every visible row in registers:
// one launch and one 28-byte read of the V rows less.  The kernel lives in mapper.hip, because what it
// decides -- is the candidate tile span empty? -- must be the counting pass's own arithmetic, compiled with the same
// flags (-ffp-contract=off).  gs_project_fwd_ex hands it the compaction's operands; gs_map_prepare_ex(binned = 1) then
// starts at the region scans.
struct GsMapBinPlan {
  int32_t width, height;
  const GsRasterConfig* cfg;
  const GsRowShard* shard;  // NULL = whole image
  void* scratch;            // the mapper scratch gs_map_prepare_ex will be given (gs_map_scratch_bytes(n, num_tiles))
  int64_t scratch_bytes;
};
struct GsCompactArgs {
  int64_t n;
  const void* st_rows;         // (n, 2) float4 staged by the projection pass; .w of the second = depth, 0 = culled
  const int* block_offsets;    // exclusive prefix of block_counts, or NULL (every workgroup sums the counts in front of it)
  const int* block_counts;     // visible rows per 256 consecutive Gaussians
  int num_blocks;              // ceil(n / 256)
  float inv_far, ndc_denom;
  float* points; float* depth; float* ndc; int64_t* indexes; int* slot_of; int* num_visible;
  float* depth_feat; int depth_feat_stride;
  void* zero_rows; int zero_row_v4;  // float4 units per row
};
int gs_map_compact_bin(const GsMapBinPlan* plan, const GsCompactArgs* c, void* stream);
int gs_map_bin_counters(const GsMapBinPlan* plan, int64_t n, int32_t** words, int32_t* count);
// experiment (GS_PROJECT_ONE_PASS): projection + cull + compaction + binning in ONE kernel with a decoupled look-back
// (mapper.hip); pa = project.hip's filled gs_proj::ProjArgs, lookback = gs_map_one_pass_scratch_bytes(n) bytes
int64_t gs_map_one_pass_scratch_bytes(int64_t n);
int gs_map_project_compact_bin(const GsMapBinPlan* plan, const void* pa, const GsCompactArgs* c, float* camera_pos,
                               void* lookback, void* stream);
int gs_map_prepare_ex(int64_t v, const int32_t* v_dev, const float* points, int32_t width, int32_t height,
                      const GsRasterConfig* cfg, int64_t k_capacity, int32_t* tile_ranges, int32_t* counts_out,
                      int32_t* counts_host, int32_t* tile_order, const GsRowShard* shard, void* scratch,
                      int64_t scratch_bytes, int binned, void* stream);
// rows[i, 0..7) += add_points[i, 0..7) and rows[i, depth_col] += add_depth[i] for i < v (either may be NULL)
int gs_rows_add(int64_t v, int32_t row_floats, float* rows, const float* add_points, const float* add_depth,
                int32_t depth_col, void* stream);

// ------------------------------------------------------------------ device helpers
#ifdef __HIPCC__

// XCD-aware block -> work-item remap: blocks b and b+8 share an XCD (round-robin dispatch), so
// giving XCD x the contiguous chunk [x*chunk, (x+1)*chunk) keeps neighbouring tiles -- which
// gather the same splat rows -- on one L2.  Speed only; any placement is correct.
// Launch with grid = 8 * ceil(n/8); returns -1 for the padding blocks.
__device__ __forceinline__ int gs_xcd_remap(int block, int n) {
  const int chunk = (n + 7) >> 3;
  const int item = (block & 7) * chunk + (block >> 3);
  return ((block >> 3) < chunk && item < n) ? item : -1;
}

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float gs_dpp_add(float x) {
  // x + dpp(x): lanes whose source is invalid or masked add 0 (old = 0, bound_ctrl = false)
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, BANK_MASK, false);
  return x + __int_as_float(moved);
}

// Sum over the 64 lanes of the wave; the total is valid in lane 63.
// row_ror 8/4/2/1 leaves every lane of a 16-lane row holding the row sum (4 fused v_add_f32_dpp),
// row_bcast:15 / row_bcast:31 then fold the rows (ISA: DPP_ROW_BCAST15 = 0x142, BCAST31 = 0x143).
__device__ __forceinline__ float gs_wave_sum_to_lane63(float x) {
  x = gs_dpp_add<0x128>(x);  // row_ror:8
  x = gs_dpp_add<0x124>(x);  // row_ror:4
  x = gs_dpp_add<0x122>(x);  // row_ror:2
  x = gs_dpp_add<0x121>(x);  // row_ror:1
  x = gs_dpp_add<0x142, 0xa>(x);  // row_bcast:15 into rows 1 and 3
  x = gs_dpp_add<0x143, 0xc>(x);  // row_bcast:31 into rows 2 and 3
  return x;
}

template <int CTRL>
__device__ __forceinline__ float gs_dpp_add_full(float x) {
  // x + dpp(x) with full row/bank masks: the form hipcc fuses into ONE v_add_f32_dpp
  return x + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xf, 0xf, true));
}

// Transposed butterfly reduction of 16 per-lane values over the 64 lanes of the wave.
// On return, lane l holds the wave total of value (l >> 2); 35 VALU instructions for 16 values,
// against 6 per value (96) for independent DPP reductions.  Each stage halves the number of lanes a
// value is spread over while packing twice as many values into a register:
//   A  v_permlane32_swap: (v[i], v[i+8]) -> lanes 0-31 carry value i, lanes 32-63 value i+8
//   B  v_permlane16_swap: rows (16 lanes) carry values i, i+4, i+8, i+12
//   C  row_ror:8 + select: half-rows     D  row_half_mirror + select: quads
//   E  quad_perm xor-1, xor-2: every lane of a quad holds the total
__device__ __forceinline__ float gs_wave_reduce16_transposed(float (&v)[16], int lane) {
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 8]), false, false);
    a[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  float b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[i]), __float_as_uint(a[i + 4]), false, false);
    b[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  const bool hi8 = (lane & 8) != 0, hi4 = (lane & 4) != 0;
  float c[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float t = gs_dpp_add_full<0x128>(b[i]);      // row_ror:8
    const float u = gs_dpp_add_full<0x128>(b[i + 2]);
    c[i] = hi8 ? u : t;
  }
  const float t = gs_dpp_add_full<0x141>(c[0]);        // row_half_mirror
  const float u = gs_dpp_add_full<0x141>(c[1]);
  float d = hi4 ? u : t;
  d = gs_dpp_add_full<0xB1>(d);                        // quad_perm:[1,0,3,2]
  d = gs_dpp_add_full<0x4E>(d);                        // quad_perm:[2,3,0,1]
  return d;
}

// The same butterfly sized for exactly N <= 16 values (no work on padding): 25 instructions for N = 9.
// After the call, lane l holds the wave total of value gs_reduce_slot<N>(l) (or garbage when that is -1).
template <int N>
struct GsReduceShape {
  static constexpr int h1 = (N + 1) / 2, h2 = (h1 + 1) / 2, h3 = (h2 + 1) / 2, h4 = (h3 + 1) / 2;
  static_assert(N >= 1 && N <= 16 && h4 == 1, "one register must remain after four halvings");
};

template <int N>
__device__ __forceinline__ int gs_reduce_slot(int lane) {
  typedef GsReduceShape<N> S;
  const int row = lane >> 4, col = lane & 15;
  const int ci = (col >> 2) & 1;                 // stage D picked c[ci]
  if (ci >= S::h3) return -1;
  const int bi = ci + ((col >> 3) & 1) * S::h3;  // stage C picked b[bi]
  if (bi >= S::h2) return -1;
  const int ai = bi + (row & 1) * S::h2;         // stage B: odd rows carry a[bi + h2]
  if (ai >= S::h1) return -1;
  const int vi = ai + (row >> 1) * S::h1;        // stage A: upper half carries v[ai + h1]
  return vi < N ? vi : -1;
}

template <int N>
__device__ __forceinline__ float gs_wave_reduce_transposed(float (&v)[N], int lane) {
  typedef GsReduceShape<N> S;
  float a[S::h1];
#pragma unroll
  for (int i = 0; i < S::h1; ++i) {
    const float partner = (i + S::h1 < N) ? v[i + S::h1 < N ? i + S::h1 : 0] : 0.0f;
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(partner), false, false);
    a[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  float b[S::h2];
#pragma unroll
  for (int i = 0; i < S::h2; ++i) {
    const float partner = (i + S::h2 < S::h1) ? a[i + S::h2 < S::h1 ? i + S::h2 : 0] : 0.0f;
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[i]), __float_as_uint(partner), false, false);
    b[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  const bool hi8 = (lane & 8) != 0, hi4 = (lane & 4) != 0;
  float c[S::h3];
#pragma unroll
  for (int i = 0; i < S::h3; ++i) {
    const float t = gs_dpp_add_full<0x128>(b[i]);  // row_ror:8
    if (i + S::h3 < S::h2) {
      const float u = gs_dpp_add_full<0x128>(b[i + S::h3 < S::h2 ? i + S::h3 : 0]);
      c[i] = hi8 ? u : t;
    } else {
      c[i] = t;
    }
  }
  float d = gs_dpp_add_full<0x141>(c[0]);            // row_half_mirror
  if (S::h3 > 1) {
    const float u = gs_dpp_add_full<0x141>(c[S::h3 > 1 ? 1 : 0]);
    d = hi4 ? u : d;
  }
  d = gs_dpp_add_full<0xB1>(d);  // quad_perm:[1,0,3,2]
  d = gs_dpp_add_full<0x4E>(d);  // quad_perm:[2,3,0,1]
  return d;
}

// Experiment (GS_BWD_BPERMUTE in raster_bwd.hip): the two swap stages of the 8-value butterfly through the LDS crossbar
// (ds_bpermute_b32: no LDS memory, but the LDS pipe) instead of v_permlane32/16_swap -- per register pair two selects,
// one exchange on the other pipe and an add: 3 VALU slots where swap + add take 4.  Measured: 0.70 against 0.61 ms
// (profiles/r3/ab_butterfly_bpermute.txt) -- six more operations per (region, splat) on the LDS pipe, which the record
// fetches and the zero reads already use, cost far more than the six VALU slots they free.  Not used.
__device__ __forceinline__ float gs_wave_reduce_transposed8_bpermute(float (&v)[8], int lane) {
  const bool up32 = (lane & 32) != 0, up16 = (lane & 16) != 0;
  const int addr32 = (lane ^ 32) << 2, addr16 = (lane ^ 16) << 2;
  float a[4], b[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float keep = up32 ? v[i + 4] : v[i], send = up32 ? v[i] : v[i + 4];
    a[i] = keep + __int_as_float(__builtin_amdgcn_ds_bpermute(addr32, __float_as_int(send)));
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float keep = up16 ? a[i + 2] : a[i], send = up16 ? a[i] : a[i + 2];
    b[i] = keep + __int_as_float(__builtin_amdgcn_ds_bpermute(addr16, __float_as_int(send)));
  }
  const float t = gs_dpp_add_full<0x128>(b[0]), u = gs_dpp_add_full<0x128>(b[1]);  // row_ror:8
  float d = gs_dpp_add_full<0x141>((lane & 8) != 0 ? u : t);                      // row_half_mirror
  d = gs_dpp_add_full<0xB1>(d);                                                   // quad_perm:[1,0,3,2]
  d = gs_dpp_add_full<0x4E>(d);                                                   // quad_perm:[2,3,0,1]
  return d;
}

// N = 9 (six moments + three colour gradients, the lean backward at F = 3): the ninth value would ride the butterfly
// alone through both swap stages (a v_permlane swap costs 3 plain instructions); six fused DPP adds take it to row 3
// instead, where lane 60 -- not an owner in the 8-value layout -- picks it up.
template <>
__device__ __forceinline__ int gs_reduce_slot<9>(int lane) {
  return lane == 60 ? 8 : gs_reduce_slot<8>(lane);
}

template <>
__device__ __forceinline__ float gs_wave_reduce_transposed<9>(float (&v)[9], int lane) {
  float w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = v[i];
  float x = v[8];
  x = gs_dpp_add_full<0x128>(x);  // row_ror:8
  x = gs_dpp_add_full<0x124>(x);  // row_ror:4
  x = gs_dpp_add_full<0x122>(x);  // row_ror:2
  x = gs_dpp_add_full<0x121>(x);  // row_ror:1 -> every lane holds its row's sum
  x = gs_dpp_add_full<0x142>(x);  // row_bcast:15: rows 1..3 add the row in front of them
  x = gs_dpp_add_full<0x143>(x);  // row_bcast:31: rows 2, 3 add (row 0 + row 1) -> row 3 holds the total
  const float d = gs_wave_reduce_transposed<8>(w, lane);
  return lane == 60 ? x : d;
}

__device__ __forceinline__ float gs_exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32
__device__ __forceinline__ float gs_rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }    // v_rcp_f32 (1 ulp)

// Which of a wave region's NB 8x8 pixel sub-blocks can a splat reach?  (Ax, Ay), (Bx, By) are the rows of the map
// d = pixel - mean  ->  t  in which alpha > alpha_threshold  <=>  |t|^2 < r2; (relx, rely) = first pixel centre of the
// region minus the mean.  Per sub-block the EXACT minimum of the quadratic |t(d)|^2 over the block's 8x8 pixel centres
// (a box): zero when the mean lies inside, otherwise attained on one of the two edges facing the mean, where it is a
// clamped 1-D parabola vertex.  Bit b is set when that minimum is below r2 (plus rounding slack) -- unlike a test of
// the ellipse's oriented bounding box this also drops the blocks that only the box's corners reach.
// Runs once per (wave, splat) on the staging lane: ~20 VALU per sub-block spread over 64 splats.
template <int NB>
__device__ __forceinline__ int gs_sub_block_mask(float Ax, float Ay, float Bx, float By, float r2, float relx,
                                                 float rely) {
  const float sxx = Ax * Ax + Bx * Bx, sxy = Ax * Ay + Bx * By, syy = Ay * Ay + By * By;
  // vertex of the parabola along an edge x = const / y = const (v_rcp_f32: the slack below covers its 1 ulp)
  const float rxy = -sxy * gs_rcp_fast(syy), ryx = -sxy * gs_rcp_fast(sxx);
  const float lim = r2 * 1.002f + 1e-3f;
  int mask = 0;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const float xa = relx + float((b & 1) * 8), ya = rely + float((b >> 1) * 8);
    const float xb = xa + 7.0f, yb = ya + 7.0f;
    const float xn = __builtin_amdgcn_fmed3f(0.0f, xa, xb), yn = __builtin_amdgcn_fmed3f(0.0f, ya, yb);
    const float ys = __builtin_amdgcn_fmed3f(rxy * xn, ya, yb);  // best y on the edge x = xn
    const float xs = __builtin_amdgcn_fmed3f(ryx * yn, xa, xb);  // best x on the edge y = yn
    // |t|^2 from the PROJECTED coordinates, as the per-pixel code forms it: the expanded quadratic
    // sxx x^2 + 2 sxy x y + syy y^2 cancels by (sigma1 / sigma2)^2 for an elongated splat far from its mean (terms
    // ~1e6 for a result ~10: an f32 error of ~0.1, far above the slack), the projections do not
    const float t1x = Ax * xn + Ay * ys, t1y = Bx * xn + By * ys;
    const float t2x = Ax * xs + Ay * yn, t2y = Bx * xs + By * yn;
    const float q1 = t1x * t1x + t1y * t1y, q2 = t2x * t2x + t2y * t2y;
    if (fminf(q1, q2) <= lim) mask |= 1 << b;
  }
  return mask;
}

// The same question for the antialiased pdf (taichi_lib/generic.py:341-357), whose support is a BOX in the splat's
// frame, not an ellipse: pdf = tau sx sy D(ux; sx) D(uy; sy) with D(x; s) = S((x + 0.5) / s) - S((x - 0.5) / s),
// S(z) = sigmoid(1.6 z + 0.07 z^3).  For |x| > 0.5:  D(x; s) <= 1 - S((|x| - 0.5) / s) <= exp(-g(z)), g(z) = 1.6 z + 0.07 z^3,
// z = (|x| - 0.5) / s; so alpha = alpha_p pdf > thr needs g(z_x) < L_x = ln(alpha_p tau sx sy D(0; sy) / thr) (and the
// same with x, y exchanged), and g(z) < L implies z < min(L / 1.6, cbrt(L / 0.07)).  Half extents X, Y of the box along
// the axis (ax, ay) and its perpendicular; separating-axis test of that oriented box against each 8x8 block of pixel
// centres (half extent 3.5).  d0x = D(0; sx), d0y = D(0; sy).  Returns 0 when the splat cannot reach the threshold.
template <int NB>
__device__ __forceinline__ int gs_sub_block_mask_antialias(float ax, float ay, float sx, float sy, float alpha,
                                                           float inv_thr, float d0x, float d0y, float relx, float rely) {
  const float base = alpha * 6.28318530717958648f * sx * sy * inv_thr;
  const float Lx = __logf(base * d0y), Ly = __logf(base * d0x);
  if (!(Lx > 0.0f && Ly > 0.0f)) return 0;
  const float zx = fminf(Lx * 0.625f, cbrtf(Lx * 14.2857142857f)), zy = fminf(Ly * 0.625f, cbrtf(Ly * 14.2857142857f));
  const float X = (0.5f + sx * zx) * 1.001f + 0.01f, Y = (0.5f + sy * zy) * 1.001f + 0.01f;
  const float h = 3.5f * (fabsf(ax) + fabsf(ay));
  const float ex = 3.5f + X * fabsf(ax) + Y * fabsf(ay), ey = 3.5f + X * fabsf(ay) + Y * fabsf(ax);
  int mask = 0;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const float cx = relx + float((b & 1) * 8) + 3.5f, cy = rely + float((b >> 1) * 8) + 3.5f;  // block centre - mean
    if (fabsf(cx * ax + cy * ay) <= X + h && fabsf(cy * ax - cx * ay) <= Y + h && fabsf(cx) <= ex && fabsf(cy) <= ey)
      mask |= 1 << b;
  }
  return mask;
}

#endif  // __HIPCC__
