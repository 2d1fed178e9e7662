// elementwise.hip -- per-pixel epilogue of render_depth=True (reference renderer.py:174-180, 213-215):
//   depth = I0 / (W + eps),  depth_var = I1 / (W + eps) - depth^2,  features = I[2:]
// where I is the rasterized (H,W,2+C) image of [z, z^2, features] and W the accumulated alpha.
// One pass each way instead of the ~25 small torch kernels (slices, divisions, their backward
// zero-fills and adds) the composed form costs.  HBM-bound: 4(F+1)P read, 4(C+2)P written.

#include "gs_common.h"

namespace {

__global__ __launch_bounds__(256) void depth_split_fwd_kernel(int64_t pixels, int C, const float* image,
                                                              const float* alpha, float eps, float* feat,
                                                              float* depth, float* var) {
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= pixels) return;
  const int F = C + 2;
  const float* px = image + i * F;
  const float w = alpha[i] + eps;
  const float d = px[0] / w;
  depth[i] = d;
  var[i] = px[1] / w - d * d;
  for (int c = 0; c < C; ++c) feat[i * C + c] = px[2 + c];
}

// grad_image[...,0] = (g_depth - 2 depth g_var) / w ; [...,1] = g_var / w ; [...,2:] = g_feat
// (image_weight is non-differentiable, rasterizer/function.py:72).  NULL upstream gradients are zeros.
__global__ __launch_bounds__(256) void depth_split_bwd_kernel(int64_t pixels, int C, const float* depth,
                                                              const float* alpha, float eps, const float* g_feat,
                                                              const float* g_depth, const float* g_var,
                                                              float* grad_image) {
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= pixels) return;
  const int F = C + 2;
  const float w = alpha[i] + eps;
  const float gd = g_depth ? g_depth[i] : 0.0f, gv = g_var ? g_var[i] : 0.0f;
  float* out = grad_image + i * F;
  out[0] = (gd - 2.0f * depth[i] * gv) / w;
  out[1] = gv / w;
  for (int c = 0; c < C; ++c) out[2 + c] = g_feat ? g_feat[i * C + c] : 0.0f;
}

// Plain per-Gaussian features (render_gaussians(use_sh=False): `features = gaussians.feature[indexes]`,
// reference renderer.py:166) gathered into the rasterizer's feature rows, and the dense adjoint: row i of the
// (N, C) gradient is the rasterizer's gradient row of its slot, or zero when Gaussian i was culled.
__global__ __launch_bounds__(256) void feature_gather_kernel(int64_t v, const int* v_dev, int C, const float* features,
                                                             const int64_t* indexes, float* out, int out_stride) {
  const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
  const int64_t live = v_dev ? min(int64_t(*v_dev), v) : v;
  const int64_t row = e / C;
  if (row >= live) return;
  const int c = int(e - row * C);
  out[row * out_stride + c] = features[indexes[row] * C + c];
}

__global__ __launch_bounds__(256) void feature_scatter_kernel(int64_t n, int C, const int* slot_of, const float* grad,
                                                              int grad_stride, float* d_features) {
  const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
  const int64_t i = e / C;
  if (i >= n) return;
  const int c = int(e - i * C);
  const int slot = slot_of[i];
  __builtin_nontemporal_store(slot >= 0 ? grad[int64_t(slot) * grad_stride + c] : 0.0f, d_features + e);
}

// Exchange buffers of a sharded frame: the rasterizer's gradient rows (V, row_floats) split into the two packed arrays
// the ranks sum -- splat columns [0, 7 + col0) and colour columns [7 + col0, 7 + F) -- one element per thread, so both
// sides are coalesced.  With `features` (the forward's SH colours, (V, F)) the colour gradient of a channel the forward
// clamped is zeroed HERE, before the sum: only the ranks that rasterized a splat evaluated its colour
// (gs_sh_fwd_shard), and every rank must end up with the same masked total.
__global__ __launch_bounds__(256) void shard_pack_kernel(int64_t v, int F, int col0, const float* rows, int row_floats,
                                                         const float* features, float* colour, float* splat) {
  const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
  const int width = 7 + F;
  const int64_t i = e / width;
  if (i >= v) return;
  const int c = int(e - i * width);
  float g = rows[i * row_floats + c];
  if (c < 7 + col0) {
    splat[i * (7 + col0) + c] = g;
  } else {
    const int k = c - 7;  // feature column
    if (features != nullptr) {
      const float o = features[i * F + k];
      if (!(o > 0.0f && o < 1.0f)) g = 0.0f;  // clamp sub-gradient (spherical_harmonics.py:118-134), as gs_sh_bwd's mask
    }
    colour[i * (F - col0) + (k - col0)] = g;
  }
}

// Sparse exchange of a sharded frame (include/gsplat_hip.h): entry e = [row id, 7 + F gradient words] of touched[e]
__global__ __launch_bounds__(256) void shard_pack_sparse_kernel(int64_t m, const int* touched, int F, int col0,
                                                                const float* rows, int row_floats,
                                                                const float* features, float* entries) {
  const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
  const int width = 8 + F;
  const int64_t i = e / width;
  if (i >= m) return;
  const int c = int(e - i * width);
  const int row = touched[i];
  if (c == 0) { entries[e] = __int_as_float(row); return; }
  float g = rows[int64_t(row) * row_floats + (c - 1)];
  const int k = c - 8;  // feature column
  if (k >= col0 && features != nullptr) {
    const float o = features[int64_t(row) * F + k];
    if (!(o > 0.0f && o < 1.0f)) g = 0.0f;
  }
  entries[e] = g;
}

// LANES (16 or 64) consecutive lanes per entry, lane = word of the entry: the loads of a wave are one contiguous run,
// the row id reaches the other lanes by a shuffle, and no thread divides anything
template <int LANES>
__global__ __launch_bounds__(256) void shard_add_sparse_kernel(int64_t m, const float* entries, int F, int col0,
                                                               int64_t v, float* colour, float* splat) {
  const int width = 8 + F;
  const int c = threadIdx.x % LANES;
  const int64_t i = int64_t(blockIdx.x) * (256 / LANES) + threadIdx.x / LANES;
  const bool live = i < m && c < width;
  const float g = live ? entries[i * width + c] : 0.0f;
  const int64_t row = __shfl(__float_as_int(g), 0, LANES);  // word 0 = row id
  if (!live || c == 0 || row < 0 || row >= v) return;       // (negative ids: padding of a fixed-size exchange buffer)
  const int k = c - 1;
  if (k < 7 + col0) splat[row * (7 + col0) + k] += g;
  else colour[row * (F - col0) + (k - 7 - col0)] += g;
}

// All lists of a sparse exchange merged in ONE pass over the dense rows (lists with ascending row ids): adding list
// after list touches every cache line of the dense buffers once per list -- a list holds every world-th row, so its
// 28-byte pieces land in every 128-byte line -- i.e. `world` read-modify-write passes over (V, 7 + F) floats.  Here a
// workgroup owns MERGE_ROWS consecutive rows: it finds its slice of every list (precomputed cuts), sums the slices into
// an LDS tile list by list in rank order (the rows of one list are distinct: plain adds, a barrier between lists --
// every rank computes the same sums bit for bit) and writes the tile: every dense row is written exactly once, zeros
// included, so the caller does not clear the buffers either.
constexpr int MERGE_ROWS = 256, MERGE_MAX_LISTS = 64;
struct MergeLists {
  const float* entries[MERGE_MAX_LISTS];
  long long count[MERGE_MAX_LISTS];
};

__global__ __launch_bounds__(256) void sparse_cuts_kernel(MergeLists lists, int world, int width, int num_blocks,
                                                          int* cuts) {
  const int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x;
  const int q = int(t / (num_blocks + 1)), b = int(t - int64_t(q) * (num_blocks + 1));
  if (q >= world) return;
  const int bound = b * MERGE_ROWS;
  const float* e = lists.entries[q];
  long long lo = 0, hi = lists.count[q];  // first entry whose row id >= bound
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (__float_as_int(e[mid * width]) < bound) lo = mid + 1; else hi = mid;
  }
  cuts[int64_t(q) * (num_blocks + 1) + b] = int(lo);
}

__global__ __launch_bounds__(256) void sparse_merge_kernel(MergeLists lists, int world, int F, int col0, int64_t v,
                                                           int num_blocks, const int* cuts, float* colour,
                                                           float* splat, const int* row_range) {
  extern __shared__ float s_rows[];  // MERGE_ROWS x (7 + F)
  const int width = 8 + F, cols = 7 + F;
  const int64_t row0 = int64_t(blockIdx.x) * MERGE_ROWS;
  // sharded gradients: only the rows of the Gaussians this rank owns are ever read -- tiles outside are not written
  if (row_range && (row0 + MERGE_ROWS <= row_range[0] || row0 >= row_range[1])) return;
  for (int e = threadIdx.x; e < MERGE_ROWS * cols; e += 256) s_rows[e] = 0.0f;
  __syncthreads();
  const int sub = threadIdx.x >> 4, c = threadIdx.x & 15;  // 16 lanes per entry, 16 entries per step
  for (int q = 0; q < world; ++q) {
    const int lo = cuts[int64_t(q) * (num_blocks + 1) + blockIdx.x], hi = cuts[int64_t(q) * (num_blocks + 1) + blockIdx.x + 1];
    const float* e = lists.entries[q];
    for (int i = lo + sub; i < hi; i += 16) {
      const int row = __float_as_int(e[int64_t(i) * width]) - int(row0);
      for (int k = c; k < cols; k += 16) s_rows[row * cols + k] += e[int64_t(i) * width + 1 + k];
    }
    __syncthreads();
  }
  const int64_t rows = v - row0 < MERGE_ROWS ? v - row0 : MERGE_ROWS;
  const int sc = 7 + col0, cc = F - col0;
  for (int e = threadIdx.x; e < rows * sc; e += 256) {
    const int r = e / sc, k = e - r * sc;
    splat[row0 * sc + e] = s_rows[r * cols + k];
  }
  for (int e = threadIdx.x; e < rows * cc; e += 256) {
    const int r = e / cc, k = e - r * cc;
    colour[row0 * cc + e] = s_rows[r * cols + sc + k];
  }
}

// gradients the caller attached to the projected splats / depths themselves, added to the rasterizer's gradient rows
__global__ __launch_bounds__(256) void rows_add_kernel(int64_t v, int row_floats, float* rows, const float* add_points,
                                                       const float* add_depth, int depth_col) {
  const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
  const int64_t i = e >> 3;
  if (i >= v) return;
  const int c = int(e & 7);
  if (c < 7) { if (add_points) rows[i * row_floats + c] += add_points[i * 7 + c]; }
  else if (add_depth) rows[i * row_floats + depth_col] += add_depth[i];
}

}  // namespace

extern "C" int gs_shard_pack_sparse(int64_t m, const int32_t* touched, int32_t num_features, int32_t colour_col0,
                                    const float* grad_rows, const float* features, float* entries, void* stream) {
  GS_REQUIRE(num_features >= 1 && num_features <= GS_MAX_FEATURES && colour_col0 >= 0 && colour_col0 < num_features,
             GS_ERR_INVALID_ARGUMENT, "gs_shard_pack_sparse: %d features, colours from column %d", num_features,
             colour_col0);
  if (m == 0) return GS_OK;
  GS_REQUIRE(touched && grad_rows && entries, GS_ERR_INVALID_ARGUMENT, "gs_shard_pack_sparse: NULL buffer");
  const int64_t total = m * (8 + num_features);
  hipLaunchKernelGGL(shard_pack_sparse_kernel, dim3(unsigned(gs_div_up(total, 256))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), m, touched, num_features, colour_col0, grad_rows,
                     gs_grad_row_floats(num_features), features, entries);
  GS_CHECK_LAUNCH("gs_shard_pack_sparse");
  return GS_OK;
}

extern "C" int gs_shard_add_sparse(int64_t m, const float* entries, int32_t num_features, int32_t colour_col0,
                                   int64_t v, float* colour_out, float* splat_out, void* stream) {
  GS_REQUIRE(num_features >= 1 && num_features <= GS_MAX_FEATURES && colour_col0 >= 0 && colour_col0 < num_features,
             GS_ERR_INVALID_ARGUMENT, "gs_shard_add_sparse: %d features, colours from column %d", num_features,
             colour_col0);
  if (m == 0 || v == 0) return GS_OK;
  GS_REQUIRE(entries && colour_out && splat_out, GS_ERR_INVALID_ARGUMENT, "gs_shard_add_sparse: NULL buffer");
  if (8 + num_features <= 16)
    hipLaunchKernelGGL(shard_add_sparse_kernel<16>, dim3(unsigned(gs_div_up(m, 16))), dim3(256), 0,
                       static_cast<hipStream_t>(stream), m, entries, num_features, colour_col0, v, colour_out, splat_out);
  else
    hipLaunchKernelGGL(shard_add_sparse_kernel<64>, dim3(unsigned(gs_div_up(m, 4))), dim3(256), 0,
                       static_cast<hipStream_t>(stream), m, entries, num_features, colour_col0, v, colour_out, splat_out);
  GS_CHECK_LAUNCH("gs_shard_add_sparse");
  return GS_OK;
}

extern "C" int gs_shard_merge_sparse(int32_t world, const float* const* entries_host, const int64_t* counts_host,
                                     int32_t num_features, int32_t colour_col0, int64_t v, float* colour_out,
                                     float* splat_out, const int32_t* row_range, void* tmp, int64_t tmp_bytes,
                                     void* stream) {
  GS_REQUIRE(num_features >= 1 && num_features <= GS_MAX_FEATURES && colour_col0 >= 0 && colour_col0 < num_features,
             GS_ERR_INVALID_ARGUMENT, "gs_shard_merge_sparse: %d features, colours from column %d", num_features,
             colour_col0);
  GS_REQUIRE(world >= 1 && world <= MERGE_MAX_LISTS && entries_host && counts_host, GS_ERR_INVALID_ARGUMENT,
             "gs_shard_merge_sparse: %d lists (1 .. %d)", world, MERGE_MAX_LISTS);
  if (v == 0) return GS_OK;
  GS_REQUIRE(colour_out && splat_out, GS_ERR_INVALID_ARGUMENT, "gs_shard_merge_sparse: NULL output");
  const int nb = int(gs_div_up(v, MERGE_ROWS));
  GS_REQUIRE(tmp && tmp_bytes >= int64_t(world) * (nb + 1) * 4, GS_ERR_SCRATCH_TOO_SMALL,
             "gs_shard_merge_sparse: tmp %lld < %lld bytes", (long long)tmp_bytes, (long long)(int64_t(world) * (nb + 1) * 4));
  MergeLists lists;
  for (int q = 0; q < world; ++q) {
    GS_REQUIRE(counts_host[q] == 0 || entries_host[q], GS_ERR_INVALID_ARGUMENT, "gs_shard_merge_sparse: list %d is NULL", q);
    lists.entries[q] = entries_host[q];
    lists.count[q] = counts_host[q];
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  int* cuts = static_cast<int*>(tmp);
  const int64_t searches = int64_t(world) * (nb + 1);
  hipLaunchKernelGGL(sparse_cuts_kernel, dim3(unsigned(gs_div_up(searches, 256))), dim3(256), 0, s, lists, world,
                     8 + num_features, nb, cuts);
  hipLaunchKernelGGL(sparse_merge_kernel, dim3(nb), dim3(256), size_t(MERGE_ROWS) * (7 + num_features) * 4, s, lists,
                     world, num_features, colour_col0, v, nb, cuts, colour_out, splat_out, row_range);
  GS_CHECK_LAUNCH("gs_shard_merge_sparse");
  return GS_OK;
}

int gs_rows_add(int64_t v, int32_t row_floats, float* rows, const float* add_points, const float* add_depth,
                int32_t depth_col, void* stream) {
  if (v == 0 || (!add_points && !add_depth)) return GS_OK;
  GS_REQUIRE(rows, GS_ERR_INVALID_ARGUMENT, "gs_rows_add: rows is NULL");
  hipLaunchKernelGGL(rows_add_kernel, dim3(unsigned(gs_div_up(v * 8, 256))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), v, row_floats, rows, add_points, add_depth, depth_col);
  GS_CHECK_LAUNCH("gs_rows_add");
  return GS_OK;
}

extern "C" int gs_shard_pack_grads(int64_t v, int32_t num_features, int32_t colour_col0, const float* grad_rows,
                                   const float* features, float* colour_out, float* splat_out, void* stream) {
  GS_REQUIRE(num_features >= 1 && num_features <= GS_MAX_FEATURES && colour_col0 >= 0 && colour_col0 < num_features,
             GS_ERR_INVALID_ARGUMENT, "gs_shard_pack_grads: %d features, colours from column %d", num_features,
             colour_col0);
  if (v == 0) return GS_OK;
  GS_REQUIRE(grad_rows && colour_out && splat_out, GS_ERR_INVALID_ARGUMENT, "gs_shard_pack_grads: NULL buffer");
  const int64_t total = v * (7 + num_features);
  hipLaunchKernelGGL(shard_pack_kernel, dim3(unsigned(gs_div_up(total, 256))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), v, num_features, colour_col0, grad_rows,
                     int(gs_grad_row_floats(num_features)), features, colour_out, splat_out);
  GS_CHECK_LAUNCH("gs_shard_pack_grads");
  return GS_OK;
}

extern "C" int gs_feature_gather_fwd(int64_t v, const int32_t* v_dev, int32_t channels, const float* features,
                                     const int64_t* indexes, float* out, int32_t out_stride, void* stream) {
  GS_REQUIRE(channels >= 1, GS_ERR_INVALID_ARGUMENT, "gs_feature_gather_fwd: %d channels", channels);
  if (v == 0) return GS_OK;
  GS_REQUIRE(features && indexes && out, GS_ERR_INVALID_ARGUMENT, "gs_feature_gather_fwd: NULL buffer");
  hipLaunchKernelGGL(feature_gather_kernel, dim3(unsigned(gs_div_up(v * channels, 256))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), v, v_dev, channels, features, indexes, out,
                     out_stride > 0 ? out_stride : channels);
  GS_CHECK_LAUNCH("gs_feature_gather_fwd");
  return GS_OK;
}

extern "C" int gs_feature_gather_bwd(int64_t n, int32_t channels, const int32_t* slot_of, const float* grad_out,
                                     int32_t grad_out_stride, float* d_features, void* stream) {
  GS_REQUIRE(channels >= 1, GS_ERR_INVALID_ARGUMENT, "gs_feature_gather_bwd: %d channels", channels);
  if (n == 0) return GS_OK;
  GS_REQUIRE(slot_of && grad_out && d_features, GS_ERR_INVALID_ARGUMENT, "gs_feature_gather_bwd: NULL buffer");
  hipLaunchKernelGGL(feature_scatter_kernel, dim3(unsigned(gs_div_up(n * channels, 256))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), n, channels, slot_of, grad_out,
                     grad_out_stride > 0 ? grad_out_stride : channels, d_features);
  GS_CHECK_LAUNCH("gs_feature_gather_bwd");
  return GS_OK;
}

extern "C" int gs_depth_split_fwd(int64_t pixels, int32_t channels, const float* image, const float* alpha,
                                  float eps, float* features, float* depth, float* depth_var, void* stream) {
  GS_REQUIRE(channels >= 1 && channels + 2 <= GS_MAX_FEATURES, GS_ERR_UNSUPPORTED, "gs_depth_split_fwd: %d channels",
             channels);
  if (pixels == 0) return GS_OK;
  GS_REQUIRE(image && alpha && features && depth && depth_var, GS_ERR_INVALID_ARGUMENT, "gs_depth_split_fwd: NULL buffer");
  hipLaunchKernelGGL(depth_split_fwd_kernel, dim3(unsigned(gs_div_up(pixels, 256))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), pixels, channels, image, alpha, eps, features, depth, depth_var);
  GS_CHECK_LAUNCH("gs_depth_split_fwd");
  return GS_OK;
}

extern "C" int gs_depth_split_bwd(int64_t pixels, int32_t channels, const float* depth, const float* alpha,
                                  float eps, const float* grad_features, const float* grad_depth,
                                  const float* grad_depth_var, float* grad_image, void* stream) {
  GS_REQUIRE(channels >= 1 && channels + 2 <= GS_MAX_FEATURES, GS_ERR_UNSUPPORTED, "gs_depth_split_bwd: %d channels",
             channels);
  if (pixels == 0) return GS_OK;
  GS_REQUIRE(depth && alpha && grad_image, GS_ERR_INVALID_ARGUMENT, "gs_depth_split_bwd: NULL buffer");
  hipLaunchKernelGGL(depth_split_bwd_kernel, dim3(unsigned(gs_div_up(pixels, 256))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), pixels, channels, depth, alpha, eps, grad_features, grad_depth,
                     grad_depth_var, grad_image);
  GS_CHECK_LAUNCH("gs_depth_split_bwd");
  return GS_OK;
}

// ---------------------------------------------------------------------------------------------
// 3D Morton codes (reference misc/morton_sort.py:10-88): cell = clamp((p - lower) / inc, 0, size-1)
// per axis, 21 bits per axis interleaved x | y<<1 | z<<2 into 63 bits.  Integer result: the f32
// subtraction and division are single correctly rounded ops, so the codes are bit-exact.
namespace {
__device__ __forceinline__ uint64_t spread_bits64(uint64_t x) {
  x &= 0x1fffffull;
  x = (x | (x << 32)) & 0x1f00000000ffffull;
  x = (x | (x << 16)) & 0x1f0000ff0000ffull;
  x = (x | (x << 8)) & 0x100f00f00f00f00full;
  x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
  x = (x | (x << 2)) & 0x1249249249249249ull;
  return x;
}

__global__ __launch_bounds__(256) void morton_kernel(int64_t n, const float* points, float lx, float ly, float lz,
                                                     float inc, int size, uint64_t* codes) {
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const float hi = float(size - 1);
  const float vx = __fdiv_rn(points[3 * i] - lx, inc), vy = __fdiv_rn(points[3 * i + 1] - ly, inc),
              vz = __fdiv_rn(points[3 * i + 2] - lz, inc);
  const uint32_t cx = uint32_t(fminf(fmaxf(vx, 0.0f), hi)), cy = uint32_t(fminf(fmaxf(vy, 0.0f), hi)),
                 cz = uint32_t(fminf(fmaxf(vz, 0.0f), hi));
  codes[i] = spread_bits64(cx) | (spread_bits64(cy) << 1) | (spread_bits64(cz) << 2);
}
}  // namespace

extern "C" int gs_morton_codes64(int64_t n, const float* points, const float* lower_host, float inc, int32_t size,
                                 uint64_t* codes, void* stream) {
  GS_REQUIRE(size > 0 && size <= (1 << 21) && inc > 0.0f, GS_ERR_INVALID_ARGUMENT, "gs_morton_codes64: grid");
  if (n == 0) return GS_OK;
  GS_REQUIRE(points && lower_host && codes, GS_ERR_INVALID_ARGUMENT, "gs_morton_codes64: NULL buffer");
  hipLaunchKernelGGL(morton_kernel, dim3(unsigned(gs_div_up(n, 256))), dim3(256), 0, static_cast<hipStream_t>(stream),
                     n, points, lower_host[0], lower_host[1], lower_host[2], inc, size, codes);
  GS_CHECK_LAUNCH("gs_morton_codes64");
  return GS_OK;
}
