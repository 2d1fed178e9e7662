// sh.hip -- view-dependent colour from real spherical harmonics, degree 0..3, and its adjoint.
// Reference: spherical_harmonics.py:38-106 (rsh_cart_0..3), :118-134 (evaluate_sh_at_kernel),
// :154-161 (backward by Taichi autodiff).  out[c] = clamp(sum_d Y_d(dir) * sh[idx,c,d] + 0.5, 0, 1),
// dir = normalize(p[idx] - camera).
//
// HBM-bound: forward reads 4CD + 12 + 8 bytes and writes 4C per visible Gaussian; backward writes
// the dense (N,C,D) gradient (zero rows for culled Gaussians).  One lane per visible Gaussian; the
// coefficient row (192 B at degree 3, C = 3) is read with 16-byte loads when its size allows.

#include "gs_common.h"

namespace {

constexpr float C1 = 0.48860251190292f, C2 = 1.09254843059208f, C3 = 0.94617469575756f, C4 = 0.31539156525252f,
                C5 = 0.54627421529604f, C6 = 0.590043589926644f, C7 = 2.89061144264055f, C8 = 0.304697199642977f,
                C9 = 1.24392110863372f, C10 = 0.497568443453487f, C11 = 1.44530572132028f;

template <int DEG>
__device__ __forceinline__ void rsh(float x, float y, float z, float* Y) {
  Y[0] = 0.282094791773878f;
  if (DEG >= 1) { Y[1] = -C1 * y; Y[2] = C1 * z; Y[3] = -C1 * x; }
  if (DEG >= 2) {
    Y[4] = C2 * (x * y); Y[5] = -C2 * (y * z); Y[6] = C3 * (z * z) - C4; Y[7] = -C2 * (x * z);
    Y[8] = C5 * (x * x) - C5 * (y * y);
  }
  if (DEG >= 3) {
    const float x2 = x * x, y2 = y * y, z2 = z * z;
    Y[9] = -C6 * y * (3.0f * x2 - y2);
    Y[10] = C7 * (x * y) * z;
    Y[11] = C8 * y * (1.5f - 7.5f * z2);
    Y[12] = C9 * z * (1.5f * z2 - 0.5f) - C10 * z;
    Y[13] = C8 * x * (1.5f - 7.5f * z2);
    Y[14] = C11 * z * (x2 - y2);
    Y[15] = -C6 * x * (x2 - 3.0f * y2);
  }
}

// g_dir = sum_d w[d] * dY_d/d(x,y,z)
template <int DEG>
__device__ __forceinline__ void rsh_grad(float x, float y, float z, const float* w, float* g) {
  g[0] = g[1] = g[2] = 0.0f;
  if (DEG >= 1) { g[1] += -C1 * w[1]; g[2] += C1 * w[2]; g[0] += -C1 * w[3]; }
  if (DEG >= 2) {
    g[0] += C2 * y * w[4];            g[1] += C2 * x * w[4];
    g[1] += -C2 * z * w[5];           g[2] += -C2 * y * w[5];
    g[2] += 2.0f * C3 * z * w[6];
    g[0] += -C2 * z * w[7];           g[2] += -C2 * x * w[7];
    g[0] += 2.0f * C5 * x * w[8];     g[1] += -2.0f * C5 * y * w[8];
  }
  if (DEG >= 3) {
    const float x2 = x * x, y2 = y * y, z2 = z * z;
    g[0] += -6.0f * C6 * x * y * w[9];          g[1] += -C6 * (3.0f * x2 - 3.0f * y2) * w[9];
    g[0] += C7 * y * z * w[10];                 g[1] += C7 * x * z * w[10];            g[2] += C7 * x * y * w[10];
    g[1] += C8 * (1.5f - 7.5f * z2) * w[11];    g[2] += -15.0f * C8 * y * z * w[11];
    g[2] += (C9 * (4.5f * z2 - 0.5f) - C10) * w[12];
    g[0] += C8 * (1.5f - 7.5f * z2) * w[13];    g[2] += -15.0f * C8 * x * z * w[13];
    g[0] += 2.0f * C11 * x * z * w[14];         g[1] += -2.0f * C11 * y * z * w[14];   g[2] += C11 * (x2 - y2) * w[14];
    g[0] += -C6 * (3.0f * x2 - 3.0f * y2) * w[15];  g[1] += 6.0f * C6 * x * y * w[15];
  }
}

template <int D>
__device__ __forceinline__ void load_row(const float* row, float* out) {
  if ((D & 3) == 0) {  // rows of 16-byte multiples (degree 1 and 3) are 16-byte aligned for any C
    const float4* r4 = reinterpret_cast<const float4*>(row);
#pragma unroll
    for (int k = 0; k < D / 4; ++k) {
      const float4 t = r4[k];
      out[4 * k] = t.x; out[4 * k + 1] = t.y; out[4 * k + 2] = t.z; out[4 * k + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < D; ++k) out[k] = row[k];
  }
}

// Sharded frame: a rank evaluates the colours of the splats that can reach its tile rows only (the others are in none
// of its tile lists).  The test is a superset of the mapper's: the row span of the ellipse's bounding box at the alpha
// threshold (taichi_lib/grid_query.py:73-91), widened by two tile rows each way -- the reference's tile test checks the
// ellipse's two axes only, so a tile just outside the bounding box can pass it (by less than one tile), and the margin
// also makes the test independent of the last bit of either computation.
#ifndef GS_SH_ROW_MARGIN
#define GS_SH_ROW_MARGIN 2
#endif
struct ShTouch {
  const int* rows;        // optional: evaluate exactly these rows (a list of *rows_count row ids), nothing else is written
  const int* rows_count;
  const float* points2d;  // (v, 7) projected splats, NULL = evaluate every row
  GsShard sh;
  float inv_tile, thr;
  int tile_rows;
};

__device__ __forceinline__ bool sh_touched(const ShTouch& t, int64_t i) {
  const float* g = t.points2d + i * 7;
  const float my = g[1], ax = g[2], ay = g[3], sgx = g[4], sgy = g[5], alpha = g[6];
  if (!(alpha > t.thr)) return false;  // in no tile list at all (mapper: explicit cull)
  const float gscale = sqrtf(2.0f * __logf(alpha / t.thr)) * 1.001f;
  const float v1y = ay * sgx, v2y = ax * sgy;
  const float ey = sqrtf(v1y * v1y + v2y * v2y) * gscale;
  const int lo = int(floorf((my - ey) * t.inv_tile)) - GS_SH_ROW_MARGIN, hi = int(ceilf((my + ey) * t.inv_tile)) + GS_SH_ROW_MARGIN;
  return gs_shard_any_row(t.sh, max(lo, 0), min(hi, t.tile_rows));
}

// CT: channel count known at compile time (3: all loads of a Gaussian's rows are issued before the first use), 0: run-time C
template <int DEG, int CT>
__global__ __launch_bounds__(256) void sh_fwd_kernel(int64_t v, const int* v_dev, int C, const float* params,
                                                     const float* positions, const int64_t* indexes,
                                                     const float* cam, float* out, int out_stride, const ShTouch touch) {
  constexpr int D = (DEG + 1) * (DEG + 1);
  int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (touch.rows != nullptr) {  // row list (sharded frame: the mapper's list of the splats that can reach the rank's rows)
    if (i >= v || i >= *touch.rows_count) return;  // v = capacity of the list here
    i = touch.rows[i];
  } else {
    if (v_dev != nullptr && i >= *v_dev) return;
    if (i >= v) return;
  }
  if (touch.points2d != nullptr && !sh_touched(touch, i)) {
    // never rasterized here; 0.5 = "not clamped" for gs_sh_bwd's mask (the owner ranks apply the real one before the
    // gradients are summed: gs_shard_pack_grads)
    for (int c = 0; c < C; ++c) out[i * out_stride + c] = 0.5f;
    return;
  }
  const int64_t idx = indexes[i];
  if (CT > 0) {
    float rows[CT > 0 ? CT : 1][D];
#pragma unroll
    for (int c = 0; c < CT; ++c) load_row<D>(params + (idx * CT + c) * D, rows[c]);
    const float dx = positions[3 * idx] - cam[0], dy = positions[3 * idx + 1] - cam[1], dz = positions[3 * idx + 2] - cam[2];
    const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
    float Y[D];
    rsh<DEG>(dx / nrm, dy / nrm, dz / nrm, Y);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      float acc = 0.0f;
#pragma unroll
      for (int d = 0; d < D; ++d) acc += Y[d] * rows[c][d];
      out[i * out_stride + c] = fminf(fmaxf(acc + 0.5f, 0.0f), 1.0f);
    }
    return;
  }
  const float dx = positions[3 * idx] - cam[0], dy = positions[3 * idx + 1] - cam[1], dz = positions[3 * idx + 2] - cam[2];
  const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
  float Y[D];
  rsh<DEG>(dx / nrm, dy / nrm, dz / nrm, Y);
  for (int c = 0; c < C; ++c) {
    float row[D];
    load_row<D>(params + (idx * C + c) * D, row);
    float acc = 0.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) acc += Y[d] * row[d];
    out[i * out_stride + c] = fminf(fmaxf(acc + 0.5f, 0.0f), 1.0f);
  }
}

// Measured in round 3 and not kept: the RGB / degree-3 forward with the coefficient rows of a wave fetched as one
// virtual 12-KiB array (lane l of load k takes 16-byte piece k * 64 + l: consecutive lanes read consecutive pieces of a
// row) into an LDS tile, as the dense adjoint writes its rows -- bit-identical colours, 46.4 us against 48.0 us for the
// lane-per-row loads above (189 MB either way, 4.1 TB/s): the forward is not bound by how its loads coalesce.

template <int DEG, bool UNIQUE>
__global__ __launch_bounds__(256) void sh_bwd_kernel(int64_t v, int C, const float* params, const float* positions,
                                                     const int64_t* indexes, const float* cam, const float* gout,
                                                     float* d_params, float* d_positions, float* d_cam) {
  constexpr int D = (DEG + 1) * (DEG + 1);
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  float gd[3] = {0, 0, 0};
  if (i < v) {
    const int64_t idx = indexes[i];
    const float dx = positions[3 * idx] - cam[0], dy = positions[3 * idx + 1] - cam[1],
                dz = positions[3 * idx + 2] - cam[2];
    const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
    const float x = dx / nrm, y = dy / nrm, z = dz / nrm;
    float Y[D], w[D];
    rsh<DEG>(x, y, z, Y);
#pragma unroll
    for (int d = 0; d < D; ++d) w[d] = 0.0f;
    for (int c = 0; c < C; ++c) {
      float row[D];
      load_row<D>(params + (idx * C + c) * D, row);
      float acc = 0.0f;
#pragma unroll
      for (int d = 0; d < D; ++d) acc += Y[d] * row[d];
      const float pre = acc + 0.5f;
      const float g = (pre >= 0.0f && pre <= 1.0f) ? gout[i * C + c] : 0.0f;  // clamp sub-gradient
      float* drow = d_params + (idx * C + c) * D;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if (UNIQUE) drow[d] = g * Y[d];
        else if (g != 0.0f) atomicAdd(drow + d, g * Y[d]);
        w[d] += g * row[d];
      }
    }
    if (DEG >= 1 && (d_positions || d_cam)) {
      float gdir[3];
      rsh_grad<DEG>(x, y, z, w, gdir);
      const float dot = x * gdir[0] + y * gdir[1] + z * gdir[2];
      gd[0] = (gdir[0] - x * dot) / nrm;
      gd[1] = (gdir[1] - y * dot) / nrm;
      gd[2] = (gdir[2] - z * dot) / nrm;
      if (d_positions) {
        if (UNIQUE) { d_positions[3 * idx] = gd[0]; d_positions[3 * idx + 1] = gd[1]; d_positions[3 * idx + 2] = gd[2]; }
        else { atomicAdd(d_positions + 3 * idx, gd[0]); atomicAdd(d_positions + 3 * idx + 1, gd[1]); atomicAdd(d_positions + 3 * idx + 2, gd[2]); }
      }
    }
  }
  if (d_cam && DEG >= 1) {
    __shared__ float s_part[4][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float tot = gs_wave_sum_to_lane63(gd[k]);
      if ((threadIdx.x & 63) == 63) s_part[threadIdx.x >> 6][k] = tot;
    }
    __syncthreads();
    if (threadIdx.x < 3)
      atomicAdd(d_cam + threadIdx.x,
                -(s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x]));
  }
}

// Dense variant for the renderer: the index list is the projection's visible list, so its inverse
// (slot_of, N entries, -1 = culled) is known.  One lane per Gaussian writes its whole gradient row --
// zeros for culled Gaussians -- so the (N,C,D) gradient is produced in a single pass with no memset
// and no atomics.
// STAGED (C*D = 48 floats, i.e. RGB at degree 3): gradient rows are staged in LDS and written out
// as contiguous 12-KiB tiles per wave.
template <int DEG, bool STAGED>
__global__ __launch_bounds__(256) void sh_bwd_dense_kernel(int64_t n, int C, const float* params,
                                                           const float* positions, const int* slot_of,
                                                           const float* cam, const float* gout, int gout_stride,
                                                           const float* fwd_out, int fwd_out_stride,
                                                           float* d_params, float* d_positions, float* d_cam) {
  constexpr int D = (DEG + 1) * (DEG + 1);
  constexpr int TILE_STRIDE = 48 + 4;  // floats per staged row (+16 B: conflict-free, still 16-B aligned)
  __shared__ __attribute__((aligned(16))) float s_tile[STAGED ? 4 : 1][STAGED ? 64 * TILE_STRIDE : 1];
  const int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x;
  float gd[3] = {0, 0, 0};
  if (idx < n) {
    const int slot = slot_of[idx];
    if (slot < 0 && STAGED) {
      float* trow = s_tile[threadIdx.x >> 6] + (threadIdx.x & 63) * TILE_STRIDE;
      for (int e = 0; e < C * D; ++e) trow[e] = 0.0f;
    } else if (slot < 0) {
      for (int c = 0; c < C; ++c) {
        float* drow = d_params + (idx * C + c) * D;
        if ((D & 3) == 0) {
          float4* d4 = reinterpret_cast<float4*>(drow);
#pragma unroll
          for (int k = 0; k < D / 4; ++k) d4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
#pragma unroll
          for (int d = 0; d < D; ++d) drow[d] = 0.0f;
        }
      }
    } else {
      const float dx = positions[3 * idx] - cam[0], dy = positions[3 * idx + 1] - cam[1],
                  dz = positions[3 * idx + 2] - cam[2];
      const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
      const float x = dx / nrm, y = dy / nrm, z = dz / nrm;
      float Y[D], w[D];
      rsh<DEG>(x, y, z, Y);
#pragma unroll
      for (int d = 0; d < D; ++d) w[d] = 0.0f;
      const bool need_dir = d_positions != nullptr || d_cam != nullptr;
      for (int c = 0; c < C; ++c) {
        float row[D];
        float g = gout[int64_t(slot) * gout_stride + c];
        if (fwd_out != nullptr && !need_dir) {
          // the forward's clamped output tells whether the clamp was active: no need to re-read the
          // (N,C,D) coefficients just to recompute it (halves the HBM traffic of this kernel)
          const float o = fwd_out[int64_t(slot) * fwd_out_stride + c];
          if (!(o > 0.0f && o < 1.0f)) g = 0.0f;
        } else {
          load_row<D>(params + (idx * C + c) * D, row);
          float acc = 0.0f;
#pragma unroll
          for (int d = 0; d < D; ++d) acc += Y[d] * row[d];
          const float pre = acc + 0.5f;
          if (!(pre >= 0.0f && pre <= 1.0f)) g = 0.0f;
#pragma unroll
          for (int d = 0; d < D; ++d) w[d] += g * row[d];
        }
        if (STAGED) {
          // row goes to the wave's LDS tile; the whole 64-row tile is stored afterwards with fully
          // coalesced 1-KiB wave stores (a lane-per-row store touches 64 different lines per instruction)
          float* trow = s_tile[threadIdx.x >> 6] + (threadIdx.x & 63) * TILE_STRIDE + c * D;
#pragma unroll
          for (int d = 0; d < D; ++d) trow[d] = g * Y[d];
        } else {
          float* drow = d_params + (idx * C + c) * D;
          if ((D & 3) == 0) {
            float4* d4 = reinterpret_cast<float4*>(drow);
#pragma unroll
            for (int k = 0; k < D / 4; ++k)
              d4[k] = make_float4(g * Y[4 * k], g * Y[4 * k + 1], g * Y[4 * k + 2], g * Y[4 * k + 3]);
          } else {
#pragma unroll
            for (int d = 0; d < D; ++d) drow[d] = g * Y[d];
          }
        }
      }
      if (DEG >= 1 && (d_positions || d_cam)) {
        float gdir[3];
        rsh_grad<DEG>(x, y, z, w, gdir);
        const float dot = x * gdir[0] + y * gdir[1] + z * gdir[2];
        gd[0] = (gdir[0] - x * dot) / nrm;
        gd[1] = (gdir[1] - y * dot) / nrm;
        gd[2] = (gdir[2] - z * dot) / nrm;
      }
    }
    if (d_positions) { d_positions[3 * idx] = gd[0]; d_positions[3 * idx + 1] = gd[1]; d_positions[3 * idx + 2] = gd[2]; }
  }
  if (STAGED) {
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row0 = int64_t(blockIdx.x) * 256 + wave * 64;  // first Gaussian of this wave
    const int64_t rows = n - row0 < 64 ? n - row0 : 64;
    float4* dst = reinterpret_cast<float4*>(d_params + row0 * 48);
    for (int k = 0; k < 12; ++k) {
      const int e = k * 64 + lane;           // float4 index inside the 64 x 12 tile
      const int r = e / 12, q = e - r * 12;
      if (r < rows) {
        typedef float vec4 __attribute__((ext_vector_type(4)));
        const vec4 val = *reinterpret_cast<const vec4*>(s_tile[wave] + r * TILE_STRIDE + q * 4);
        __builtin_nontemporal_store(val, reinterpret_cast<vec4*>(dst) + e);  // written once, read by the optimizer later
      }
    }
  }
  if (d_cam && DEG >= 1) {
    __shared__ float s_part[4][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float tot = gs_wave_sum_to_lane63(gd[k]);
      if ((threadIdx.x & 63) == 63) s_part[threadIdx.x >> 6][k] = tot;
    }
    __syncthreads();
    if (threadIdx.x < 3)
      atomicAdd(d_cam + threadIdx.x,
                -(s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x]));
  }
}

}  // namespace

namespace {
int sh_fwd_launch(int64_t v, const int32_t* v_dev, int32_t channels, int32_t degree, const float* params,
                  const float* positions, const int64_t* indexes, const float* camera_pos, float* out,
                  int32_t out_stride, const ShTouch& touch, void* stream, const char* who) {
  GS_REQUIRE(degree >= 0 && degree <= 3, GS_ERR_UNSUPPORTED, "%s: SH degree %d not in [0,3]", who, degree);
  GS_REQUIRE(channels >= 1 && channels <= GS_MAX_SH_CHANNELS, GS_ERR_UNSUPPORTED, "%s: %d channels", who, channels);
  if (v == 0) return GS_OK;
  GS_REQUIRE(params && positions && indexes && camera_pos && out, GS_ERR_INVALID_ARGUMENT, "%s: NULL buffer", who);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (out_stride <= 0) out_stride = channels;
  const dim3 grid(unsigned(gs_div_up(v, 256))), block(256);
#define SH_FWD_C(DEG, CT)                                                                                         \
  hipLaunchKernelGGL((sh_fwd_kernel<DEG, CT>), grid, block, 0, s, v, v_dev, channels, params, positions, indexes, \
                     camera_pos, out, out_stride, touch)
#define SH_FWD(DEG)                                                                                               \
  if (channels == 3) SH_FWD_C(DEG, 3);                                                                            \
  else SH_FWD_C(DEG, 0)
  switch (degree) {
    case 0: SH_FWD(0); break;
    case 1: SH_FWD(1); break;
    case 2: SH_FWD(2); break;
    default: SH_FWD(3); break;
  }
  GS_CHECK_LAUNCH(who);
  return GS_OK;
}
}  // namespace

extern "C" int gs_sh_fwd(int64_t v, const int32_t* v_dev, int32_t channels, int32_t degree, const float* params,
                         const float* positions, const int64_t* indexes, const float* camera_pos, float* out,
                         int32_t out_stride, void* stream) {
  ShTouch touch{};
  touch.points2d = nullptr;
  return sh_fwd_launch(v, v_dev, channels, degree, params, positions, indexes, camera_pos, out, out_stride, touch,
                       stream, "gs_sh_fwd");
}

extern "C" int gs_sh_fwd_rows(int64_t v, const int32_t* rows, const int32_t* rows_count, int32_t channels,
                              int32_t degree, const float* params, const float* positions, const int64_t* indexes,
                              const float* camera_pos, float* out, int32_t out_stride, void* stream) {
  GS_REQUIRE(rows && rows_count, GS_ERR_INVALID_ARGUMENT, "gs_sh_fwd_rows: NULL row list");
  ShTouch touch{};
  touch.rows = rows;
  touch.rows_count = rows_count;
  return sh_fwd_launch(v, nullptr, channels, degree, params, positions, indexes, camera_pos, out, out_stride, touch,
                       stream, "gs_sh_fwd_rows");
}

extern "C" int gs_sh_fwd_shard(int64_t v, const int32_t* v_dev, int32_t channels, int32_t degree, const float* params,
                               const float* positions, const int64_t* indexes, const float* camera_pos,
                               const float* points2d, int32_t height, const GsRasterConfig* cfg,
                               const GsRowShard* shard, float* out, int32_t out_stride, void* stream) {
  GS_REQUIRE(cfg && cfg->tile_size >= 1 && height >= 0, GS_ERR_INVALID_ARGUMENT, "gs_sh_fwd_shard: config / height");
  GS_REQUIRE(points2d || v == 0, GS_ERR_INVALID_ARGUMENT, "gs_sh_fwd_shard: points2d is NULL");
  ShTouch touch{};
  touch.points2d = points2d;
  touch.tile_rows = int(gs_div_up(height, cfg->tile_size));
  if (int rc = gs_make_shard(shard, touch.tile_rows, &touch.sh)) return rc;
  touch.inv_tile = 1.0f / float(cfg->tile_size);
  touch.thr = cfg->alpha_threshold;
  return sh_fwd_launch(v, v_dev, channels, degree, params, positions, indexes, camera_pos, out, out_stride, touch,
                       stream, "gs_sh_fwd_shard");
}

#define SH_BWD_LAUNCH(DEG, UNIQ)                                                                                    \
  hipLaunchKernelGGL((sh_bwd_kernel<DEG, UNIQ>), grid, block, 0, s, v, channels, params, positions, indexes,        \
                     camera_pos, grad_out, d_params, d_positions, d_camera_pos)

extern "C" int gs_sh_bwd(int64_t n, int64_t v, int32_t channels, int32_t degree, const float* params,
                         const float* positions, const int64_t* indexes, int32_t indexes_unique,
                         const int32_t* slot_of, const float* camera_pos, const float* grad_out,
                         int32_t grad_out_stride, const float* fwd_out, int32_t fwd_out_stride, float* d_params,
                         float* d_positions, float* d_camera_pos, void* stream) {
  GS_REQUIRE(degree >= 0 && degree <= 3, GS_ERR_UNSUPPORTED, "gs_sh_bwd: SH degree %d not in [0,3]", degree);
  GS_REQUIRE(channels >= 1 && channels <= GS_MAX_SH_CHANNELS, GS_ERR_UNSUPPORTED, "gs_sh_bwd: %d channels", channels);
  GS_REQUIRE(n == 0 || d_params, GS_ERR_INVALID_ARGUMENT, "gs_sh_bwd: d_params is NULL");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int D = (degree + 1) * (degree + 1);
  if (grad_out_stride <= 0) grad_out_stride = channels;
  if (slot_of != nullptr && n > 0) {
    // dense single pass over all n Gaussians (the renderer's case)
    GS_REQUIRE(params && positions && camera_pos && grad_out, GS_ERR_INVALID_ARGUMENT, "gs_sh_bwd: NULL buffer");
    if (d_camera_pos && hipMemsetAsync(d_camera_pos, 0, 12, s) != hipSuccess) {
      gs_set_error("gs_sh_bwd: hipMemsetAsync failed");
      return GS_ERR_LAUNCH;
    }
    const dim3 grid(unsigned(gs_div_up(n, 256))), block(256);
#define SH_DENSE(DEG)                                                                                               \
  hipLaunchKernelGGL((sh_bwd_dense_kernel<DEG, false>), grid, block, 0, s, n, channels, params, positions, slot_of, \
                     camera_pos, grad_out, grad_out_stride, fwd_out, fwd_out_stride > 0 ? fwd_out_stride : channels,   \
                     d_params, d_positions, d_camera_pos)
    if (degree == 3 && channels == 3) {
      hipLaunchKernelGGL((sh_bwd_dense_kernel<3, true>), grid, block, 0, s, n, channels, params, positions, slot_of,
                         camera_pos, grad_out, grad_out_stride, fwd_out, fwd_out_stride > 0 ? fwd_out_stride : channels,
                         d_params, d_positions, d_camera_pos);
    } else {
      switch (degree) {
        case 0: SH_DENSE(0); break;
        case 1: SH_DENSE(1); break;
        case 2: SH_DENSE(2); break;
        default: SH_DENSE(3); break;
      }
    }
    GS_CHECK_LAUNCH("gs_sh_bwd/dense");
    return GS_OK;
  }
  GS_REQUIRE(grad_out_stride == channels, GS_ERR_INVALID_ARGUMENT, "gs_sh_bwd: strided grad_out needs slot_of");
  bool ok = true;
  if (n > 0) ok &= hipMemsetAsync(d_params, 0, size_t(n) * channels * D * 4, s) == hipSuccess;
  if (n > 0 && d_positions) ok &= hipMemsetAsync(d_positions, 0, size_t(n) * 12, s) == hipSuccess;
  if (d_camera_pos) ok &= hipMemsetAsync(d_camera_pos, 0, 12, s) == hipSuccess;
  if (!ok) { gs_set_error("gs_sh_bwd: hipMemsetAsync failed"); return GS_ERR_LAUNCH; }
  if (v == 0) return GS_OK;
  GS_REQUIRE(params && positions && indexes && camera_pos && grad_out, GS_ERR_INVALID_ARGUMENT, "gs_sh_bwd: NULL buffer");
  const dim3 grid(unsigned(gs_div_up(v, 256))), block(256);
  if (indexes_unique) {
    switch (degree) {
      case 0: SH_BWD_LAUNCH(0, true); break;
      case 1: SH_BWD_LAUNCH(1, true); break;
      case 2: SH_BWD_LAUNCH(2, true); break;
      default: SH_BWD_LAUNCH(3, true); break;
    }
  } else {
    switch (degree) {
      case 0: SH_BWD_LAUNCH(0, false); break;
      case 1: SH_BWD_LAUNCH(1, false); break;
      case 2: SH_BWD_LAUNCH(2, false); break;
      default: SH_BWD_LAUNCH(3, false); break;
    }
  }
  GS_CHECK_LAUNCH("gs_sh_bwd");
  return GS_OK;
}
