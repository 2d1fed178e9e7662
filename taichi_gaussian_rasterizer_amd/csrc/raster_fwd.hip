// raster_fwd.hip -- front-to-back alpha-blend forward (reference rasterizer/forward.py:25-137).
//
// MI355X mapping (not the reference's 256-thread / 1-pixel-per-thread block):
//   * ONE wave64 per 16x16 pixel region (a whole tile at tile_size 16, a quadrant at 32; an 8x8
//     tile is one sub-block).  Lane l owns pixel (l&7, l>>3) of each of the region's four 8x8
//     sub-blocks, i.e. 4 pixels per lane.  A workgroup is a single wave, so there are no
//     s_barriers and no cross-wave LDS traffic.  Grids too small to fill the chip that way (fewer
//     than ~2k regions: training-size images, strips of a sharded frame) use 16x8 or 8x8 regions
//     (2 / 1 pixels per lane, 2x / 4x the waves; NB template parameter): 1.5x faster at 256x256.
//   * The tile's splat list is staged 64 at a time: lane j gathers splat j (28 B + 4F B row,
//     index from overlap_to_point), pre-multiplies the ellipse frame (axis/sigma scaled so that
//     alpha = a * exp2(-(tx^2+ty^2))) and writes one LDS record.  The blend loop then reads each
//     record with wave-uniform (broadcast) ds_read_b128s.
//   * While staging, each lane also tests its splat against the four 8x8 sub-blocks in the
//     ellipse frame (conservative, with a margin that keeps alpha strictly below the threshold
//     outside) and stores a 4-bit mask; the blend loop branches on it with SCALAR branches, so
//     sub-blocks a splat cannot touch cost nothing and nothing diverges.  Results are identical
//     to evaluating every pixel: a skipped pixel would have failed `alpha > alpha_threshold`.
//   * Launch: grid = regions, XCD-aware remap so that neighbouring tiles (which gather the same
//     splat rows) share an L2.
//
// Roofline: algorithmic HBM bytes K*(4 + 28 + 4F) + 16T + 4P(F+1); the kernel is
// VALU-bound (about 20 VALU + 1 v_exp_f32 per evaluated pixel-splat pair).

#include "gs_common.h"

namespace {

struct FwdArgs {
  const float* points;
  const float* features;
  const int2* ranges;
  const int* o2p;
  float* image;
  float* alpha;
  float* visibility;
  int W, H, F;
  int tiles_wide;
  int tile_size;
  int sub_x, sub_y;  // wave regions per tile along x / y
  int num_items;
  int num_tiles;
  const int* heavy;  // optional (device): the first *heavy entries of tile_order get four 8x8 workgroups each
  int heavy_cap;
  const int* tile_order;  // optional launch order of the items (heaviest first)  // tiles * sub * sub
  float cmax, thr, inv_thr, sat_level;
  float cut;  // a region is walked while some pixel's transmittance is above this (GsRasterConfig.forward_cut)
  int blend, vis, aa;
  GsShard sh;  // owned tile rows: tile ids are local, H is the full image height, the image holds the owned rows
};

// pixel origin of a (local) tile in the full image, and the row of the output buffer it starts at
__device__ __forceinline__ void tile_origin(const FwdArgs& a, int tile, int& x0, int& y0, int& yout0) {
  const int lty = tile / a.tiles_wide;
  x0 = (tile - lty * a.tiles_wide) * a.tile_size;
  y0 = gs_shard_global_row(a.sh, lty) * a.tile_size;
  yout0 = lty * a.tile_size;
}

__device__ __forceinline__ float s_sig(float x, float inv_sigma) {
  const float z = x * inv_sigma;
  const float e = -1.6f * z - 0.07f * z * z * z;
  return gs_rcp_fast(1.0f + gs_exp2_fast(e * 1.44269504088896341f));
}

// One axis of the antialiased pdf (taichi_lib/generic.py:341-357): D(t) = S((t + 0.5) / s) - S((t - 0.5) / s) with
// S(z) = 1 / (1 + e(z)), e(z) = exp(-(1.6 z + 0.07 z^3)), returned as num / den = (e_b - e_a) / ((1 + e_a)(1 + e_b)) so
// that both axes share ONE reciprocal.  D is even (S(-z) = 1 - S(z)), so |t| is used: a >= 0 keeps e_a <= 1, and b is
// clamped at -5, where S(b) < 5e-8 (e_b stays below 2e7: no overflow in the product of the two axes).
__device__ __forceinline__ void aa_axis(float t, float inv_sigma, float& num, float& den) {
  const float u = fabsf(t);
  const float za = (u + 0.5f) * inv_sigma, zb = fmaxf((u - 0.5f) * inv_sigma, -5.0f);
  const float c1 = -1.6f * 1.44269504088896341f, c3 = -0.07f * 1.44269504088896341f;
  const float ea = gs_exp2_fast(za * (c1 + c3 * za * za)), eb = gs_exp2_fast(zb * (c1 + c3 * zb * zb));
  num = eb - ea;
  den = (1.0f + ea) * (1.0f + eb);
}

// NB: 8x8 sub-blocks per wave (1, 2 or 4; gs_raster_sub_blocks picks it from the grid size).  FP: padded feature width.
// MODE 3: lean quantile pass (no blending, no antialias, no statistics: the median-depth pass of renderer.py:203-208);
// MODE 0: blend only (lean); 1: blend + per-splat visibility (training with pruning statistics);
// 2: runtime switches for quantile mode / antialias (+ visibility).
// (Measured with it and not kept: the blend code twice, with and without the v_med3_f32 of min(alpha, clamp_max_alpha)
// -- which can only bite when the splat's opacity exceeds the clamp -- and a scalar branch per splat on a ballot of
// (opacity > clamp): 0.258 against 0.215 ms; four copies of the blend code undo the two-register-set pipeline.)
#ifndef GS_FWD_MASK_BALLOTS
#define GS_FWD_MASK_BALLOTS 1
#endif
template <int NB, int FP, int MODE>
__device__ __forceinline__ void raster_fwd_body(const FwdArgs& a, int tile, int x0, int y0, int yout0,
                                                float4 (*s_geo)[(MODE == 2 ? 3 : 2) + (FP + 3) / 4], float* s_vis,
                                                int* s_idx) {
  // staged record of a splat: GEO_V4 float4s of geometry, then its feature row (one LDS address, b128 reads)
  constexpr int GEO_V4 = MODE == 2 ? 3 : 2, FEAT_V4 = (FP + 3) / 4;
  constexpr bool FULL = MODE == 2, VIS = MODE == 1 || MODE == 2, QUANT = MODE == 3;
  const bool blend = QUANT ? false : (FULL ? a.blend != 0 : true);
  const int lane = threadIdx.x;
  const int lx = lane & 7, ly = lane >> 3;
  // Tr = 1 - (accumulated weight W of forward.py:84-128): the transmittance in front of the next splat
  float Xf[NB], Yf[NB], Tr[NB], acc[NB][FP];
  bool inb[NB], done[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int X = x0 + (b & 1) * 8 + lx, Y = y0 + (b >> 1) * 8 + ly;
    inb[b] = X < a.W && Y < a.H;
    // lean modes: pixel centres relative to the wave's origin (see the staging below); general mode: absolute
    Xf[b] = FULL ? float(X) + 0.5f : float((b & 1) * 8 + lx) + 0.5f;
    Yf[b] = FULL ? float(Y) + 0.5f : float((b >> 1) * 8 + ly) + 0.5f;
    Tr[b] = inb[b] ? 1.0f : 0.0f;  // forward.py:53-54: out-of-image pixels start with W = 1
    done[b] = false;
#pragma unroll
    for (int c = 0; c < FP; ++c) acc[b][c] = 0.0f;
  }

  const int2 range_v = a.ranges[tile];
  int2 range;  // wave-uniform: loop bounds in scalar registers
  range.x = __builtin_amdgcn_readfirstlane(range_v.x);
  range.y = __builtin_amdgcn_readfirstlane(range_v.y);
  const float k_exp = 0.84932180028801904f;  // sqrt(0.5 * log2(e)): exp(-0.5 t^2) = exp2(-(k t)^2)

  for (int g0 = range.x; g0 < range.y; g0 += 64) {
    // The reference's forward never stops (forward.py:84-128).  Once every pixel of the region has less than
    // cfg->forward_cut of its transmittance left, everything still to come changes a pixel by less than
    // forward_cut * max|feature| in total, so the rest of a crowded tile's list is skipped (forward_cut = 0 acts as
    // 2^-25, see gs_raster_fwd: within N * 2^-24 * max|feature| of the reference for N remaining splats).
    // The same holds per 8x8 sub-block: a saturated one is masked out for the rest of the list.
    int live = 0;
    if (blend) {
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if (__ballot(Tr[b] > a.cut) != 0ull) live |= 1 << b;
    } else {
      // quantile mode (forward.py:109-114): a pixel is finished once a splat has taken it to the level; the walk ends
      // with the region's last unfinished pixel (the median-depth pass stops after the front of each list)
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if (__ballot(inb[b] && !done[b]) != 0ull) live |= 1 << b;
    }
    if (live == 0) break;
    const int cnt = __builtin_amdgcn_readfirstlane(min(64, range.y - g0));
    int staged_mask = 0;
    // ---- stage up to 64 splats: lane j <- splat g0 + j
    if (lane < cnt) {
      const int idx = a.o2p[g0 + lane];
      const float* p = a.points + int64_t(idx) * 7;
      const float mx = p[0], my = p[1], ax = p[2], ay = p[3], sx = p[4], sy = p[5], al = p[6];
      const float isx = gs_rcp_fast(sx), isy = gs_rcp_fast(sy);  // v_rcp_f32: 1 ulp
      const float Ax = ax * isx * k_exp, Ay = ay * isx * k_exp;
      const float Bx = -ay * isy * k_exp, By = ax * isy * k_exp;
      // conservative sub-block mask: alpha*exp2(-(tx^2+ty^2)) > thr needs tx^2+ty^2 < log2(alpha/thr)
      int mask = 0;
      if (FULL && a.aa) {
        mask = gs_sub_block_mask_antialias<NB>(ax, ay, sx, sy, al, a.inv_thr, s_sig(0.5f, isx) - s_sig(-0.5f, isx),
                                               s_sig(0.5f, isy) - s_sig(-0.5f, isy), float(x0) + 0.5f - mx,
                                               float(y0) + 0.5f - my);
      } else if (al > a.thr) {
        mask = gs_sub_block_mask<NB>(Ax, Ay, Bx, By, __log2f(al * a.inv_thr), float(x0) + 0.5f - mx,
                                     float(y0) + 0.5f - my);
      }
      staged_mask = mask;
      // Lean modes: the ellipse-frame coordinates of a pixel are tx = A . (X - m) = A . (X - origin) + A . (origin - m):
      // the second term is formed once per (region, splat) here, and a pixel's tx is two fma on its origin-relative
      // centre (|X - origin| < 16: no cancellation beyond what X - m has) instead of two subtractions, a multiply and
      // an fma.  The backward uses the same expression: same bits, same hit / miss per pixel.
      const float ox = float(x0) - mx, oy = float(y0) - my;
      if (FULL) s_geo[lane][0] = make_float4(mx, my, Ax, Ay);
      else s_geo[lane][0] = make_float4(__builtin_fmaf(Ax, ox, Ay * oy), __builtin_fmaf(Bx, ox, By * oy), Ax, Ay);
      // lean modes carry -log2(opacity): it starts the exponent's fma chain, so v_exp_f32 returns alpha itself
      s_geo[lane][1] = make_float4(Bx, By, FULL ? al : -__log2f(al), __int_as_float(mask));
      if (FULL) s_geo[lane][2] = make_float4(ax, ay, isx, isy);
      if (VIS) {
        s_vis[lane] = 0.0f;
        s_idx[lane] = idx;
      }
      const float* f = a.features + int64_t(idx) * a.F;
#pragma unroll
      for (int q = 0; q < FEAT_V4; ++q) {
        float fv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) fv[k] = (4 * q + k < FP && 4 * q + k < a.F) ? f[4 * q + k] : 0.0f;
        s_geo[lane][GEO_V4 + q] = make_float4(fv[0], fv[1], fv[2], fv[3]);
      }
    }
    __syncthreads();  // single-wave workgroup: compiles to a wait on the LDS writes, no s_barrier

    // ---- blend.  The next splat's record is fetched from LDS while the current one is blended (one wave-uniform
    // ds_read burst per splat, its latency hidden behind ~40-90 VALU instructions).  Two register sets take turns
    // (A is blended while B is in flight and vice versa): rotating ONE set cost 12 v_mov per splat, a fifth of the
    // kernel's vector instructions.
    struct Rec {
      float4 g0, g1, g2;
      float f[FP];
    };
    auto fetch = [&](int j, Rec& r) {
      const int jj = j < cnt ? j : cnt - 1;
      r.g0 = s_geo[jj][0];
      r.g1 = s_geo[jj][1];
      if (FULL) r.g2 = s_geo[jj][2];
#pragma unroll
      for (int q = 0; q < FEAT_V4; ++q) {
        const float4 fq = s_geo[jj][GEO_V4 + q];
        const float fv[4] = {fq.x, fq.y, fq.z, fq.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (4 * q + k < FP) r.f[4 * q + k] = fv[k];
      }
    };
    // GS_FWD_MASK_BALLOTS: the staged splats' sub-block masks as four scalar ballots (see raster_bwd.hip)
    uint64_t reach[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
      reach[b] = (GS_FWD_MASK_BALLOTS && ((live >> b) & 1)) ? __ballot((staged_mask >> b) & 1) : 0ull;
    auto blend_splat = [&](int j, const Rec& r) {
      const float4 g0v = r.g0, g1v = r.g1, g2v = FULL ? r.g2 : make_float4(0, 0, 0, 0);
      const float(&feat)[FP] = r.f;
      int mask = 0;
      if (GS_FWD_MASK_BALLOTS) {
#pragma unroll
        for (int b = 0; b < NB; ++b) mask |= int((reach[b] >> j) & 1ull) << b;
      } else {
        mask = __builtin_amdgcn_readfirstlane(__float_as_int(g1v.w)) & live;
      }
      float vis_sum = 0.0f;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (!(mask & (1 << b))) continue;  // scalar branch
        const float dx = FULL ? Xf[b] - g0v.x : 0.0f, dy = FULL ? Yf[b] - g0v.y : 0.0f;
        float p, alpha;
        if (FULL && a.aa) {
          // taichi_lib/generic.py:347-357
          const float tx = dx * g2v.x + dy * g2v.y, ty = dy * g2v.x - dx * g2v.y;
          float nx, dx_, ny, dy_;
          aa_axis(tx, g2v.z, nx, dx_);
          aa_axis(ty, g2v.w, ny, dy_);
          // tau sx sy D(tx) D(ty); sx sy = 1 / (isx isy) goes into the same reciprocal
          p = 6.28318530717958648f * nx * ny * gs_rcp_fast(dx_ * dy_ * g2v.z * g2v.w);
          alpha = g1v.z * p;
        } else {
          const float tx = FULL ? dx * g0v.z + dy * g0v.w
                                : __builtin_fmaf(g0v.z, Xf[b], __builtin_fmaf(g0v.w, Yf[b], g0v.x));
          const float ty = FULL ? dx * g1v.x + dy * g1v.y
                                : __builtin_fmaf(g1v.x, Xf[b], __builtin_fmaf(g1v.y, Yf[b], g0v.y));
          if (FULL) {
            alpha = g1v.z * gs_exp2_fast(-(tx * tx + ty * ty));
          } else {
            alpha = gs_exp2_fast(-__builtin_fmaf(ty, ty, __builtin_fmaf(tx, tx, g1v.z)));  // opacity * pdf
          }
        }
        const float al = __builtin_amdgcn_fmed3f(alpha, a.cmax, -1.0f);  // min(alpha, cmax) (forward.py:98-99)
        bool hit = al > a.thr;
        if (FULL || QUANT) hit = hit && !done[b];
        const float w = (hit ? al : 0.0f) * Tr[b];
        Tr[b] -= w;
        if (blend) {
#pragma unroll
          for (int c = 0; c < FP; ++c) acc[b][c] += feat[c] * w;
        } else if (hit) {  // forward.py:109-114 quantile mode: first splat that reaches the level
          if (1.0f - Tr[b] >= a.sat_level) {
#pragma unroll
            for (int c = 0; c < FP; ++c) acc[b][c] = feat[c];
            done[b] = true;
          }
        }
        vis_sum += w;
      }
      if (VIS && a.vis) {
        // forward.py:116-128: per-splat visibility = sum of blend weights over the tile's pixels
        if (__ballot(vis_sum != 0.0f) != 0ull) {  // wave-uniform by construction: a scalar compare of the ballot
          const float tot = gs_wave_sum_to_lane63(vis_sum);
          if (lane == 63) s_vis[j] = tot;  // each staged splat is visited once per round: a store, not a read-modify-write
        }
      }
    };
    Rec ra, rb;
    fetch(0, ra);
    for (int j = 0; j < cnt; j += 2) {
      fetch(j + 1, rb);
      blend_splat(j, ra);
      fetch(j + 2, ra);
      if (j + 1 < cnt) blend_splat(j + 1, rb);
    }
    if (VIS && a.vis) {
      __syncthreads();
      if (lane < cnt && s_vis[lane] != 0.0f) atomicAdd(a.visibility + s_idx[lane], s_vis[lane]);
    }
    __syncthreads();  // records are overwritten by the next group
  }

#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (!inb[b]) continue;
    const int X = x0 + (b & 1) * 8 + lx, Y = y0 + (b >> 1) * 8 + ly;
    const int64_t pix = int64_t(Y - y0 + yout0) * a.W + X;
    float* out = a.image + pix * a.F;
#pragma unroll
    for (int c = 0; c < FP; ++c)
      if (c < a.F) out[c] = acc[b][c];
    a.alpha[pix] = blend ? 1.0f - Tr[b] : (Tr[b] < 1.0f ? 1.0f : 0.0f);  // forward.py:134-137
  }
}

// Block -> work.  With a launch order from the mapper: its first `*heavy` tiles (the fullest ones; tile_size 16
// only) are rasterized by FOUR workgroups each, one per 8x8 quadrant, the others by workgroups of the grid's
// own wave region -- a launch cannot end before its fullest tile has been walked by one wave, which is what
// bounds small grids (strips of a sharded frame, training-size images).  Without an order: XCD-contiguous bands.
template <int NB, int FP, int MODE>
__global__ __launch_bounds__(64) void raster_fwd_kernel(const FwdArgs a) {
  __shared__ float4 s_geo[64][(MODE == 2 ? 3 : 2) + (FP + 3) / 4];
  __shared__ float s_vis[(MODE == 1 || MODE == 2) ? 64 : 1];
  __shared__ int s_idx[(MODE == 1 || MODE == 2) ? 64 : 1];
  const int per_tile = a.sub_x * a.sub_y;
  constexpr int RW = NB == 1 ? 8 : 16, RH = NB == 4 ? 16 : 8;  // the wave's pixel region: NB 8x8 sub-blocks
  int tile, quad;
  if (a.tile_order) {
    const int b = blockIdx.x;
    const int heavy = (NB > 1 && a.heavy) ? min(*a.heavy, a.heavy_cap) : 0;
    if (NB > 1 && b < 4 * heavy) {
      tile = a.tile_order[b >> 2];
      int x0, y0, yout0;
      tile_origin(a, tile, x0, y0, yout0);
      x0 += (b & 1) * 8; y0 += ((b >> 1) & 1) * 8; yout0 += ((b >> 1) & 1) * 8;
      if (x0 < a.W && y0 < a.H) raster_fwd_body<1, FP, MODE>(a, tile, x0, y0, yout0, s_geo, s_vis, s_idx);
      return;
    }
    const int c = b - 4 * heavy, rank = heavy + c / per_tile;
    if (rank >= a.num_tiles) return;
    tile = a.tile_order[rank];
    quad = c % per_tile;
  } else {
    const int item = gs_xcd_remap(blockIdx.x, a.num_items);
    if (item < 0) return;
    tile = item / per_tile;
    quad = item - tile * per_tile;
  }
  int x0, y0, yout0;
  tile_origin(a, tile, x0, y0, yout0);
  x0 += (quad % a.sub_x) * RW; y0 += (quad / a.sub_x) * RH; yout0 += (quad / a.sub_x) * RH;
  if (x0 >= a.W || y0 >= a.H) return;
  raster_fwd_body<NB, FP, MODE>(a, tile, x0, y0, yout0, s_geo, s_vis, s_idx);
}

template <int NB, int MODE>
int launch_fp(const FwdArgs& a, hipStream_t s) {
  const int grid = 8 * int(gs_div_up(a.num_items + (a.heavy ? 4 * a.heavy_cap : 0), 8));
  if (MODE == 3 && a.F == 1) hipLaunchKernelGGL((raster_fwd_kernel<NB, 1, MODE>), dim3(grid), dim3(64), 0, s, a);
  else if (a.F <= 3) hipLaunchKernelGGL((raster_fwd_kernel<NB, 3, MODE>), dim3(grid), dim3(64), 0, s, a);
  else if (a.F <= 5) hipLaunchKernelGGL((raster_fwd_kernel<NB, 5, MODE>), dim3(grid), dim3(64), 0, s, a);
  else if (a.F <= 8) hipLaunchKernelGGL((raster_fwd_kernel<NB, 8, MODE>), dim3(grid), dim3(64), 0, s, a);
  else hipLaunchKernelGGL((raster_fwd_kernel<NB, 32, MODE>), dim3(grid), dim3(64), 0, s, a);
  GS_CHECK_LAUNCH("gs_raster_fwd");
  return GS_OK;
}

}  // namespace

extern "C" int gs_raster_fwd(int64_t v, int32_t num_features, const float* points, const float* features,
                             const int32_t* tile_ranges, const int32_t* overlap_to_point, int64_t k, int32_t width,
                             int32_t height, const GsRasterConfig* cfg, const int32_t* tile_order,
                             const int32_t* heavy_tiles, float* image, float* alpha, float* visibility,
                             const GsRowShard* shard, void* stream) {
  if (int rc = gs_check_cfg(cfg)) return rc;
  GS_REQUIRE(width > 0 && height > 0, GS_ERR_INVALID_ARGUMENT, "gs_raster_fwd: image size %dx%d", width, height);
  GS_REQUIRE(num_features >= 1 && num_features <= GS_MAX_FEATURES, GS_ERR_UNSUPPORTED,
             "gs_raster_fwd: feature width %d not in [1,%d]", num_features, GS_MAX_FEATURES);
  GS_REQUIRE(image && alpha && tile_ranges, GS_ERR_INVALID_ARGUMENT, "gs_raster_fwd: NULL output or ranges");
  GS_REQUIRE(k == 0 || (points && features && overlap_to_point), GS_ERR_INVALID_ARGUMENT,
             "gs_raster_fwd: NULL input with %lld overlaps", (long long)k);
  const bool vis = cfg->compute_visibility || cfg->compute_point_heuristic;
  GS_REQUIRE(!vis || visibility || v == 0, GS_ERR_INVALID_ARGUMENT, "gs_raster_fwd: visibility buffer is NULL");
  const int ts = cfg->tile_size;
  FwdArgs a;
  a.points = points; a.features = features; a.ranges = reinterpret_cast<const int2*>(tile_ranges);
  a.o2p = overlap_to_point; a.image = image; a.alpha = alpha; a.visibility = visibility;
  a.W = width; a.H = height; a.F = num_features;
  a.tiles_wide = int(gs_div_up(width, ts));
  a.tile_size = ts;
  if (int rc = gs_make_shard(shard, int(gs_div_up(height, ts)), &a.sh)) return rc;
  const int num_tiles = a.tiles_wide * a.sh.local_rows;
  if (num_tiles == 0) return GS_OK;
  const int nb = gs_raster_sub_blocks(cfg, num_tiles, 0);
  a.sub_x = ts / (nb == 1 ? 8 : 16);
  a.sub_y = ts / (nb == 4 ? 16 : 8);
  a.num_items = num_tiles * a.sub_x * a.sub_y;
  a.tile_order = tile_order;
  a.num_tiles = num_tiles;
  // the split needs the 2x2-quadrant geometry of a 16-pixel tile and a launch order to index into
  a.heavy = (tile_order && ts == 16 && nb > 1) ? heavy_tiles : nullptr;
  a.heavy_cap = num_tiles / 4;
  if (cfg->tune_no_heavy_split) a.heavy = nullptr;
  a.cmax = cfg->clamp_max_alpha; a.thr = cfg->alpha_threshold; a.inv_thr = 1.0f / cfg->alpha_threshold;
  a.sat_level = 1.0f - cfg->saturate_threshold;
  // below 2^-25 (half an ulp of 1) the reference's own f32 accumulation W += w no longer changes W, but it still adds
  // alpha * (1 - W) * feature for every remaining splat with 1 - W stuck at ~2^-24; this kernel carries T itself and
  // stops here: forward_cut = 0 differs from the reference by < N * 2^-24 * max|feature| (N remaining splats)
  a.cut = cfg->forward_cut > 2.98023223876953125e-08f ? cfg->forward_cut : 2.98023223876953125e-08f;
  a.blend = cfg->use_alpha_blending; a.vis = vis; a.aa = cfg->antialias;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int mode = (!a.blend && !a.aa && !a.vis) ? 3 : (!a.blend || a.aa) ? 2 : a.vis ? 1 : 0;
  if (nb == 1)
    return mode == 3 ? launch_fp<1, 3>(a, s) : mode == 2 ? launch_fp<1, 2>(a, s) : mode == 1 ? launch_fp<1, 1>(a, s)
                                                                                              : launch_fp<1, 0>(a, s);
  if (nb == 2)
    return mode == 3 ? launch_fp<2, 3>(a, s) : mode == 2 ? launch_fp<2, 2>(a, s) : mode == 1 ? launch_fp<2, 1>(a, s)
                                                                                              : launch_fp<2, 0>(a, s);
  return mode == 3 ? launch_fp<4, 3>(a, s) : mode == 2 ? launch_fp<4, 2>(a, s) : mode == 1 ? launch_fp<4, 1>(a, s)
                                                                                            : launch_fp<4, 0>(a, s);
}
