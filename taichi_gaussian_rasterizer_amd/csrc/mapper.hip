// mapper.hip -- tile-overlap cull, per-tile bucketing and per-tile depth sort
// (reference mapper/tile_mapper.py:74-196, taichi_lib/grid_query.py:10-91).
//
// COMPILED WITH -ffp-contract=off.  Every f32 operation of the grid query is a single correctly
// rounded IEEE op in the same order as oracle/gsplat_oracle.cpp (the square roots through gs_det_sqrtf: hipcc's
// __fsqrt_rn is the 1-ulp native instruction), and the logarithm is
// gs_det_logf (include/gs_detmath.h), so the integer results -- which tiles a splat touches, the
// sort keys, the order inside every tile -- are bit-identical to the CPU oracle.
//
// Fused path (gs_map_prepare / gs_map_finish), designed for MI355X rather than around a
// library radix sort:
//   bin     : order the Gaussians by the screen region (8x8 tiles; larger when the image has more than
//             1024 of them) of their centre -- LDS histograms per workgroup of 1024 Gaussians, then either one
//             returning atomic per (workgroup, region) (small frames: no launch of its own) or a per-region scan
//             over the workgroups (large frames), and a scatter.  In the frame calls the histogram pass IS the
//             projection's compaction pass (compact_bin_kernel).
//   count   : 1 lane per Gaussian, OBB query; overlaps are counted in an LDS window over the
//             workgroup's region (+ border) and flushed with one atomic per window tile into a T-entry
//             histogram (scattered 4-B global atomics only reach ~20 G/s on MI355X).
//   scan    : single workgroup exclusive scan of the histogram -> tile_ranges, cursors, K, max, and the
//             rasterizer's launch order (tiles by descending population).
//   emit    : the query again; one returning atomic per (workgroup, tile) reserves a range of the
//             tile's bucket, LDS atomics place the 64-bit composites (depth key << 32 | index).
//   sort    : one wave per tile rank-sorts its bucket (bitonic in LDS / global for crowded tiles).
//             The composite makes the result independent of the atomic arrival order and equal
//             to the reference's stable radix sort of (tile << 32 | depth) in generation order.
// HBM traffic: K*8 B written by emit, K*8 B read + K*4 B written by sort, against
// 6 passes * 2 * 12 B * K for the reference's Onesweep sort (profiles/bicycle_2048.txt:7).
//
// The reference-shaped primitives (gs_tile_count, gs_full_cumsum_i32, gs_tile_emit_keys,
// gs_find_ranges; gs_radix_sort_pairs lives in radix_sort.hip) run the reference's own stage
// sequence and are used to cross-check the fused path.

#include "gs_common.h"
#include "../../include/gs_detmath.h"
#include "project_math.h"

namespace {

struct GridQuery {
  float ib00, ib01, ib10, ib11;
  float rel_min_x, rel_min_y;
  int min_tx, min_ty, span_x, span_y;
};

// taichi_lib/grid_query.py:73-91 (obb_grid_query) + :10-27 (tile_ranges)
__device__ __forceinline__ GridQuery grid_query(const float* g, int Wp, int Hp, int tile_size, float alpha_thr) {
  GridQuery q;
  const float mx = g[0], my = g[1], ax = g[2], ay = g[3], sgx = g[4], sgy = g[5], alpha = g[6];
  if (!(alpha > alpha_thr)) {  // explicit cull; the reference yields NaN bounds here (SURVEY 8a')
    q.ib00 = q.ib01 = q.ib10 = q.ib11 = q.rel_min_x = q.rel_min_y = 0.f;
    q.min_tx = q.min_ty = q.span_x = q.span_y = 0;
    return q;
  }
  const float gscale = gs_det_sqrtf(2.0f * gs_det_logf(__fdiv_rn(alpha, alpha_thr)));
  const float sx = sgx * gscale, sy = sgy * gscale;
  const float a2x = -ay, a2y = ax;
  const float v1x = ax * sx, v1y = ay * sx, v2x = a2x * sy, v2y = a2y * sy;
  const float ex = gs_det_sqrtf(v1x * v1x + v2x * v2x), ey = gs_det_sqrtf(v1y * v1y + v2y * v2y);
  const float lox = mx - ex, loy = my - ey, hix = mx + ex, hiy = my + ey;
  q.ib00 = __fdiv_rn(ax, sx); q.ib01 = __fdiv_rn(ay, sx); q.ib10 = __fdiv_rn(a2x, sy); q.ib11 = __fdiv_rn(a2y, sy);
  const float ts = float(tile_size);
  const int max_tx = (Wp - 1) / tile_size, max_ty = (Hp - 1) / tile_size;
  int min_tx = int(floorf(__fdiv_rn(lox, ts))), min_ty = int(floorf(__fdiv_rn(loy, ts)));
  min_tx = max(min_tx, 0); min_ty = max(min_ty, 0);
  int hi_tx = int(ceilf(__fdiv_rn(hix, ts))), hi_ty = int(ceilf(__fdiv_rn(hiy, ts)));
  hi_tx = min(max(hi_tx, min_tx + 1), max_tx + 1);
  hi_ty = min(max(hi_ty, min_ty + 1), max_ty + 1);
  q.min_tx = min_tx; q.min_ty = min_ty;
  q.span_x = max(hi_tx - min_tx, 0); q.span_y = max(hi_ty - min_ty, 0);
  q.rel_min_x = float(min_tx * tile_size) - mx;
  q.rel_min_y = float(min_ty * tile_size) - my;
  return q;
}

// taichi_lib/grid_query.py:30-43 (separates_bbox) / :58-61 (test_tile)
__device__ __forceinline__ bool test_tile(const GridQuery& q, int u, int v, int tile_size) {
  const float lx = q.rel_min_x + float(u * tile_size), ly = q.rel_min_y + float(v * tile_size);
  const float ux = lx + float(tile_size), uy = ly + float(tile_size);
  {
    const float t0 = q.ib00 * lx + q.ib01 * ly, t1 = q.ib00 * ux + q.ib01 * ly;
    const float t2 = q.ib00 * ux + q.ib01 * uy, t3 = q.ib00 * lx + q.ib01 * uy;
    const float mn = fminf(fminf(t0, t1), fminf(t2, t3)), mxv = fmaxf(fmaxf(t0, t1), fmaxf(t2, t3));
    if (mn > 1.0f || mxv < -1.0f) return false;
  }
  {
    const float t0 = q.ib10 * lx + q.ib11 * ly, t1 = q.ib10 * ux + q.ib11 * ly;
    const float t2 = q.ib10 * ux + q.ib11 * uy, t3 = q.ib10 * lx + q.ib11 * uy;
    const float mn = fminf(fminf(t0, t1), fminf(t2, t3)), mxv = fmaxf(fmaxf(t0, t1), fmaxf(t2, t3));
    if (mn > 1.0f || mxv < -1.0f) return false;
  }
  return true;
}

// mapper/tile_mapper.py:34-40 (32-bit depth) / :53-59 (16-bit depth): the depth part only
__device__ __forceinline__ uint32_t depth_key(float depth, bool depth16) {
  if (!depth16) return gs_f32_bits(depth);
  const float d = depth < 0.f ? 0.f : (depth > 1.f ? 1.f : depth);
  return uint32_t(d * 65535.0f);
}

struct MapArgs {
  const float* points;
  const float* depth;
  int64_t v;            // number of Gaussians, or the buffer capacity when v_dev is set
  const int* v_dev;     // optional: the actual count lives on the device (no host read-back)
  int Wp, Hp, tile_size, tiles_wide;
  float thr;
  int depth16;
  GsShard sh;           // owned tile rows (the whole image when the call is not sharded)
};

__device__ __forceinline__ bool any_owned_row(const GsShard& sh, int lo, int hi) { return gs_shard_any_row(sh, lo, hi); }
// local tile id of an owned tile
__device__ __forceinline__ int local_tile(const MapArgs& a, int gx, int gy) {
  return gs_shard_local_row(a.sh, gy) * a.tiles_wide + gx;
}

__device__ __forceinline__ int64_t live_count(const MapArgs& a) {
  if (a.v_dev == nullptr) return a.v;
  const int64_t d = *a.v_dev;
  return d < a.v ? d : a.v;
}

// ---- fused path -------------------------------------------------------------------------
// inclusive scan over the wave: row_shr 1/2/4/8 inside each 16-lane row, then row_bcast:15 / :31 across rows
__device__ __forceinline__ int wave_inclusive_scan(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
  return x;
}

// exclusive prefix of x over the 1024 threads of the workgroup (s_wave: 16 ints of LDS); total -> sum
__device__ __forceinline__ int block_exclusive_scan(int x, int* s_wave, int& sum) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int inc = wave_inclusive_scan(x);
  if (lane == 63) s_wave[wv] = inc;
  __syncthreads();
  int before = 0;
  sum = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const int tot = s_wave[w];
    sum += tot;
    before += w < wv ? tot : 0;
  }
  __syncthreads();  // s_wave is reused by the caller's next round
  return before + inc - x;
}

// One workgroup: exclusive scan of the T-entry tile histogram (coalesced rounds of 1024 tiles) ->
// tile_ranges, bucket cursors, K, fullest tile, overflow flag; beside it a second workgroup builds the rasterizer's
// launch order: tiles by descending population (counting sort on min(count, 1023); the order inside a
// bin comes from LDS atomics and is arbitrary -- it only affects scheduling).
__global__ __launch_bounds__(1024) void map_scan_kernel(int num_tiles, const int* tile_hist, int2* tile_ranges,
                                                        int* cursors, int* counts_out, int64_t k_capacity,
                                                        int* tile_order, const int* v_dev, int* counts_host,
                                                        const int* touched_dev) {
  // workgroup 0: tile ranges, cursors, K / fullest tile / overflow; workgroup 1 (launched only with a tile_order):
  // the launch order and the heavy-tile count.  Both read the same histogram and neither waits for the other.
  __shared__ int s_wave[16];
  __shared__ int s_bin[1024];
  const int t = threadIdx.x;
  if (blockIdx.x == 0) {
    // Round 3: every thread owns PER consecutive tiles of a round (64 bytes: four 16-byte loads), scans them serially
    // in registers and takes part in ONE workgroup scan of the per-thread totals per 16 384 tiles -- two barriers per
    // round instead of the 32 of a scan per 1024 tiles (14.5 -> ~5 us at 16 384 tiles).  Ranges and cursors leave as
    // 16-byte stores.
    constexpr int PER = 16;
    int carry = 0, mx = 0;
    for (int base0 = 0; base0 < num_tiles; base0 += 1024 * PER) {
      const int i0 = base0 + t * PER;
      int c[PER];
      const bool whole = i0 + PER <= num_tiles;
      if (whole) {
        const int4* src = reinterpret_cast<const int4*>(tile_hist + i0);
#pragma unroll
        for (int q = 0; q < PER / 4; ++q) {
          const int4 v4 = src[q];
          c[4 * q] = v4.x; c[4 * q + 1] = v4.y; c[4 * q + 2] = v4.z; c[4 * q + 3] = v4.w;
        }
      } else {
#pragma unroll
        for (int j = 0; j < PER; ++j) c[j] = i0 + j < num_tiles ? tile_hist[i0 + j] : 0;
      }
      int sum = 0;
#pragma unroll
      for (int j = 0; j < PER; ++j) { sum += c[j]; mx = max(mx, c[j]); }
      int total;
      int run = carry + block_exclusive_scan(sum, s_wave, total);
      carry += total;
      int2 rg[PER];
      int cur[PER];
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        // a tile that would run past the caller's pair capacity is dropped (and flagged below): the
        // caller re-runs with a larger buffer; nothing downstream may index past k_capacity
        const bool fits = k_capacity <= 0 || int64_t(run) + c[j] <= k_capacity;
        rg[j] = (c[j] > 0 && fits) ? make_int2(run, run + c[j]) : make_int2(0, 0);  // tile_mapper.py:186
        cur[j] = fits ? run : -(1 << 30);  // negative for the whole launch: any returning add reports it
        run += c[j];
      }
      if (whole) {
        int4* dr = reinterpret_cast<int4*>(tile_ranges + i0);
        int4* dc = reinterpret_cast<int4*>(cursors + i0);
#pragma unroll
        for (int q = 0; q < PER / 2; ++q) dr[q] = make_int4(rg[2 * q].x, rg[2 * q].y, rg[2 * q + 1].x, rg[2 * q + 1].y);
#pragma unroll
        for (int q = 0; q < PER / 4; ++q) dc[q] = make_int4(cur[4 * q], cur[4 * q + 1], cur[4 * q + 2], cur[4 * q + 3]);
      } else {
#pragma unroll
        for (int j = 0; j < PER; ++j)
          if (i0 + j < num_tiles) { tile_ranges[i0 + j] = rg[j]; cursors[i0 + j] = cur[j]; }
      }
    }
    // fullest tile
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off));
    if ((t & 63) == 0) s_wave[t >> 6] = mx;
    __syncthreads();
    if (t == 0) {
      int m = 0;
      for (int w = 0; w < 16; ++w) m = max(m, s_wave[w]);
      const int over = (k_capacity > 0 && int64_t(carry) > k_capacity) ? 1 : 0;
      counts_out[0] = carry;
      counts_out[1] = m;
      counts_out[2] = over;
      if (!tile_order) counts_out[3] = 0;
      if (counts_host) {  // pinned host words, visible to the host once this kernel has completed: no copy launch
        counts_host[0] = carry;
        counts_host[1] = m;
        counts_host[2] = over;
        if (!tile_order) counts_host[3] = 0;
        counts_host[4] = v_dev ? *v_dev : 0;
        counts_host[5] = touched_dev ? *touched_dev : 0;  // Gaussians in the region order = splats that can reach an owned row
      }
    }
    return;
  }
  // ---- workgroup 1: counting sort of the tiles by descending overlap count (1024 bins, the last one open-ended).
  // One pass over the histogram: a thread keeps its PER consecutive counts of a round in registers between the
  // binning and the scatter (rounds beyond the first -- more than 16 384 tiles -- read the histogram twice).
  constexpr int PER = 16;
  s_bin[t] = 0;
  __syncthreads();
  int sum = 0;
  int c0[PER];
  for (int base0 = 0; base0 < num_tiles; base0 += 1024 * PER) {
    const int i0 = base0 + t * PER;
    int c[PER];
    if (i0 + PER <= num_tiles) {
      const int4* src = reinterpret_cast<const int4*>(tile_hist + i0);
#pragma unroll
      for (int q = 0; q < PER / 4; ++q) {
        const int4 v4 = src[q];
        c[4 * q] = v4.x; c[4 * q + 1] = v4.y; c[4 * q + 2] = v4.z; c[4 * q + 3] = v4.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < PER; ++j) c[j] = i0 + j < num_tiles ? tile_hist[i0 + j] : -1;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      if (base0 == 0) c0[j] = c[j];
      if (c[j] >= 0) {
        sum += c[j];
        atomicAdd(&s_bin[min(c[j], 1023)], 1);
      }
    }
  }
  int unused, k_total;
  block_exclusive_scan(sum, s_wave, k_total);
  // start of bin b in descending order = number of tiles in fuller bins: scan the bins from the top
  const int start = block_exclusive_scan(s_bin[1023 - t], s_wave, unused);
  s_bin[1023 - t] = start;  // read above and written here by the same thread only
  // counts_out[3] = number of "heavy" tiles at the head of the order, which the rasterizer splits into four
  // 8x8 workgroups each.  A launch cannot end before ONE wave has walked its fullest tile (n splats take about
  // 1.75 n c when the wave has a SIMD to itself), while the whole launch takes about K c / 1024 on 1024 SIMDs:
  // tiles with n > K / 1792 are the ones that bound it.  Large grids (K / 1792 above every tile) split nothing.
  {
    const int thr_bin = min(max(k_total / 1792, 96), 1022);
    if (1023 - t == thr_bin) {
      const int heavy = min(start, num_tiles / 4);  // tiles in bins above thr_bin
      counts_out[3] = heavy;
      if (counts_host) counts_host[3] = heavy;
    }
  }
  __syncthreads();
  for (int base0 = 0; base0 < num_tiles; base0 += 1024 * PER) {
    const int i0 = base0 + t * PER;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int c = base0 == 0 ? c0[j] : (i0 + j < num_tiles ? tile_hist[i0 + j] : -1);
      if (c >= 0 && i0 + j < num_tiles) tile_order[atomicAdd(&s_bin[min(c, 1023)], 1)] = i0 + j;
    }
  }
}

// ---- region-binned counting / bucketing --------------------------------------------------
// Scattered 4-byte global atomics run at only ~20 G/s on MI355X (every one is its own 64-B
// memory-side request), which made the plain count / emit kernels above atomic-bound (K atomics
// each).  The binned path first orders the Gaussians by the screen REGION of their centre, then lets
// every workgroup -- whose CHUNK Gaussians now share one region -- count its overlaps in an LDS window
// covering the region plus a border, and touch global memory once per window tile.
// The region edge is the smallest power of two >= RG_MIN that keeps the region count <= MAX_REGIONS: the
// smaller the region, the more of a workgroup's overlaps share a window tile, i.e. the fewer global atomics
// (8x8-tile regions: ~5 overlaps per atomic at 256 Gaussians per workgroup; 32x32: ~1.4).
constexpr int RG_MIN = 8;              // smallest region edge in tiles
constexpr int RB = 4;                  // window border in tiles (splats reaching further fall back to global atomics)
constexpr int MAX_REGIONS = 1024;
constexpr int WIDE_SPAN = 64;          // candidate tiles above which a splat is walked by its whole wave
#ifndef GS_MAP_CHUNK
#define GS_MAP_CHUNK 512
#endif
constexpr int CHUNK = GS_MAP_CHUNK;     // Gaussians (= threads) per counting / bucketing workgroup

struct RegionGrid {
  int tiles_x, tiles_y, regions_x, num_regions;
  int rg;    // region edge in tiles
  int win;   // window edge = rg + 2 RB; the LDS window holds win * win ints
  int row0;  // first tile row the grid covers (a contiguous shard's first row; 0 otherwise)
};

// Query cache (round 3): the counting pass leaves the outcome of its OBB query and per-tile tests as 16 bytes per
// Gaussian, in the REGION order it works in: the accepted tiles of the candidate span as a 64-bit mask (bit = ty *
// span_x + tx, owned rows only) and the span itself.  The bucketing pass, which walks the same order, reads those 16
// bytes sequentially instead of gathering the 28-byte row again and repeating the query and the tests (emit 48 -> 34 us
// at C3).  Spans above WIDE_SPAN tiles are flagged and walked by their whole wave from the row, as before.
// (Filling the cache in the binning pass instead, in index order, made that pass 23 us slower and the two consumers
// gather it at random: 7 us worse in all.)
struct QueryCache {
  unsigned long long accept;
  unsigned int x;  // min_tx (20 bits: up to 2^20 tiles) | span_x << 20
  unsigned int y;  // min_ty (20 bits) | span_y << 20 | wide << 31
};
constexpr unsigned QC_WIDE = 0x80000000u;
__device__ __forceinline__ int qc_min_tx(const QueryCache& q) { return int(q.x & 0xfffffu); }
__device__ __forceinline__ int qc_min_ty(const QueryCache& q) { return int(q.y & 0xfffffu); }
__device__ __forceinline__ int qc_span_x(const QueryCache& q) { return int((q.x >> 20) & 0x7fu); }

// Does the Gaussian enter the region order at all?  Leaving one out is only allowed when it has no tile here: the exact
// query decides for a sharded frame (most splats miss a rank's rows, and the sparse exchange lists exactly the ones that
// do not).  For a whole image nearly every visible Gaussian has a tile, and one that has none simply contributes
// nothing to the counting pass: the query is skipped (the answer may be conservative, never the other way round).
__device__ __forceinline__ bool enters_order(const float* g, const MapArgs& a) {
  if (a.sh.period == 1 && a.sh.begin == 0 && a.sh.end * a.tile_size >= a.Hp) return true;
  const GridQuery q = grid_query(g, a.Wp, a.Hp, a.tile_size, a.thr);
  return q.span_x > 0 && any_owned_row(a.sh, q.min_ty, q.min_ty + q.span_y);
}

__device__ __forceinline__ int region_of_gaussian(const float* g, const MapArgs& a, const RegionGrid& rg) {
  const float ts = float(a.tile_size);
  int tx = int(floorf(g[0] / ts)), ty = int(floorf(g[1] / ts));
  tx = min(max(tx, 0), rg.tiles_x - 1);
  ty = min(max(ty - rg.row0, 0), rg.tiles_y - 1);  // the region grid starts at the shard's first row
  return (ty / rg.rg) * rg.regions_x + (tx / rg.rg);
}

// K1: per-workgroup region populations.  A workgroup reserves its place inside every region it holds Gaussians of with
// ONE returning atomic per (workgroup, region) on the region's counter and keeps the offset in part[region][workgroup]
// (round 3; ~250 atomics per workgroup of 1 024 Gaussians at C3, spread over 256 counters).  The order of the
// workgroups inside a region is the order their atomics arrive in -- it only decides which counting workgroup a
// Gaussian lands in, never a result: every tile's bucket is sorted on (depth, index) afterwards.  (Rounds 1 - 2 kept
// the order deterministic with two scan launches over the part matrix: 5 + 5 us at any size.)
// region_count must be zero when the pass starts (gs_map_bin_counters).
constexpr int BIN = 1024;
// region_count == nullptr: the scan launches follow (K2a / K2b); part[r][b] is the workgroup's own count
__device__ __forceinline__ void publish_region_hist(const int* s_hist, const RegionGrid& rg, int num_wg, int blk,
                                                    int* part, int* region_count) {
  for (int r = threadIdx.x; r < rg.num_regions; r += BIN) {
    const int h = s_hist[r];
    if (region_count == nullptr) part[int64_t(r) * num_wg + blk] = h;
    else if (h > 0) part[int64_t(r) * num_wg + blk] = atomicAdd(region_count + r, h);
  }
}
// The rows of workgroup b are [block_start[b], block_start[b + 1]): b * BIN .. here; the ranges the projection's
// compaction pass produced when that pass did the binning itself (compact_bin_kernel below).
__global__ __launch_bounds__(BIN) void region_count_kernel(MapArgs a, RegionGrid rg, int num_wg, int* region_of,
                                                           int* part, int* touched_blocks, int* block_start,
                                                           int* region_count) {
  __shared__ int s_hist[MAX_REGIONS];
  for (int r = threadIdx.x; r < rg.num_regions; r += BIN) s_hist[r] = 0;
  __syncthreads();
  const int64_t i = int64_t(blockIdx.x) * BIN + threadIdx.x;
  bool mine = false;
  const int64_t live = live_count(a);
  if (threadIdx.x == 0) {
    block_start[blockIdx.x] = int(min(int64_t(blockIdx.x) * BIN, live));
    if (int(blockIdx.x) == num_wg - 1) block_start[num_wg] = int(live);
  }
  if (i < live) {
    // Gaussians whose candidate span is empty (off-screen within the cull margin; above or below this rank's
    // strip when the frame is sharded) are left out of the ordering, so the counting and bucketing passes never
    // see them.  (A non-empty span whose tiles all fail the OBB test is rare and simply contributes nothing.)
    const bool any = enters_order(a.points + 7 * i, a);
    const int r = any ? region_of_gaussian(a.points + 7 * i, a, rg) : -1;
    region_of[i] = r;
    if (r >= 0) atomicAdd(&s_hist[r], 1);
    mine = r >= 0;
  }
  // how many of this workgroup's rows are in the ordering at all (gs_map_touched_list compacts them in ascending order)
  const int touched = __syncthreads_count(mine);
  if (threadIdx.x == 0) touched_blocks[blockIdx.x] = touched;
  publish_region_hist(s_hist, rg, num_wg, blockIdx.x, part, region_count);
}

// Two ways from the per-workgroup histograms to "where does workgroup b start inside region r" (part[r][b]), chosen by
// bin_with_atomics() from the sizes alone:
//  * few (workgroup, region) pairs (small scenes, a rank's strip): publish_region_hist above -- no launch of its own, and
//    K3 forms the region starts itself.  At C2 the two scan launches it replaces cost 10 of the mapper's 80 us;
//  * many (C3: 977 workgroups x 256 regions = 250 k pairs): returning atomics on 256 counters run at ~30 G/s and cost the
//    binning pass 8 us, more than the scan launches (K2a + K2b, 5 + 5 us at any size) they save.
// K2a: one workgroup per region: exclusive scan of part[region][*] in place, total -> region_count.
__global__ __launch_bounds__(1024) void region_part_scan_kernel(int num_wg, int* part, int* region_count,
                                                                int) {
  __shared__ int s_wave[16];
  int* row = part + int64_t(blockIdx.x) * num_wg;
  int carry = 0;
  for (int base = 0; base < num_wg; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < num_wg ? row[i] : 0;
    int total;
    const int before = block_exclusive_scan(v, s_wave, total);
    if (i < num_wg) row[i] = carry + before;
    carry += total;
  }
  if (threadIdx.x == 0) region_count[blockIdx.x] = carry;
}

// K2b: exclusive scan of the region populations -> start of each region in the ordered list, and of
// the per-region chunk counts (a chunk = up to CHUNK Gaussians of ONE region = one workgroup later on).
__global__ __launch_bounds__(1024) void region_scan_kernel(int num_regions, const int* region_count, int* region_start,
                                                           int* chunk_start) {
  __shared__ int s_wave[16];
  const int r = threadIdx.x;  // num_regions <= MAX_REGIONS = 1024
  const int c = r < num_regions ? region_count[r] : 0;
  const int ch = (c + CHUNK - 1) / CHUNK;
  int total_c, total_ch;
  const int start = block_exclusive_scan(c, s_wave, total_c);
  const int chunk = block_exclusive_scan(ch, s_wave, total_ch);
  if (r < num_regions) {
    region_start[r] = start;
    chunk_start[r] = chunk;
  }
  if (r == 0) {
    region_start[num_regions] = total_c;
    chunk_start[num_regions] = total_ch;
  }
}

// K3: write the Gaussian indices grouped by region: position = region start + this workgroup's offset inside the region
// (K1's atomic) + rank inside the workgroup (LDS atomic).  The region starts are the exclusive scan of the <= 1 024 region
// counters, which every workgroup forms for itself in LDS (4 KB out of L2) instead of reading it from a scan launch;
// workgroup 0 also leaves it -- and the per-region chunk starts (a chunk = up to CHUNK Gaussians of ONE region = one
// workgroup of the counting / bucketing passes) -- in global memory for those passes, and the whole grid clears the tile
// histogram the counting pass adds into.
__global__ __launch_bounds__(BIN) void region_scatter_kernel(RegionGrid rg, int num_wg, const int* region_of,
                                                             const int* part, const int* region_count,
                                                             int* region_start, int* chunk_start,
                                                             const int* block_start, int* order, int* tile_hist,
                                                             int num_tiles, int scanned) {
  __shared__ int s_cnt[MAX_REGIONS];
  __shared__ int s_start[MAX_REGIONS];
  __shared__ int s_wave[16];
  for (int i = blockIdx.x * BIN + threadIdx.x; i < num_tiles; i += gridDim.x * BIN) tile_hist[i] = 0;
  const int t = threadIdx.x;  // num_regions <= MAX_REGIONS = BIN
  s_cnt[t] = 0;
  if (scanned) {  // K2a / K2b have run: the starts are in global memory
    s_start[t] = t < rg.num_regions ? region_start[t] : 0;
  } else {
    const int c = t < rg.num_regions ? region_count[t] : 0;
    int total_c;
    const int start = block_exclusive_scan(c, s_wave, total_c);
    s_start[t] = start;
    if (blockIdx.x == 0) {
      int total_ch;
      const int chunk = block_exclusive_scan((c + CHUNK - 1) / CHUNK, s_wave, total_ch);
      if (t < rg.num_regions) {
        region_start[t] = start;
        chunk_start[t] = chunk;
      }
      if (t == 0) {
        region_start[rg.num_regions] = total_c;
        chunk_start[rg.num_regions] = total_ch;
      }
    }
  }
  __syncthreads();
  const int first = block_start[blockIdx.x];
  const int i = first + t;
  if (i < block_start[blockIdx.x + 1]) {
    const int r = region_of[i];
    if (r >= 0) {
      const int local = atomicAdd(&s_cnt[r], 1);
      order[s_start[r] + part[int64_t(r) * num_wg + blockIdx.x] + local] = i;
    }
  }
}

// The projection's stable compaction (project.hip: compact_kernel, whose outputs these are, bit for bit) with K1 folded
// in: a workgroup of BIN = 1024 staged rows writes its visible ones -- one contiguous range of compact rows -- and, while
// it still holds them in registers, runs K1's query on them.  Frame calls only (gs_project_fwd_ex with a GsMapBinPlan).
__global__ __launch_bounds__(BIN) void compact_bin_kernel(GsCompactArgs c, MapArgs a, RegionGrid rg, int num_wg,
                                                          int* region_of, int* part, int* touched_blocks,
                                                          int* block_start, int* region_count) {
  __shared__ int s_hist[MAX_REGIONS];
  __shared__ int s_cnt[16];
  __shared__ int s_before[16];
  for (int r = threadIdx.x; r < rg.num_regions; r += BIN) s_hist[r] = 0;
  const int64_t i = int64_t(blockIdx.x) * BIN + threadIdx.x;
  const float4* st_rows = static_cast<const float4*>(c.st_rows);
  float4 r0 = make_float4(0, 0, 0, 0), r1 = r0;
  bool vis = false;
  if (i < c.n) {
    r0 = st_rows[2 * i];
    r1 = st_rows[2 * i + 1];
    vis = r1.w != 0.0f;
  }
  const uint64_t b = __ballot(vis);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) s_cnt[wave] = __popcll(b);
  const int first_small = int(blockIdx.x) * (BIN / 256);  // the projection pass counted per 256 Gaussians
  int before = 0;
  if (!c.block_offsets) {
    for (int j = threadIdx.x; j < first_small; j += BIN) before += c.block_counts[j];
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off);
    if (lane == 0) s_before[wave] = before;
  }
  __syncthreads();
  int base = 0;
  if (c.block_offsets) {
    base = first_small < c.num_blocks ? c.block_offsets[first_small] : 0;
  } else {
#pragma unroll
    for (int w = 0; w < 16; ++w) base += s_before[w];
  }
  const int first = base;
  int mine_total = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    base += w < wave ? s_cnt[w] : 0;
    mine_total += s_cnt[w];
  }
  bool binned = false;
  if (i < c.n) {
    int slot = -1;
    if (vis) {
      slot = base + __popcll(b & ((1ull << lane) - 1ull));
      float* p = c.points + int64_t(slot) * 7;
      p[0] = r0.x; p[1] = r0.y; p[2] = r0.z; p[3] = r0.w; p[4] = r1.x; p[5] = r1.y; p[6] = r1.z;
      c.depth[slot] = r1.w;
      if (c.depth_feat) {
        c.depth_feat[int64_t(slot) * c.depth_feat_stride] = r1.w;
        c.depth_feat[int64_t(slot) * c.depth_feat_stride + 1] = r1.w * r1.w;
      }
      const float inv_d = __fdiv_rn(1.0f, r1.w);  // the sort key: fixed f32 op order
      c.ndc[slot] = 1.0f - __fdiv_rn(inv_d - c.inv_far, c.ndc_denom);
      c.indexes[slot] = i;
      // K1 on the row in registers
      const float g[7] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z};
      const bool any = enters_order(g, a);
      const int r = any ? region_of_gaussian(g, a, rg) : -1;
      region_of[slot] = r;
      if (r >= 0) atomicAdd(&s_hist[r], 1);
      binned = r >= 0;
    }
    c.slot_of[i] = slot;
  }
  const int touched = __syncthreads_count(binned);
  if (threadIdx.x == 0) {
    touched_blocks[blockIdx.x] = touched;
    block_start[blockIdx.x] = first;
    if (int(blockIdx.x) == num_wg - 1) {
      block_start[num_wg] = first + mine_total;
      *c.num_visible = first + mine_total;
    }
  }
  // the returning atomics of publish_region_hist are issued first and their results stored last: the zero fill of the
  // gradient rows -- a third of this pass's traffic -- runs underneath their latency (num_regions <= BIN: one region
  // per thread)
  const int my_hist = int(threadIdx.x) < rg.num_regions ? s_hist[threadIdx.x] : 0;
  int my_offset = my_hist;
  if (region_count != nullptr && my_hist > 0) my_offset = atomicAdd(region_count + threadIdx.x, my_hist);
  if (c.zero_rows) {  // the frame's gradient rows, cleared by the pass that streams the V rows anyway (project.hip)
    float4* dst = static_cast<float4*>(c.zero_rows) + int64_t(first) * c.zero_row_v4;
    for (int e = threadIdx.x; e < mine_total * c.zero_row_v4; e += BIN) dst[e] = make_float4(0, 0, 0, 0);
  }
  if (int(threadIdx.x) < rg.num_regions && (region_count == nullptr || my_hist > 0))
    part[int64_t(threadIdx.x) * num_wg + blockIdx.x] = my_offset;
}

// One pass: projection + cull + stable compaction + K1 (experiment, GS_PROJECT_ONE_PASS).  The visible
// counts of the workgroups in front are not read from a counting pass but found by a decoupled look-back over one
// 64-bit descriptor per workgroup -- flag << 32 | count, flag 1 = this workgroup's own count ("aggregate"), 2 = the count
// of everything up to and including it ("prefix") -- written and polled with agent-scope atomics (they bypass the
// XCD-local L2).  Workgroup ids are tickets drawn from an atomic counter, so every predecessor a workgroup polls has
// started before it: resident workgroups only ever wait for resident or finished ones.  The polling is bounded: a
// workgroup that does not see its predecessors after LOOKBACK_POLLS rounds gives up, sets the failure word and writes
// nothing (the experiment does not surface that word to the caller; a shipped version would poison V with it).
// Measured at C3 (profiles/r3/ab_project_one_pass.txt): bit-identical outputs, gs_project_fwd 56 us against 48 us for
// project_kernel + compact_bin_kernel -- the first cohort of resident workgroups resolves its look-back in ~8 serial
// windows of 64 while nothing is written.  Variant 2 (count, then project again: ab_project_count_then_recompute.txt):
// 64 us -- the projection's arithmetic (IEEE divisions, square roots, exp, the exact log) is not free next to its 44
// bytes.  Neither is used.
constexpr int LOOKBACK_POLLS = 1 << 18;
constexpr unsigned long long LB_AGGREGATE = 1ull << 32, LB_PREFIX = 2ull << 32;

// (variant 2, LOOKBACK = false: a counting pass -- the projection, nothing written but one count per workgroup -- in
// front, and this kernel adds up the counts of the workgroups before it: the projection is evaluated twice, the 32-byte
// staging rows are neither written nor read.)
__global__ __launch_bounds__(BIN) void project_count_kernel(gs_proj::ProjArgs pa, int* counts) {
  __shared__ int s_cnt[16];
  const int64_t i = int64_t(blockIdx.x) * BIN + threadIdx.x;
  bool vis = false;
  if (i < pa.n) {
    const gs_proj::Cam cam = gs_proj::load_cam(pa.T44, pa.proj);
    gs_proj::Fwd f;
    gs_proj::forward(pa, cam, i, f);
    vis = gs_proj::visible(pa, f);
  }
  const uint64_t b = __ballot(vis);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < 16; ++w) t += s_cnt[w];
    counts[blockIdx.x] = t;
  }
}

template <bool LOOKBACK>
__global__ __launch_bounds__(BIN) void project_compact_bin_kernel(gs_proj::ProjArgs pa, GsCompactArgs c, MapArgs a,
                                                                  RegionGrid rg, int num_wg, int* region_of, int* part,
                                                                  int* touched_blocks, int* block_start,
                                                                  int* region_count, unsigned long long* desc,
                                                                  int* ticket, int* failed, float* cam_out) {
  __shared__ int s_hist[MAX_REGIONS];
  __shared__ int s_cnt[16];
  __shared__ int s_scalar[2];
  __shared__ int s_before[16];
  if (LOOKBACK && threadIdx.x == 0) s_scalar[0] = atomicAdd(ticket, 1);
  for (int r = threadIdx.x; r < rg.num_regions; r += BIN) s_hist[r] = 0;
  __syncthreads();
  const int blk = LOOKBACK ? s_scalar[0] : int(blockIdx.x);
  if (blk >= num_wg) return;  // (cannot happen: the grid is num_wg workgroups)
  if (cam_out && blk == 0 && threadIdx.x == 0) gs_proj::camera_position(pa.T44, cam_out);
  const int64_t i = int64_t(blk) * BIN + threadIdx.x;
  gs_proj::Fwd f;
  bool vis = false;
  if (i < pa.n) {
    const gs_proj::Cam cam = gs_proj::load_cam(pa.T44, pa.proj);
    gs_proj::forward(pa, cam, i, f);
    vis = gs_proj::visible(pa, f);
  }
  const uint64_t b = __ballot(vis);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) s_cnt[wave] = __popcll(b);
  if (!LOOKBACK) {
    int before = 0;
    for (int j = threadIdx.x; j < blk; j += BIN) before += c.block_counts[j];
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off);
    if (lane == 0) s_before[wave] = before;
  }
  __syncthreads();
  int mine_total = 0, base = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    base += w < wave ? s_cnt[w] : 0;
    mine_total += s_cnt[w];
  }
  if (!LOOKBACK && threadIdx.x == 0) {
    int before = 0;
    for (int w = 0; w < 16; ++w) before += s_before[w];
    s_scalar[0] = before;
    s_scalar[1] = 1;
  }
  if (LOOKBACK && wave == 0) {
    if (lane == 0)
      __hip_atomic_store(desc + blk, (blk == 0 ? LB_PREFIX : LB_AGGREGATE) | (unsigned long long)(unsigned)mine_total,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int before = 0;
    bool ok = true;
    for (int look = blk - 1; look >= 0; look -= 64) {
      const int j = look - lane;  // lane 0 = the nearest predecessor
      unsigned long long d = LB_PREFIX;  // in front of workgroup 0: nothing
      int polls = 0;
      if (j >= 0) d = __hip_atomic_load(desc + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__ballot((d >> 32) == 0ull) != 0ull) {
        if (++polls > LOOKBACK_POLLS) { ok = false; break; }
        __builtin_amdgcn_s_sleep(2);
        if ((d >> 32) == 0ull) d = __hip_atomic_load(desc + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (!ok) break;
      const uint64_t prefixes = __ballot((d >> 32) == 2ull);
      const int first = prefixes != 0ull ? __ffsll(static_cast<unsigned long long>(prefixes)) - 1 : 63;
      int v = lane <= first ? int(unsigned(d)) : 0;
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
      before += v;
      if (prefixes != 0ull) break;
    }
    if (lane == 0) {
      if (ok && blk > 0)
        __hip_atomic_store(desc + blk, LB_PREFIX | (unsigned long long)(unsigned)(before + mine_total), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      if (!ok) {  // successors must not wait for this one for ever either
        __hip_atomic_store(desc + blk, LB_PREFIX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *failed = 1;
      }
      s_scalar[0] = before;
      s_scalar[1] = ok ? 1 : 0;
    }
  }
  __syncthreads();
  const int first = s_scalar[0];
  const bool ok = s_scalar[1] != 0;
  base += first;
  bool binned = false;
  if (i < pa.n && ok) {
    int slot = -1;
    if (vis && base + __popcll(b & ((1ull << lane) - 1ull)) < c.n) {  // (always, unless the two passes disagreed)
      slot = base + __popcll(b & ((1ull << lane) - 1ull));
      const float z = f.cam[2];
      float* p = c.points + int64_t(slot) * 7;
      p[0] = f.u; p[1] = f.v; p[2] = f.ax; p[3] = f.ay; p[4] = f.s1; p[5] = f.s2; p[6] = f.alpha;
      c.depth[slot] = z;
      if (c.depth_feat) {
        c.depth_feat[int64_t(slot) * c.depth_feat_stride] = z;
        c.depth_feat[int64_t(slot) * c.depth_feat_stride + 1] = z * z;
      }
      const float inv_d = __fdiv_rn(1.0f, z);
      c.ndc[slot] = 1.0f - __fdiv_rn(inv_d - c.inv_far, c.ndc_denom);
      c.indexes[slot] = i;
      const float g[7] = {f.u, f.v, f.ax, f.ay, f.s1, f.s2, f.alpha};
      const bool any = enters_order(g, a);
      const int r = any ? region_of_gaussian(g, a, rg) : -1;
      region_of[slot] = r;
      if (r >= 0) atomicAdd(&s_hist[r], 1);
      binned = r >= 0;
    }
    c.slot_of[i] = slot;
  }
  const int touched = __syncthreads_count(binned);
  if (threadIdx.x == 0) {
    touched_blocks[blk] = touched;
    block_start[blk] = first;
    if (blk == num_wg - 1) {
      block_start[num_wg] = first + mine_total;
      *c.num_visible = first + mine_total;
    }
  }
  publish_region_hist(s_hist, rg, num_wg, blk, part, region_count);
  if (c.zero_rows && ok) {
    float4* dst = static_cast<float4*>(c.zero_rows) + int64_t(first) * c.zero_row_v4;
    for (int e = threadIdx.x; e < mine_total * c.zero_row_v4; e += BIN) dst[e] = make_float4(0, 0, 0, 0);
  }
}

// which chunk of which region does this workgroup process?
__device__ __forceinline__ bool locate_chunk(int block, const RegionGrid& rg, const int* region_start,
                                             const int* chunk_start, int& region, int& first, int& count) {
  __shared__ int s_loc[3];
  if (threadIdx.x == 0) {
    int r = -1;
    if (block < chunk_start[rg.num_regions]) {
      int lo = 0, hi = rg.num_regions;  // last r with chunk_start[r] <= block
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (chunk_start[mid] <= block) lo = mid; else hi = mid;
      }
      r = lo;
      const int c = block - chunk_start[r];
      s_loc[1] = region_start[r] + c * CHUNK;
      s_loc[2] = min(CHUNK, region_start[r + 1] - s_loc[1]);
    }
    s_loc[0] = r;
  }
  __syncthreads();
  region = s_loc[0];
  first = s_loc[1];
  count = s_loc[2];
  return region >= 0;
}

// K4: per-tile histogram through the LDS window.
__global__ __launch_bounds__(CHUNK) void count_binned_kernel(MapArgs a, RegionGrid rg, const int* order,
                                                           const int* region_start, const int* chunk_start,
                                                           int* tile_hist, QueryCache* qcache) {
  extern __shared__ int s_win[];  // win * win
  const int WIN = rg.win, WIN_TILES = WIN * WIN;
  int region, first, count;
  if (!locate_chunk(blockIdx.x, rg, region_start, chunk_start, region, first, count)) return;
  for (int e = threadIdx.x; e < WIN_TILES; e += CHUNK) s_win[e] = 0;
  __syncthreads();
  // the window is laid out in full-image tile coordinates; rows the shard does not own stay zero
  const int wx0 = (region % rg.regions_x) * rg.rg - RB, wy0 = (region / rg.regions_x) * rg.rg - RB + rg.row0;
  auto add_tile = [&](int gx, int gy) {
    if (!gs_shard_owns(a.sh, gy)) return;
    const int lx = gx - wx0, ly = gy - wy0;
    if (unsigned(lx) < unsigned(WIN) && unsigned(ly) < unsigned(WIN)) atomicAdd(&s_win[ly * WIN + lx], 1);
    else atomicAdd(tile_hist + local_tile(a, gx, gy), 1);
  };
  int i = 0;
  bool wide = false;
  if (int(threadIdx.x) < count) {
    i = order[first + threadIdx.x];
    const GridQuery q = grid_query(a.points + 7 * int64_t(i), a.Wp, a.Hp, a.tile_size, a.thr);
    wide = q.span_x * q.span_y > WIDE_SPAN;
    QueryCache qc;
    qc.accept = 0ull;
    qc.x = unsigned(q.min_tx) | (wide ? 0u : unsigned(q.span_x) << 20);
    qc.y = unsigned(q.min_ty) | (wide ? QC_WIDE : unsigned(q.span_y) << 20);
    if (!wide) {
      int bit = 0;
      for (int ty = 0; ty < q.span_y; ++ty)
        for (int tx = 0; tx < q.span_x; ++tx, ++bit)
          if (gs_shard_owns(a.sh, ty + q.min_ty) && test_tile(q, tx, ty, a.tile_size)) {
            qc.accept |= 1ull << bit;
            add_tile(tx + q.min_tx, ty + q.min_ty);
          }
    }
    // by position in the region order: the bucketing pass reads it back with consecutive 16-byte loads
    *reinterpret_cast<uint4*>(qcache + first + threadIdx.x) = *reinterpret_cast<const uint4*>(&qc);
  }
  // Splats with a wide candidate span (hundreds of tiles for a floater that covers the screen) are walked by
  // the whole wave, 64 tiles per step: one lane looping over them alone would hold its workgroup for
  // milliseconds (a dependent global atomic per tile).
  for (uint64_t todo = __ballot(wide); todo != 0ull; todo &= todo - 1ull) {
    const int gi = __shfl(i, __ffsll(static_cast<unsigned long long>(todo)) - 1);
    const GridQuery q = grid_query(a.points + 7 * int64_t(gi), a.Wp, a.Hp, a.tile_size, a.thr);
    const int total = q.span_x * q.span_y;
    for (int k = int(threadIdx.x & 63); k < total; k += 64) {
      const int ty = k / q.span_x, tx = k - ty * q.span_x;
      if (test_tile(q, tx, ty, a.tile_size)) add_tile(tx + q.min_tx, ty + q.min_ty);
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < WIN_TILES; e += CHUNK) {
    const int c = s_win[e];
    if (c > 0) {
      const int gx = wx0 + e % WIN, gy = wy0 + e / WIN;
      atomicAdd(tile_hist + local_tile(a, gx, gy), c);
    }
  }
}

// K6: bucket the (depth key, index) pairs.  Pass A counts into the LDS window, pass B reserves one
// contiguous range per window tile with a single returning global atomic, pass C places the pairs
// with LDS atomics.  The accepted-tile set of a lane is kept as a 64-bit mask between the passes
// (so the OBB tests run once); splats whose candidate span exceeds 64 tiles are walked by the whole wave.
__global__ __launch_bounds__(CHUNK) void emit_binned_kernel(MapArgs a, RegionGrid rg, const int* order,
                                                          const int* region_start, const int* chunk_start,
                                                          int* cursors, uint64_t* pairs, const QueryCache* qcache) {
  extern __shared__ int s_dyn[];  // 2 * win * win
  const int WIN = rg.win, WIN_TILES = WIN * WIN;
  int* s_cnt = s_dyn;
  int* s_base = s_dyn + WIN_TILES;
  int region, first, count;
  if (!locate_chunk(blockIdx.x, rg, region_start, chunk_start, region, first, count)) return;
  for (int e = threadIdx.x; e < WIN_TILES; e += CHUNK) s_cnt[e] = 0;
  __syncthreads();
  const int wx0 = (region % rg.regions_x) * rg.rg - RB, wy0 = (region / rg.regions_x) * rg.rg - RB + rg.row0;
  const bool active = int(threadIdx.x) < count;
  const int lane = int(threadIdx.x & 63);
  unsigned long long accept = 0ull;
  int min_tx = 0, min_ty = 0, span_x = 1;
  bool wide = false;  // candidate span above WIDE_SPAN tiles: walked by the whole wave in both passes
  int i = 0;
  if (active) {
    i = order[first + threadIdx.x];
    const uint4 raw = *reinterpret_cast<const uint4*>(qcache + first + threadIdx.x);
    const QueryCache qc = *reinterpret_cast<const QueryCache*>(&raw);
    wide = (qc.y & QC_WIDE) != 0u;
    if (!wide) {
      accept = qc.accept;
      min_tx = qc_min_tx(qc); min_ty = qc_min_ty(qc); span_x = qc_span_x(qc);
      for (unsigned long long todo = accept; todo != 0ull; todo &= todo - 1ull) {
        const int bit = __ffsll(todo) - 1;
        const int ty = bit / span_x, tx = bit - ty * span_x;
        const int lx = tx + min_tx - wx0, ly = ty + min_ty - wy0;
        if (unsigned(lx) < unsigned(WIN) && unsigned(ly) < unsigned(WIN)) atomicAdd(&s_cnt[ly * WIN + lx], 1);
      }
    }
  }
  const uint64_t wide_lanes = __ballot(wide);
  for (uint64_t todo = wide_lanes; todo != 0ull; todo &= todo - 1ull) {
    const int gi = __shfl(i, __ffsll(static_cast<unsigned long long>(todo)) - 1);
    const GridQuery w = grid_query(a.points + 7 * int64_t(gi), a.Wp, a.Hp, a.tile_size, a.thr);
    const int total = w.span_x * w.span_y;
    for (int k = lane; k < total; k += 64) {
      const int ty = k / w.span_x, tx = k - ty * w.span_x;
      if (!gs_shard_owns(a.sh, ty + w.min_ty) || !test_tile(w, tx, ty, a.tile_size)) continue;
      const int lx = tx + w.min_tx - wx0, ly = ty + w.min_ty - wy0;
      if (unsigned(lx) < unsigned(WIN) && unsigned(ly) < unsigned(WIN)) atomicAdd(&s_cnt[ly * WIN + lx], 1);
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < WIN_TILES; e += CHUNK) {
    const int c = s_cnt[e];
    if (c > 0) {
      const int gx = wx0 + e % WIN, gy = wy0 + e / WIN;
      // a tile dropped by the capacity clamp has a hugely negative cursor: the returned value says so
      // (no separate load in front of the atomic: the reservations of a workgroup must pipeline)
      const int base = atomicAdd(cursors + local_tile(a, gx, gy), c);
      s_base[e] = base < 0 ? -1 : base;
    }
    s_cnt[e] = 0;
  }
  __syncthreads();
  auto place = [&](int gx, int gy, uint64_t pair) {
    const int lx = gx - wx0, ly = gy - wy0;
    int slot;
    if (unsigned(lx) < unsigned(WIN) && unsigned(ly) < unsigned(WIN)) {
      const int e = ly * WIN + lx;
      if (s_base[e] < 0) return;
      slot = s_base[e] + atomicAdd(&s_cnt[e], 1);
    } else {
      slot = atomicAdd(cursors + local_tile(a, gx, gy), 1);
      if (slot < 0) return;
    }
    pairs[slot] = pair;
  };
  if (active && !wide) {
    const uint64_t pair = (uint64_t(depth_key(a.depth[i], a.depth16 != 0)) << 32) | uint64_t(uint32_t(i));
    for (unsigned long long todo = accept; todo != 0ull; todo &= todo - 1ull) {
      const int bit = __ffsll(todo) - 1;
      const int ty = bit / span_x, tx = bit - ty * span_x;
      place(tx + min_tx, ty + min_ty, pair);
    }
  }
  for (uint64_t todo = wide_lanes; todo != 0ull; todo &= todo - 1ull) {
    const int gi = __shfl(i, __ffsll(static_cast<unsigned long long>(todo)) - 1);
    const GridQuery w = grid_query(a.points + 7 * int64_t(gi), a.Wp, a.Hp, a.tile_size, a.thr);
    const uint64_t pair = (uint64_t(depth_key(a.depth[gi], a.depth16 != 0)) << 32) | uint64_t(uint32_t(gi));
    const int total = w.span_x * w.span_y;
    for (int k = lane; k < total; k += 64) {
      const int ty = k / w.span_x, tx = k - ty * w.span_x;
      if (gs_shard_owns(a.sh, ty + w.min_ty) && test_tile(w, tx, ty, a.tile_size))
        place(tx + w.min_tx, ty + w.min_ty, pair);
    }
  }
}

// One workgroup per tile.  n <= CAP: bitonic sort in LDS.  n > CAP: the same network in place in
// global memory (rare: more than CAP splats on one tile); a workgroup lives on one CU, so its own
// global writes are visible to it after __syncthreads().
// The network is the direction-free bitonic formulation: each merge of width k starts with a
// mirror step (partner i ^ (k-1)) and continues with partners i ^ j, j = k/4 .. 1; every
// compare-exchange puts the minimum at the lower index.  Elements at index >= n are virtual +inf:
// they never move, so no padding is stored and n need not be a power of two.
template <int THREADS>
__device__ __forceinline__ void bitonic_sort(uint64_t* data, int n, int t) {
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  for (int k = 2; k <= np2; k <<= 1) {
    for (int i = t; i < n; i += THREADS) {
      const int l = i ^ (k - 1);
      if (l > i && l < n) {
        const uint64_t x = data[i], y = data[l];
        if (x > y) { data[i] = y; data[l] = x; }
      }
    }
    __syncthreads();
    for (int j = k >> 2; j > 0; j >>= 1) {
      for (int i = t; i < n; i += THREADS) {
        const int l = i ^ j;
        if (l > i && l < n) {
          const uint64_t x = data[i], y = data[l];
          if (x > y) { data[i] = y; data[l] = x; }
        }
      }
      __syncthreads();
    }
  }
}

// Merge sort of one crowded bucket (n <= CAP keys) in LDS by a whole workgroup.  Chunks of 256 are sorted by one
// wave each the way the wave rank sort does it; then runs are merged pairwise, each key finding its place by a
// lower bound in the partner run (keys are unique), ping-pong between two LDS buffers: log2(n / 256) barriers,
// against 78 for the bitonic network at 4096 keys.
template <int THREADS, int CAP>
__device__ __forceinline__ uint64_t* lds_merge_sort(uint64_t* a, uint64_t* b, int n, int t) {
  const int np = (n + 255) & ~255;  // pad to whole chunks with unique keys above every real one
  for (int i = n + t; i < np; i += THREADS) a[i] = 0xFFFFFFFF00000000ull | uint64_t(i);
  __syncthreads();
  // chunks of 256: one wave each, the two-level rank sort of rank_sort_rows<4> (rows of 64 against themselves,
  // lower bounds across the four rows), a -> b.  Only the wave's own chunk is touched, so no workgroup barrier.
  const int lane = t & 63;
  for (int base = (t >> 6) * 256; base < np; base += (THREADS >> 6) * 256) {
    uint64_t* chunk = a + base;
    uint64_t mine[4];
    int rank[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { mine[q] = chunk[q * 64 + lane]; rank[q] = 0; }
    for (int j = 0; j < 64; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) rank[q] += chunk[q * 64 + j] < mine[q] ? 1 : 0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) chunk[q * 64 + rank[q]] = mine[q];
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own LDS writes, before its lanes read them
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (p == q) continue;
        const uint64_t* row = chunk + p * 64;
        int pos = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) pos += row[pos + step - 1] < mine[q] ? step : 0;
        pos += row[pos] < mine[q] ? 1 : 0;
        rank[q] += pos;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) b[base + rank[q]] = mine[q];
  }
  __syncthreads();
  uint64_t* src = b;
  uint64_t* dst = a;
  for (int width = 256; width < np; width <<= 1) {
    for (int i = t; i < np; i += THREADS) {
      const uint64_t mine = src[i];
      const int run = i / width, pos = i - run * width;
      const int pstart = (run ^ 1) * width;
      const int plen = max(0, min(width, np - pstart));
      const uint64_t* partner = src + pstart;
      int lo = 0, hi = plen;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (partner[mid] < mine) lo = mid + 1; else hi = mid;
      }
      dst[(run & ~1) * width + pos + lo] = mine;
    }
    __syncthreads();
    uint64_t* tmp = src; src = dst; dst = tmp;
  }
  return src;
}

// Catch-all for buckets fuller than the wave rank sort covers (n > min_n), grid-stride over the tiles (almost
// every tile is skipped).  n <= CAP: merge sort in LDS.  Beyond: the bitonic network in place in global memory.
template <int THREADS, int CAP>
__global__ __launch_bounds__(THREADS) void tile_sort_kernel(int num_tiles, const int2* tile_ranges, uint64_t* pairs,
                                                            int* o2p, uint64_t* keys_out, int depth16, int min_n,
                                                            int max_n) {
  extern __shared__ uint64_t s_sort[];  // 2 * CAP keys (dynamic: 128 KB of the CU's 160 KB at CAP = 8192)
  uint64_t* s_a = s_sort;
  uint64_t* s_b = s_sort + CAP;
  for (int tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    const int2 r = tile_ranges[tile];
    const int n = r.y - r.x;
    if (n <= min_n || n > max_n) continue;  // uniform over the workgroup; another launch covers the rest
    uint64_t* seg = pairs + r.x;
    const int t = threadIdx.x;
    const uint64_t* data = seg;
    if (n <= CAP) {
      for (int i = t; i < n; i += THREADS) s_a[i] = seg[i];
      data = lds_merge_sort<THREADS, CAP>(s_a, s_b, n, t);  // starts with a barrier
    } else {
      __syncthreads();
      bitonic_sort<THREADS>(seg, n, t);
    }
    const int shift = depth16 ? 16 : 32;
    for (int i = t; i < n; i += THREADS) {
      const uint64_t kv = data[i];
      o2p[r.x + i] = int(uint32_t(kv));
      if (keys_out) keys_out[r.x + i] = (kv >> 32) | (uint64_t(uint32_t(tile)) << shift);
    }
    __syncthreads();
  }
}

// cuda_lib.segmented_sort_pairs (cuda_lib/segmented_sort_pairs.cu:8-78): ascending sort of (key, value) pairs inside
// each [start, end) segment; signed 16- or 32-bit keys, int32 values.  One workgroup per segment (grid-stride) on the
// same machinery as the crowded-tile sort: composites (biased key << 32 | position in the segment) are unique, so the
// result is the stable order.
template <typename K, int THREADS, int CAP>
__global__ __launch_bounds__(THREADS) void segmented_sort_kernel(int num_segments, const int64_t* seg_start,
                                                                 const int64_t* seg_end, const K* keys,
                                                                 const int* values, K* keys_out, int* values_out,
                                                                 uint64_t* scratch) {
  extern __shared__ uint64_t s_sort[];
  uint64_t* s_a = s_sort;
  uint64_t* s_b = s_sort + CAP;
  const int t = threadIdx.x;
  const uint32_t bias = sizeof(K) == 2 ? 0x8000u : 0x80000000u;  // signed -> unsigned order
  for (int seg = blockIdx.x; seg < num_segments; seg += gridDim.x) {
    const int64_t lo = seg_start[seg];
    const int n = int(seg_end[seg] - lo);
    if (n <= 0) continue;
    uint64_t* stage = n <= CAP ? s_a : scratch + lo;
    for (int i = t; i < n; i += THREADS) {
      const uint32_t k = (sizeof(K) == 2 ? uint32_t(uint16_t(keys[lo + i])) : uint32_t(keys[lo + i])) ^ bias;
      stage[i] = (uint64_t(k) << 32) | uint64_t(uint32_t(i));
    }
    const uint64_t* data;
    if (n <= CAP) {
      data = lds_merge_sort<THREADS, CAP>(s_a, s_b, n, t);
    } else {
      __syncthreads();
      bitonic_sort<THREADS>(stage, n, t);
      data = stage;
    }
    for (int i = t; i < n; i += THREADS) {
      const uint64_t kv = data[i];
      keys_out[lo + i] = K(uint32_t(kv >> 32) ^ bias);
      values_out[lo + i] = values[lo + int(uint32_t(kv))];
    }
    __syncthreads();
  }
}

// Rank sort, one WAVE per tile, for buckets of up to 64*R pairs (the common case: a few hundred
// splats per tile).  Keys are unique (the Gaussian index is the low word), so the rank of a key
// -- the number of keys below it -- is its final position.  Each lane keeps R keys in registers
// (row q = keys q*64 .. q*64+63).  Two levels:
//   1. every row is ranked against ITSELF: 64 wave-uniform LDS broadcasts per row, lanes count
//      (n compares per lane instead of n*R), and the row is written back to LDS in sorted order;
//   2. a key's rank among the other rows is a lower bound in each of those sorted rows: 7 probes.
// No barriers (a workgroup is one wave), no data-dependent control flow.  Slots past n hold pad keys
// 0xFFFFFFFF'00000000 | slot: unique, above every real key (the high word of a real key is the bit
// pattern of a depth in [0,1] or a 16-bit code), so they sort to the end of their row.
template <int R>
__device__ __forceinline__ void rank_sort_rows(uint64_t* s_key, int n, int lane, int start, int tile, int* o2p,
                                               uint64_t* keys_out, int shift) {
  uint64_t mine[R];
  int rank[R];
#pragma unroll
  for (int q = 0; q < R; ++q) {
    mine[q] = s_key[q * 64 + lane];
    rank[q] = 0;
  }
  if (R == 1) {
    for (int j = 0; j < n; ++j) rank[0] += s_key[j] < mine[0] ? 1 : 0;
  } else {
    for (int j = 0; j < 64; ++j) {
#pragma unroll
      for (int q = 0; q < R; ++q) rank[q] += s_key[q * 64 + j] < mine[q] ? 1 : 0;
    }
#pragma unroll
    for (int q = 0; q < R; ++q) s_key[q * 64 + rank[q]] = mine[q];  // in place: every broadcast above is done
    __syncthreads();
#pragma unroll
    for (int q = 0; q < R; ++q) {
#pragma unroll
      for (int p = 0; p < R; ++p) {
        if (p == q) continue;
        const uint64_t* row = s_key + p * 64;
        int pos = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) pos += row[pos + step - 1] < mine[q] ? step : 0;
        pos += row[pos] < mine[q] ? 1 : 0;
        rank[q] += pos;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < R; ++q) {
    if ((mine[q] >> 32) != 0xFFFFFFFFull) {
      o2p[start + rank[q]] = int(uint32_t(mine[q]));
      if (keys_out) keys_out[start + rank[q]] = (mine[q] >> 32) | (uint64_t(uint32_t(tile)) << shift);
    }
  }
}

// Bucket sort of the same buckets, tried first (round 3): depths inside a tile are spread out, so a counting sort on
// a 256-bin digit of the depth word leaves ~1 key per bin and a key's final position is its bin's start plus its rank
// among the handful of keys that share the bin -- ~10 + 2 (keys per bin) compares per key instead of 64 broadcast
// compares per row and key plus 7 probes for every other row.  The digit is floor((depth word - min) * 256 / (max -
// min + 1)) in f32: monotonic in the depth word, which is all the final order needs (ties inside a bin are resolved on
// the full 64-bit composite).  Keys are scattered IN PLACE (s_key is dead once every lane holds its rows in registers).
// Returns false, leaving the registers' worth of keys unplaced, when some bin holds more than BIN_LIMIT keys (depths
// clustered on one surface): the caller reloads the bucket and runs the rank sort above, whose cost does not depend
// on the distribution.
#ifndef GS_SORT_BINS
#define GS_SORT_BINS 1
#endif
constexpr int SORT_BINS = 256, BIN_LIMIT = 40;

template <int R>
__device__ __forceinline__ bool bucket_sort_rows(uint64_t* s_key, int* s_hist, int n, int lane, int start, int tile,
                                                 int* o2p, uint64_t* keys_out, int shift) {
  uint64_t mine[R];
  uint32_t mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
  for (int q = 0; q < R; ++q) {
    mine[q] = s_key[q * 64 + lane];
    if (q * 64 + lane < n) {
      const uint32_t hi = uint32_t(mine[q] >> 32);
      mn = min(mn, hi);
      mx = max(mx, hi);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mn = min(mn, uint32_t(__shfl_xor(int(mn), off)));
    mx = max(mx, uint32_t(__shfl_xor(int(mx), off)));
  }
  const float scale = float(SORT_BINS) / (float(mx - mn) + 1.0f);
#pragma unroll
  for (int k = 0; k < SORT_BINS / 64; ++k) s_hist[k * 64 + lane] = 0;
  __syncthreads();
  int bin[R], arrival[R];
#pragma unroll
  for (int q = 0; q < R; ++q) {
    bin[q] = 0; arrival[q] = 0;
    if (q * 64 + lane < n) {
      bin[q] = min(SORT_BINS - 1, int(float(uint32_t(mine[q] >> 32) - mn) * scale));
      arrival[q] = atomicAdd(&s_hist[bin[q]], 1);
    }
  }
  __syncthreads();
  // exclusive scan of the bins: lane l owns bins 4 l .. 4 l + 3
  int c[SORT_BINS / 64], local = 0, fullest = 0;
#pragma unroll
  for (int k = 0; k < SORT_BINS / 64; ++k) {
    c[k] = s_hist[lane * (SORT_BINS / 64) + k];
    local += c[k];
    fullest = max(fullest, c[k]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) fullest = max(fullest, __shfl_xor(fullest, off));
  if (fullest > BIN_LIMIT) return false;  // wave-uniform
  int run = wave_inclusive_scan(local) - local;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < SORT_BINS / 64; ++k) {
    s_hist[lane * (SORT_BINS / 64) + k] = run;
    run += c[k];
  }
  if (lane == 63) s_hist[SORT_BINS] = run;  // = n
  __syncthreads();
#pragma unroll
  for (int q = 0; q < R; ++q)
    if (q * 64 + lane < n) s_key[s_hist[bin[q]] + arrival[q]] = mine[q];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < R; ++q) {
    if (q * 64 + lane >= n) continue;
    const int b0 = s_hist[bin[q]], b1 = s_hist[bin[q] + 1];
    int rank = b0;
    for (int j = b0; j < b1; ++j) rank += s_key[j] < mine[q] ? 1 : 0;
    o2p[start + rank] = int(uint32_t(mine[q]));
    if (keys_out) keys_out[start + rank] = (mine[q] >> 32) | (uint64_t(uint32_t(tile)) << shift);
  }
  return true;
}

// RMAX bounds the LDS buffer (64*RMAX keys); the number of register rows is chosen PER TILE from its
// own population, so a 100-splat tile in a frame whose fullest tile holds 500 does 2 rows of
// compares, not 8.
template <int RMAX>
// __launch_bounds__(64, 6): left alone the compiler unrolls the 64 broadcast rounds with ~30 keys in flight and ends up
// at 195 VGPRs = 2 waves per SIMD for a kernel that waits on dependent LDS probes; asked for 6 waves per SIMD (<= 80
// VGPRs) the sort takes 37 us instead of 55 at C3 (measured: 3 / 4 / 6 / 8 waves -> 92 / 87 / 85 / 92 us for
// gs_map_finish).
__global__ __launch_bounds__(64, 6) void tile_rank_sort_kernel(int num_tiles, const int2* tile_ranges,
                                                            uint64_t* pairs, int* o2p, uint64_t* keys_out,
                                                            int depth16, int skip_full) {
  __shared__ uint64_t s_key[64 * RMAX];
  __shared__ int s_hist[SORT_BINS + 1];
  const int tile = gs_xcd_remap(blockIdx.x, num_tiles);
  if (tile < 0) return;
  const int2 r = tile_ranges[tile];
  const int n = r.y - r.x;
  if (n <= 0) return;
  const int lane = threadIdx.x;
  const int shift = depth16 ? 16 : 32;
  if (n > 64 * RMAX) {
    if (skip_full) return;  // a dedicated bitonic launch follows for these
    // fuller than this launch was sized for (only when the caller's hint was low): this wave sorts the
    // bucket in place in global memory -- slow, rare, and never wrong
    uint64_t* gseg = pairs + r.x;
    bitonic_sort<64>(gseg, n, lane);
    for (int i = lane; i < n; i += 64) {
      const uint64_t kv = gseg[i];
      o2p[r.x + i] = int(uint32_t(kv));
      if (keys_out) keys_out[r.x + i] = (kv >> 32) | (uint64_t(uint32_t(tile)) << shift);
    }
    return;
  }
  const uint64_t* seg = pairs + r.x;
  const int rows = (n + 63) >> 6;  // exactly as many register rows as the bucket needs (cost grows with rows^2)
  for (int i = lane; i < rows * 64; i += 64) s_key[i] = i < n ? seg[i] : (0xFFFFFFFF00000000ull | uint64_t(i));
  __syncthreads();
  if (GS_SORT_BINS && rows >= 2) {
    bool done = false;
    switch (rows) {
      case 2: done = bucket_sort_rows<2>(s_key, s_hist, n, lane, r.x, tile, o2p, keys_out, shift); break;
      case 3: done = bucket_sort_rows<3>(s_key, s_hist, n, lane, r.x, tile, o2p, keys_out, shift); break;
      case 4: done = bucket_sort_rows<4>(s_key, s_hist, n, lane, r.x, tile, o2p, keys_out, shift); break;
      case 5: done = bucket_sort_rows<(RMAX >= 8 ? 5 : 2)>(s_key, s_hist, n, lane, r.x, tile, o2p, keys_out, shift); break;
      case 6: done = bucket_sort_rows<(RMAX >= 8 ? 6 : 2)>(s_key, s_hist, n, lane, r.x, tile, o2p, keys_out, shift); break;
      case 7: done = bucket_sort_rows<(RMAX >= 8 ? 7 : 2)>(s_key, s_hist, n, lane, r.x, tile, o2p, keys_out, shift); break;
      default: done = bucket_sort_rows<(RMAX >= 8 ? 8 : 2)>(s_key, s_hist, n, lane, r.x, tile, o2p, keys_out, shift); break;
    }
    if (done) return;
    // clustered depths: s_key is untouched up to here (the scatter comes after the bin-size check)
  }
  switch (rows) {
    case 1: rank_sort_rows<1>(s_key, n, lane, r.x, tile, o2p, keys_out, shift); break;
    case 2: rank_sort_rows<2>(s_key, n, lane, r.x, tile, o2p, keys_out, shift); break;
    case 3: rank_sort_rows<3>(s_key, n, lane, r.x, tile, o2p, keys_out, shift); break;
    case 4: rank_sort_rows<4>(s_key, n, lane, r.x, tile, o2p, keys_out, shift); break;
    case 5: rank_sort_rows<(RMAX >= 8 ? 5 : 1)>(s_key, n, lane, r.x, tile, o2p, keys_out, shift); break;
    case 6: rank_sort_rows<(RMAX >= 8 ? 6 : 1)>(s_key, n, lane, r.x, tile, o2p, keys_out, shift); break;
    case 7: rank_sort_rows<(RMAX >= 8 ? 7 : 1)>(s_key, n, lane, r.x, tile, o2p, keys_out, shift); break;
    default: rank_sort_rows<(RMAX >= 8 ? 8 : 1)>(s_key, n, lane, r.x, tile, o2p, keys_out, shift); break;
  }
}

// ---- reference-shaped primitives ----------------------------------------------------------
__global__ __launch_bounds__(256) void tile_count_kernel(MapArgs a, int* counts) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= live_count(a)) return;
  const GridQuery q = grid_query(a.points + 7 * i, a.Wp, a.Hp, a.tile_size, a.thr);
  int c = 0;
  for (int ty = 0; ty < q.span_y; ++ty)
    for (int tx = 0; tx < q.span_x; ++tx) c += test_tile(q, tx, ty, a.tile_size) ? 1 : 0;
  counts[i] = c;
}

__global__ __launch_bounds__(256) void tile_emit_keys_kernel(MapArgs a, const int* offsets, uint64_t* keys,
                                                             int* values) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= live_count(a)) return;
  const GridQuery q = grid_query(a.points + 7 * i, a.Wp, a.Hp, a.tile_size, a.thr);
  int64_t k = offsets[i];
  const uint64_t dk = depth_key(a.depth[i], a.depth16 != 0);
  const int shift = a.depth16 ? 16 : 32;
  // ti.ndrange(span.x, span.y): x outer, y inner (tile_mapper.py:134)
  for (int tx = 0; tx < q.span_x; ++tx)
    for (int ty = 0; ty < q.span_y; ++ty)
      if (test_tile(q, tx, ty, a.tile_size)) {
        const int tile_id = (tx + q.min_tx) + (ty + q.min_ty) * a.tiles_wide;
        keys[k] = dk | (uint64_t(uint32_t(tile_id)) << shift);
        values[k] = int(i);
        ++k;
      }
}

__global__ __launch_bounds__(256) void find_ranges_kernel(int64_t k, const uint64_t* keys, int shift, int* ranges) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= k) return;
  const int64_t t = int64_t(keys[i] >> shift);
  if (i == 0 || int64_t(keys[i - 1] >> shift) != t) ranges[2 * t] = int(i);
  if (i + 1 == k || int64_t(keys[i + 1] >> shift) != t) ranges[2 * t + 1] = int(i + 1);
}

// block-level exclusive scan, 3 kernels: (1) per-block sums, (2) scan of sums (one block),
// (3) per-block scan + offset.  1024 elements per block (256 threads x 4).
constexpr int SCAN_BLOCK = 1024;

__global__ __launch_bounds__(256) void scan_block_sums(int64_t n, const int* in, int* sums) {
  __shared__ int s[256];
  const int64_t base = int64_t(blockIdx.x) * SCAN_BLOCK;
  int acc = 0;
  for (int e = 0; e < 4; ++e) {
    const int64_t i = base + threadIdx.x * 4 + e;
    if (i < n) acc += in[i];
  }
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[blockIdx.x] = s[0];
}

__global__ __launch_bounds__(1024) void scan_sums(int nb, int* sums) {  // exclusive, in place, single block
  __shared__ int s[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nb ? sums[i] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      int x = s[threadIdx.x];
      if (threadIdx.x >= off) x += s[threadIdx.x - off];
      __syncthreads();
      s[threadIdx.x] = x;
      __syncthreads();
    }
    if (i < nb) sums[i] = carry + s[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += s[1023];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void scan_apply(int64_t n, const int* in, const int* sums, int* out) {
  __shared__ int s[256];
  const int64_t base = int64_t(blockIdx.x) * SCAN_BLOCK;
  int v[4], acc = 0;
  for (int e = 0; e < 4; ++e) {
    const int64_t i = base + threadIdx.x * 4 + e;
    v[e] = i < n ? in[i] : 0;
    acc += v[e];
  }
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    int x = s[threadIdx.x];
    if (threadIdx.x >= off) x += s[threadIdx.x - off];
    __syncthreads();
    s[threadIdx.x] = x;
    __syncthreads();
  }
  int run = sums[blockIdx.x] + s[threadIdx.x] - acc;
  for (int e = 0; e < 4; ++e) {
    const int64_t i = base + threadIdx.x * 4 + e;
    if (i < n) out[i] = run;
    run += v[e];
    if (i == n - 1) out[n] = run;  // the total, appended (full_cumsum.cu:36-41)
  }
}

int fill_args(MapArgs& a, int64_t v, const float* points, const float* depth, int width, int height,
              const GsRasterConfig* cfg, int depth16, const GsRowShard* shard = nullptr) {
  if (int rc = gs_check_cfg(cfg)) return rc;
  GS_REQUIRE(width > 0 && height > 0, GS_ERR_INVALID_ARGUMENT, "mapper: image size %dx%d", width, height);
  GS_REQUIRE(v >= 0 && v < (int64_t(1) << 31), GS_ERR_INVALID_ARGUMENT, "mapper: %lld gaussians", (long long)v);
  const int ts = cfg->tile_size;
  a.points = points; a.depth = depth; a.v = v; a.v_dev = nullptr;
  a.Wp = int(gs_div_up(width, ts)) * ts;  // pad_to_tile, tile_mapper.py:18-22
  a.Hp = int(gs_div_up(height, ts)) * ts;
  a.tile_size = ts;
  a.tiles_wide = a.Wp / ts;
  a.thr = cfg->alpha_threshold;
  a.depth16 = depth16;
  return gs_make_shard(shard, a.Hp / ts, &a.sh);
}

}  // namespace

namespace {
struct MapScratch {
  int* hist; int* cursors; int* region_of; int* order; int* region_count; int* region_start; int* part;
  int* chunk_start;
  QueryCache* qcache;
  int* touched_blocks;  // per 1024-row workgroup of the binning pass: rows that entered the ordering
  int* block_start;     // rows of workgroup b of the binning pass: [block_start[b], block_start[b + 1])
};
// part[region][workgroup]; the region count is bounded by the tile count and by MAX_REGIONS
int64_t part_entries(int64_t v, int64_t num_tiles) {
  const int64_t regions = num_tiles < MAX_REGIONS ? (num_tiles < 1 ? 1 : num_tiles) : MAX_REGIONS;
  return regions * gs_div_up(v > 0 ? v : 1, BIN);
}
MapScratch carve(void* scratch, int64_t v, int64_t num_tiles) {
  char* p = static_cast<char*>(scratch);
  MapScratch m;
  auto take = [&](int64_t bytes) { int* r = reinterpret_cast<int*>(p); p += gs_align_up(bytes, 256); return r; };
  m.hist = take(num_tiles * 4);
  m.cursors = take(num_tiles * 4);
  m.region_of = take(v * 4);
  m.order = take(v * 4);
  m.region_count = take((MAX_REGIONS + 1) * 4);
  m.region_start = take((MAX_REGIONS + 1) * 4);
  m.chunk_start = take((MAX_REGIONS + 1) * 4);
  m.part = take(part_entries(v, num_tiles) * 4);
  m.qcache = reinterpret_cast<QueryCache*>(take(v * int64_t(sizeof(QueryCache))));
  m.touched_blocks = take((gs_div_up(v > 0 ? v : 1, BIN) + 1) * 4);
  m.block_start = take((gs_div_up(v > 0 ? v : 1, BIN) + 1) * 4);
  return m;
}
RegionGrid make_grid(const MapArgs& a) {
  RegionGrid rg;
  rg.tiles_x = a.tiles_wide;
  rg.row0 = a.sh.period == 1 ? a.sh.begin : 0;
  rg.tiles_y = a.sh.period == 1 ? a.sh.end - a.sh.begin : a.Hp / a.tile_size;
  rg.rg = RG_MIN;
  while (gs_div_up(rg.tiles_x, rg.rg) * gs_div_up(rg.tiles_y, rg.rg) > MAX_REGIONS) rg.rg *= 2;
  rg.win = rg.rg + 2 * RB;
  rg.regions_x = int(gs_div_up(rg.tiles_x, rg.rg));
  rg.num_regions = rg.regions_x * int(gs_div_up(rg.tiles_y, rg.rg));
  return rg;
}
}  // namespace

extern "C" int64_t gs_map_scratch_bytes(int64_t v, int64_t num_tiles) {
  return gs_align_up(num_tiles * 4, 256) * 2 + gs_align_up(v * 4, 256) * 2 +
         gs_align_up((MAX_REGIONS + 1) * 4, 256) * 3 + gs_align_up(part_entries(v, num_tiles) * 4, 256) +
         gs_align_up(v * int64_t(sizeof(QueryCache)), 256) + 2 * gs_align_up((gs_div_up(v > 0 ? v : 1, BIN) + 1) * 4, 256);
}

extern "C" int64_t gs_map_touched_offset(int64_t v, int64_t num_tiles) {
  const MapScratch m = carve(nullptr, v, num_tiles);
  return reinterpret_cast<char*>(m.order) - static_cast<char*>(nullptr);
}

namespace {
// ascending list of the rows with region_of >= 0: a stable compaction on the per-workgroup counts the binning pass
// left (every workgroup adds up the counts in front of it, as the projection's compaction does)
__global__ __launch_bounds__(1024) void touched_write_kernel(const int* block_start, const int* region_of,
                                                             const int* block_counts, int* touched, int* count_out) {
  __shared__ int s_wave[16];
  __shared__ int s_before[16];
  const int i = block_start[blockIdx.x] + int(threadIdx.x);  // the binning pass's own rows (ascending over workgroups)
  const bool flag = i < block_start[blockIdx.x + 1] && region_of[i] >= 0;
  const uint64_t b = __ballot(flag);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) s_wave[wave] = __popcll(b);
  int before = 0;
  for (int j = threadIdx.x; j < int(blockIdx.x); j += 1024) before += block_counts[j];
  for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off);
  if (lane == 0) s_before[wave] = before;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < 16; ++w) base += s_before[w];
  for (int w = 0; w < wave; ++w) base += s_wave[w];
  if (flag) touched[base + __popcll(b & ((1ull << lane) - 1ull))] = i;
  if (count_out && blockIdx.x == gridDim.x - 1 && threadIdx.x == 1023) {  // the last thread knows the total
    int total = base + __popcll(b);
    for (int w = wave + 1; w < 16; ++w) total += s_wave[w];  // (wave 15 here: nothing behind it)
    *count_out = total;
  }
}

// owner r's share of the ascending list: [first row whose Gaussian index >= r chunk, ... (r + 1) chunk)
__global__ void owner_cuts_kernel(const int* block_counts, int num_blocks, const int* touched, const int64_t* indexes,
                                  int64_t chunk, int world, int64_t* owner_counts, const int* v_dev, int64_t v,
                                  int owner, int* owned_rows) {
  __shared__ int64_t s_cut[65];
  __shared__ int s_part[128];
  const int t = threadIdx.x;
  int part = 0;
  for (int j = t; j < num_blocks; j += 128) part += block_counts[j];
  s_part[t] = part;
  __syncthreads();
  int m = 0;
  for (int j = 0; j < 128; ++j) m += s_part[j];
  if (t <= world) {
    const int64_t bound = int64_t(t) * chunk;
    int lo = 0, hi = m;  // first e with indexes[touched[e]] >= bound
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (indexes[touched[mid]] < bound) lo = mid + 1; else hi = mid;
    }
    s_cut[t] = t == world ? m : lo;
  }
  __syncthreads();
  if (t < world) owner_counts[t] = s_cut[t + 1] - s_cut[t];
  // the rows (of the whole visible list, not only the touched ones) whose Gaussians `owner` owns: [first row with
  // index >= owner chunk, first row with index >= (owner + 1) chunk)
  if (owned_rows && t >= 126) {
    const int64_t bound = int64_t(owner + (t - 126)) * chunk;
    int64_t lo = 0, hi = v_dev ? (int64_t(*v_dev) < v ? int64_t(*v_dev) : v) : v;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (indexes[mid] < bound) lo = mid + 1; else hi = mid;
    }
    owned_rows[t - 126] = int(lo);
  }
}
}  // namespace

extern "C" int gs_map_touched_list(int64_t v, const int32_t* v_dev, int64_t num_tiles, const void* scratch,
                                   int64_t scratch_bytes, int32_t* touched_out, int32_t* count_out,
                                   const int64_t* indexes, int64_t n, int32_t world, int64_t* owner_counts,
                                   int32_t owner, int32_t* owned_rows, void* stream) {
  GS_REQUIRE(v >= 0 && v < (int64_t(1) << 31) && num_tiles >= 1, GS_ERR_INVALID_ARGUMENT, "gs_map_touched_list: sizes");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (owner_counts) {
    GS_REQUIRE(world >= 1 && world <= 64 && indexes && n >= 0, GS_ERR_INVALID_ARGUMENT,
               "gs_map_touched_list: %d owners (1 .. 64) need the index list", world);
    if (hipMemsetAsync(owner_counts, 0, size_t(world) * 8, s) != hipSuccess) {
      gs_set_error("gs_map_touched_list: hipMemsetAsync failed");
      return GS_ERR_LAUNCH;
    }
  }
  if (count_out && v == 0 && hipMemsetAsync(count_out, 0, 4, s) != hipSuccess) {
    gs_set_error("gs_map_touched_list: hipMemsetAsync failed");
    return GS_ERR_LAUNCH;
  }
  if (v == 0) return GS_OK;
  const int nb = int(gs_div_up(v, 1024));
  GS_REQUIRE(scratch && scratch_bytes >= gs_map_scratch_bytes(v, num_tiles), GS_ERR_SCRATCH_TOO_SMALL,
             "gs_map_touched_list: scratch is not the one gs_map_prepare filled");
  GS_REQUIRE(touched_out, GS_ERR_INVALID_ARGUMENT, "gs_map_touched_list: touched_out is NULL");
  const MapScratch m = carve(const_cast<void*>(scratch), v, num_tiles);
  const int* block_counts = m.touched_blocks;  // BIN = 1024 rows per workgroup of the binning pass
  hipLaunchKernelGGL(touched_write_kernel, dim3(nb), dim3(1024), 0, s, m.block_start, m.region_of, block_counts,
                     touched_out, count_out);
  if (owner_counts) {
    GS_REQUIRE(!owned_rows || (owner >= 0 && owner < world), GS_ERR_INVALID_ARGUMENT,
               "gs_map_touched_list: owned_rows needs the owner's rank");
    hipLaunchKernelGGL(owner_cuts_kernel, dim3(1), dim3(128), 0, s, block_counts, nb, touched_out, indexes,
                       gs_div_up(n > 0 ? n : 1, world), world, owner_counts, v_dev, v, owner, owned_rows);
  }
  GS_CHECK_LAUNCH("gs_map_touched_list");
  return GS_OK;
}

// (workgroup, region) pairs up to which the binning pass reserves its places with atomics (see K1 / K2)
bool bin_with_atomics(int64_t num_wg, const RegionGrid& rg) { return num_wg * rg.num_regions <= (int64_t(1) << 16); }

// the region counters of the plan's mapper scratch: whoever runs in front of the binning pass clears them (the frame
// calls: the projection's first pass; standalone: a memset)
int gs_map_bin_counters(const GsMapBinPlan* plan, int64_t n, int32_t** words, int32_t* count) {
  MapArgs a;
  if (int rc = fill_args(a, n, nullptr, nullptr, plan->width, plan->height, plan->cfg, 0, plan->shard)) return rc;
  GS_REQUIRE(a.sh.local_rows > 0 && plan->scratch, GS_ERR_INVALID_ARGUMENT, "gs_map_bin_counters: nothing to bin");
  const MapScratch m = carve(plan->scratch, n, int64_t(a.tiles_wide) * a.sh.local_rows);
  const bool atomics = bin_with_atomics(gs_div_up(n > 0 ? n : 1, BIN), make_grid(a));
  *words = atomics ? m.region_count : nullptr;  // the scan launches write every counter themselves
  *count = atomics ? MAX_REGIONS + 1 : 0;
  return GS_OK;
}

int gs_map_compact_bin(const GsMapBinPlan* plan, const GsCompactArgs* c, void* stream) {
  MapArgs a;
  if (int rc = fill_args(a, c->n, c->points, nullptr, plan->width, plan->height, plan->cfg, 0, plan->shard)) return rc;
  GS_REQUIRE(a.sh.local_rows > 0 && c->n > 0, GS_ERR_INVALID_ARGUMENT, "gs_map_compact_bin: nothing to bin");
  const int num_tiles = a.tiles_wide * a.sh.local_rows;
  GS_REQUIRE(plan->scratch && plan->scratch_bytes >= gs_map_scratch_bytes(c->n, num_tiles), GS_ERR_SCRATCH_TOO_SMALL,
             "gs_map_compact_bin: mapper scratch %lld < %lld bytes", (long long)plan->scratch_bytes,
             (long long)gs_map_scratch_bytes(c->n, num_tiles));
  const RegionGrid rg = make_grid(a);
  GS_REQUIRE(rg.num_regions <= MAX_REGIONS, GS_ERR_UNSUPPORTED, "gs_map_compact_bin: %d regions", rg.num_regions);
  const MapScratch m = carve(plan->scratch, c->n, num_tiles);
  const unsigned vb = unsigned(gs_div_up(c->n, BIN));
  hipLaunchKernelGGL(compact_bin_kernel, dim3(vb), dim3(BIN), 0, static_cast<hipStream_t>(stream), *c, a, rg, int(vb),
                     m.region_of, m.part, m.touched_blocks, m.block_start,
                     bin_with_atomics(vb, rg) ? m.region_count : nullptr);
  GS_CHECK_LAUNCH("gs_map_compact_bin");
  return GS_OK;
}

int64_t gs_map_one_pass_scratch_bytes(int64_t n) { return gs_align_up((gs_div_up(n > 0 ? n : 1, BIN) + 2) * 8, 256); }

// one pass: `pa` is project.hip's filled gs_proj::ProjArgs; lookback: gs_map_one_pass_scratch_bytes(n) bytes
int gs_map_project_compact_bin(const GsMapBinPlan* plan, const void* pa, const GsCompactArgs* c, float* camera_pos,
                               void* lookback, void* stream) {
  MapArgs a;
  if (int rc = fill_args(a, c->n, c->points, nullptr, plan->width, plan->height, plan->cfg, 0, plan->shard)) return rc;
  GS_REQUIRE(a.sh.local_rows > 0 && c->n > 0 && lookback, GS_ERR_INVALID_ARGUMENT, "gs_map_project_compact_bin: nothing to bin");
  const int num_tiles = a.tiles_wide * a.sh.local_rows;
  GS_REQUIRE(plan->scratch && plan->scratch_bytes >= gs_map_scratch_bytes(c->n, num_tiles), GS_ERR_SCRATCH_TOO_SMALL,
             "gs_map_project_compact_bin: mapper scratch %lld < %lld bytes", (long long)plan->scratch_bytes,
             (long long)gs_map_scratch_bytes(c->n, num_tiles));
  const RegionGrid rg = make_grid(a);
  GS_REQUIRE(rg.num_regions <= MAX_REGIONS, GS_ERR_UNSUPPORTED, "gs_map_project_compact_bin: %d regions", rg.num_regions);
  const MapScratch m = carve(plan->scratch, c->n, num_tiles);
  const unsigned vb = unsigned(gs_div_up(c->n, BIN));
  hipStream_t s = static_cast<hipStream_t>(stream);
  int* counters = bin_with_atomics(vb, rg) ? m.region_count : nullptr;
  if (counters && hipMemsetAsync(counters, 0, size_t(MAX_REGIONS + 1) * 4, s) != hipSuccess) {
    gs_set_error("gs_map_project_compact_bin: hipMemsetAsync failed");
    return GS_ERR_LAUNCH;
  }
  if (c->block_offsets == reinterpret_cast<const int*>(1)) {  // variant 2: count, then project again
    GsCompactArgs c2 = *c;
    c2.block_offsets = nullptr;
    c2.block_counts = static_cast<int*>(lookback);
    hipLaunchKernelGGL(project_count_kernel, dim3(vb), dim3(BIN), 0, s, *static_cast<const gs_proj::ProjArgs*>(pa),
                       static_cast<int*>(lookback));
    hipLaunchKernelGGL(project_compact_bin_kernel<false>, dim3(vb), dim3(BIN), 0, s,
                       *static_cast<const gs_proj::ProjArgs*>(pa), c2, a, rg, int(vb), m.region_of, m.part,
                       m.touched_blocks, m.block_start, counters, nullptr, nullptr, nullptr, camera_pos);
    GS_CHECK_LAUNCH("gs_map_project_compact_bin");
    return GS_OK;
  }
  // descriptors, then the ticket counter and the failure flag
  if (hipMemsetAsync(lookback, 0, size_t(gs_map_one_pass_scratch_bytes(c->n)), s) != hipSuccess) {
    gs_set_error("gs_map_project_compact_bin: hipMemsetAsync failed");
    return GS_ERR_LAUNCH;
  }
  unsigned long long* desc = static_cast<unsigned long long*>(lookback);
  int* ticket = reinterpret_cast<int*>(desc + vb);
  hipLaunchKernelGGL(project_compact_bin_kernel<true>, dim3(vb), dim3(BIN), 0, s,
                     *static_cast<const gs_proj::ProjArgs*>(pa), *c, a, rg, int(vb), m.region_of, m.part,
                     m.touched_blocks, m.block_start, counters, desc, ticket, ticket + 1, camera_pos);
  GS_CHECK_LAUNCH("gs_map_project_compact_bin");
  return GS_OK;
}

extern "C" int gs_map_prepare(int64_t v, const int32_t* v_dev, const float* points, int32_t width, int32_t height,
                              const GsRasterConfig* cfg, int64_t k_capacity, int32_t* tile_ranges,
                              int32_t* counts_out, int32_t* counts_host, int32_t* tile_order,
                              const GsRowShard* shard, void* scratch, int64_t scratch_bytes, void* stream) {
  return gs_map_prepare_ex(v, v_dev, points, width, height, cfg, k_capacity, tile_ranges, counts_out, counts_host,
                           tile_order, shard, scratch, scratch_bytes, 0, stream);
}

// binned != 0: gs_map_compact_bin has filled region_of / part / touched_blocks / block_start of `scratch` already
int gs_map_prepare_ex(int64_t v, const int32_t* v_dev, const float* points, int32_t width, int32_t height,
                      const GsRasterConfig* cfg, int64_t k_capacity, int32_t* tile_ranges, int32_t* counts_out,
                      int32_t* counts_host, int32_t* tile_order, const GsRowShard* shard, void* scratch,
                      int64_t scratch_bytes, int binned, void* stream) {
  MapArgs a;
  if (int rc = fill_args(a, v, points, nullptr, width, height, cfg, 0, shard)) return rc;
  a.v_dev = v_dev;
  GS_REQUIRE(a.sh.local_rows > 0, GS_ERR_INVALID_ARGUMENT, "gs_map_prepare: the shard owns no tile row");
  const int num_tiles = a.tiles_wide * a.sh.local_rows;
  GS_REQUIRE(tile_ranges && counts_out && scratch, GS_ERR_INVALID_ARGUMENT, "gs_map_prepare: NULL buffer");
  GS_REQUIRE(scratch_bytes >= gs_map_scratch_bytes(v, num_tiles), GS_ERR_SCRATCH_TOO_SMALL,
             "gs_map_prepare: scratch %lld < %lld bytes", (long long)scratch_bytes,
             (long long)gs_map_scratch_bytes(v, num_tiles));
  GS_REQUIRE(num_tiles <= (1 << 20), GS_ERR_UNSUPPORTED, "gs_map_prepare: %d tiles (limit 2^20)", num_tiles);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const MapScratch m = carve(scratch, v, num_tiles);
  int* hist = m.hist;
  int* cursors = m.cursors;
  const RegionGrid rg = make_grid(a);
  GS_REQUIRE(rg.num_regions <= MAX_REGIONS && rg.win * rg.win * 8 <= 65536, GS_ERR_UNSUPPORTED,
             "gs_map_prepare: tile grid %dx%d needs %d regions of edge %d", rg.tiles_x, rg.tiles_y, rg.num_regions,
             rg.rg);
  if (v == 0 && hipMemsetAsync(hist, 0, size_t(num_tiles) * 4, s) != hipSuccess) {
    gs_set_error("gs_map_prepare: hipMemsetAsync failed");
    return GS_ERR_LAUNCH;
  }
  if (v > 0) {
    GS_REQUIRE(points, GS_ERR_INVALID_ARGUMENT, "gs_map_prepare: points is NULL");
    const unsigned vb = unsigned(gs_div_up(v, BIN));
    const bool atomics = bin_with_atomics(vb, rg);
    if (!binned) {
      if (atomics && hipMemsetAsync(m.region_count, 0, size_t(MAX_REGIONS + 1) * 4, s) != hipSuccess) {
        gs_set_error("gs_map_prepare: hipMemsetAsync failed");
        return GS_ERR_LAUNCH;
      }
      hipLaunchKernelGGL(region_count_kernel, dim3(vb), dim3(BIN), 0, s, a, rg, int(vb), m.region_of, m.part,
                         m.touched_blocks, m.block_start, atomics ? m.region_count : nullptr);
    }
    if (!atomics) {
      hipLaunchKernelGGL(region_part_scan_kernel, dim3(rg.num_regions), dim3(1024), 0, s, int(vb), m.part,
                         m.region_count, 0);
      hipLaunchKernelGGL(region_scan_kernel, dim3(1), dim3(1024), 0, s, rg.num_regions, m.region_count,
                         m.region_start, m.chunk_start);
    }
    hipLaunchKernelGGL(region_scatter_kernel, dim3(vb), dim3(BIN), 0, s, rg, int(vb), m.region_of, m.part,
                       m.region_count, m.region_start, m.chunk_start, m.block_start, m.order, hist, num_tiles,
                       atomics ? 0 : 1);
    // one workgroup per chunk of <= CHUNK Gaussians of one region; surplus workgroups exit at once
    hipLaunchKernelGGL(count_binned_kernel, dim3(unsigned(gs_div_up(v, CHUNK)) + unsigned(rg.num_regions)), dim3(CHUNK),
                       size_t(rg.win) * rg.win * 4, s, a, rg, m.order,
                       m.region_start, m.chunk_start, hist, m.qcache);
    GS_CHECK_LAUNCH("gs_map_prepare/count");
  }
  hipLaunchKernelGGL(map_scan_kernel, dim3(tile_order ? 2 : 1), dim3(1024), 0, s, num_tiles, hist,
                     reinterpret_cast<int2*>(tile_ranges), cursors, counts_out, k_capacity, tile_order, v_dev,
                     counts_host, v > 0 ? m.region_start + rg.num_regions : nullptr);
  GS_CHECK_LAUNCH("gs_map_prepare/scan");
  return GS_OK;
}

extern "C" int gs_map_finish(int64_t v, const int32_t* v_dev, int64_t k, int32_t max_tile_count, const float* points,
                             const float* depth, int32_t width, int32_t height, const GsRasterConfig* cfg,
                             int32_t use_depth16, const int32_t* tile_ranges, int32_t* overlap_to_point,
                             uint64_t* sorted_keys, void* pair_scratch, const GsRowShard* shard, void* scratch,
                             int64_t scratch_bytes, void* stream) {
  MapArgs a;
  if (int rc = fill_args(a, v, points, depth, width, height, cfg, use_depth16, shard)) return rc;
  a.v_dev = v_dev;
  if (k == 0 || v == 0) return GS_OK;
  const int num_tiles = a.tiles_wide * a.sh.local_rows;
  GS_REQUIRE(points && depth && tile_ranges && overlap_to_point && pair_scratch && scratch, GS_ERR_INVALID_ARGUMENT,
             "gs_map_finish: NULL buffer");
  GS_REQUIRE(scratch_bytes >= gs_map_scratch_bytes(v, num_tiles), GS_ERR_SCRATCH_TOO_SMALL,
             "gs_map_finish: scratch too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const MapScratch m = carve(scratch, v, num_tiles);
  int* cursors = m.cursors;
  const RegionGrid rg = make_grid(a);
  uint64_t* pairs = static_cast<uint64_t*>(pair_scratch);
  // the region ordering left in scratch by gs_map_prepare is reused here
  hipLaunchKernelGGL(emit_binned_kernel, dim3(unsigned(gs_div_up(v, CHUNK)) + unsigned(rg.num_regions)), dim3(CHUNK),
                     size_t(rg.win) * rg.win * 8, s,
                     a, rg, m.order, m.region_start, m.chunk_start, cursors, pairs, m.qcache);
  GS_CHECK_LAUNCH("gs_map_finish/emit");
  const int grid = 8 * int(gs_div_up(num_tiles, 8));
  const int2* r = reinterpret_cast<const int2*>(tile_ranges);
  // max_tile_count > 0: exact population of the fullest tile (read back by the caller);
  // max_tile_count <= 0: unknown -- |max_tile_count| is a hint (0 = none).  A wrong hint costs time only.
  const bool exact = max_tile_count > 0;
  const int guess = exact ? max_tile_count : (max_tile_count < 0 ? -max_tile_count : 1024);
  // Wave rank sort for buckets of up to 256 / 512 pairs (4 KB of LDS per wave at most, so a few crowded tiles
  // do not cost every tile its occupancy); fuller buckets go to the workgroup-per-tile launch when such tiles
  // are known or expected, otherwise (a hint that turns out low) the rank-sort wave sorts them itself, slowly.
  const bool big_pass = guess > 512;
  const int skip_full = (exact || big_pass) ? 1 : 0;
  int covered;
  if (guess <= 256) {
    covered = 256;
    hipLaunchKernelGGL((tile_rank_sort_kernel<4>), dim3(grid), dim3(64), 0, s, num_tiles, r, pairs, overlap_to_point,
                       sorted_keys, use_depth16, skip_full);
  } else {
    covered = 512;
    hipLaunchKernelGGL((tile_rank_sort_kernel<8>), dim3(grid), dim3(64), 0, s, num_tiles, r, pairs, overlap_to_point,
                       sorted_keys, use_depth16, skip_full);
  }
  if (big_pass) {
    // 513 .. 1024 and 1025 .. 2048 pairs: 256-thread workgroups with 16 / 32 KB of LDS (ten / five per CU -- in dense
    // scenes most tiles are here); above that: 1024 threads and 128 KB (one per CU).  A size class is launched only
    // when such tiles are expected.
    constexpr int SMALL = 1024, MID = 2048, CAP = 8192;
    const bool mid_pass = guess > SMALL, huge_pass = guess > MID;
    static const hipError_t lds_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_sort_kernel<1024, CAP>),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CAP * 8);
    GS_REQUIRE(lds_ok == hipSuccess, GS_ERR_LAUNCH, "gs_map_finish: cannot reserve %d bytes of LDS", 2 * CAP * 8);
    // each launch takes the sizes the later ones do not cover (the last one launched takes everything above)
    hipLaunchKernelGGL((tile_sort_kernel<256, SMALL>), dim3(min(num_tiles, 8192)), dim3(256), 2 * SMALL * 8, s,
                       num_tiles, r, pairs, overlap_to_point, sorted_keys, use_depth16, covered,
                       mid_pass ? SMALL : 0x7fffffff);
    if (mid_pass)
      hipLaunchKernelGGL((tile_sort_kernel<256, MID>), dim3(min(num_tiles, 8192)), dim3(256), 2 * MID * 8, s,
                         num_tiles, r, pairs, overlap_to_point, sorted_keys, use_depth16, SMALL,
                         huge_pass ? MID : 0x7fffffff);
    if (huge_pass)
      hipLaunchKernelGGL((tile_sort_kernel<1024, CAP>), dim3(min(num_tiles, 2048)), dim3(1024), 2 * CAP * 8, s,
                         num_tiles, r, pairs, overlap_to_point, sorted_keys, use_depth16, MID, 0x7fffffff);
  }
  GS_CHECK_LAUNCH("gs_map_finish/sort");
  return GS_OK;
}

extern "C" int gs_segmented_sort_pairs(int64_t num_items, int32_t key_bytes, const void* keys, const int32_t* values,
                                       void* keys_out, int32_t* values_out, int64_t num_segments,
                                       const int64_t* start_offsets, const int64_t* end_offsets, void* scratch,
                                       int64_t scratch_bytes, void* stream) {
  GS_REQUIRE(key_bytes == 2 || key_bytes == 4, GS_ERR_UNSUPPORTED,
             "gs_segmented_sort_pairs: %d-byte keys (int16 and int32 are implemented, as in the reference)", key_bytes);
  GS_REQUIRE(num_items >= 0 && num_items < (int64_t(1) << 31) && num_segments >= 0, GS_ERR_INVALID_ARGUMENT,
             "gs_segmented_sort_pairs: %lld items, %lld segments", (long long)num_items, (long long)num_segments);
  if (num_items == 0 || num_segments == 0) return GS_OK;
  GS_REQUIRE(keys && values && keys_out && values_out && start_offsets && end_offsets, GS_ERR_INVALID_ARGUMENT,
             "gs_segmented_sort_pairs: NULL buffer");
  GS_REQUIRE(scratch && scratch_bytes >= num_items * 8, GS_ERR_SCRATCH_TOO_SMALL,
             "gs_segmented_sort_pairs: scratch %lld < %lld bytes", (long long)scratch_bytes, (long long)num_items * 8);
  constexpr int CAP = 8192;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(unsigned(num_segments < 4096 ? num_segments : 4096));
  if (key_bytes == 4) {
    static const hipError_t ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&segmented_sort_kernel<int32_t, 1024, CAP>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CAP * 8);
    GS_REQUIRE(ok == hipSuccess, GS_ERR_LAUNCH, "gs_segmented_sort_pairs: cannot reserve LDS");
    hipLaunchKernelGGL((segmented_sort_kernel<int32_t, 1024, CAP>), grid, dim3(1024), 2 * CAP * 8, s, int(num_segments),
                       start_offsets, end_offsets, static_cast<const int32_t*>(keys), values,
                       static_cast<int32_t*>(keys_out), values_out, static_cast<uint64_t*>(scratch));
  } else {
    static const hipError_t ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&segmented_sort_kernel<int16_t, 1024, CAP>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CAP * 8);
    GS_REQUIRE(ok == hipSuccess, GS_ERR_LAUNCH, "gs_segmented_sort_pairs: cannot reserve LDS");
    hipLaunchKernelGGL((segmented_sort_kernel<int16_t, 1024, CAP>), grid, dim3(1024), 2 * CAP * 8, s, int(num_segments),
                       start_offsets, end_offsets, static_cast<const int16_t*>(keys), values,
                       static_cast<int16_t*>(keys_out), values_out, static_cast<uint64_t*>(scratch));
  }
  GS_CHECK_LAUNCH("gs_segmented_sort_pairs");
  return GS_OK;
}

namespace {
__global__ __launch_bounds__(256) void detmath_kernel(int64_t n, const float* x, float* sqrt_out, float* log_out) {
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  if (sqrt_out) sqrt_out[i] = gs_det_sqrtf(x[i]);
  if (log_out) log_out[i] = gs_det_logf(x[i]);
}
}  // namespace

extern "C" int gs_selftest_detmath(int64_t n, const float* x, float* sqrt_out, float* log_out, void* stream) {
  if (n == 0) return GS_OK;
  GS_REQUIRE(x, GS_ERR_INVALID_ARGUMENT, "gs_selftest_detmath: x is NULL");
  hipLaunchKernelGGL(detmath_kernel, dim3(unsigned(gs_div_up(n, 256))), dim3(256), 0, static_cast<hipStream_t>(stream),
                     n, x, sqrt_out, log_out);
  GS_CHECK_LAUNCH("gs_selftest_detmath");
  return GS_OK;
}

extern "C" int gs_tile_count(int64_t v, const float* points, int32_t width, int32_t height,
                             const GsRasterConfig* cfg, int32_t* counts, void* stream) {
  MapArgs a;
  if (int rc = fill_args(a, v, points, nullptr, width, height, cfg, 0)) return rc;
  if (v == 0) return GS_OK;
  GS_REQUIRE(points && counts, GS_ERR_INVALID_ARGUMENT, "gs_tile_count: NULL buffer");
  hipLaunchKernelGGL(tile_count_kernel, dim3(unsigned(gs_div_up(v, 256))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a, counts);
  GS_CHECK_LAUNCH("gs_tile_count");
  return GS_OK;
}

extern "C" int64_t gs_cumsum_scratch_bytes(int64_t n) { return gs_align_up((gs_div_up(n, SCAN_BLOCK) + 1) * 4, 256); }

extern "C" int gs_full_cumsum_i32(int64_t n, const int32_t* in, int32_t* out, void* scratch, int64_t scratch_bytes,
                                  void* stream) {
  GS_REQUIRE(n >= 0 && out, GS_ERR_INVALID_ARGUMENT, "gs_full_cumsum_i32: bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n == 0) {
    if (hipMemsetAsync(out, 0, 4, s) != hipSuccess) { gs_set_error("gs_full_cumsum_i32: memset failed"); return GS_ERR_LAUNCH; }
    return GS_OK;
  }
  GS_REQUIRE(in && scratch && scratch_bytes >= gs_cumsum_scratch_bytes(n), GS_ERR_SCRATCH_TOO_SMALL,
             "gs_full_cumsum_i32: scratch %lld < %lld", (long long)scratch_bytes, (long long)gs_cumsum_scratch_bytes(n));
  const int nb = int(gs_div_up(n, SCAN_BLOCK));
  int* sums = static_cast<int*>(scratch);
  hipLaunchKernelGGL(scan_block_sums, dim3(nb), dim3(256), 0, s, n, in, sums);
  hipLaunchKernelGGL(scan_sums, dim3(1), dim3(1024), 0, s, nb, sums);
  hipLaunchKernelGGL(scan_apply, dim3(nb), dim3(256), 0, s, n, in, sums, out);
  GS_CHECK_LAUNCH("gs_full_cumsum_i32");
  return GS_OK;
}

extern "C" int gs_tile_emit_keys(int64_t v, const float* points, const float* depth, const int32_t* offsets,
                                 int32_t width, int32_t height, const GsRasterConfig* cfg, int32_t use_depth16,
                                 uint64_t* keys, int32_t* values, void* stream) {
  MapArgs a;
  if (int rc = fill_args(a, v, points, depth, width, height, cfg, use_depth16)) return rc;
  if (v == 0) return GS_OK;
  GS_REQUIRE(points && depth && offsets && keys && values, GS_ERR_INVALID_ARGUMENT, "gs_tile_emit_keys: NULL buffer");
  hipLaunchKernelGGL(tile_emit_keys_kernel, dim3(unsigned(gs_div_up(v, 256))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a, offsets, keys, values);
  GS_CHECK_LAUNCH("gs_tile_emit_keys");
  return GS_OK;
}

extern "C" int gs_find_ranges(int64_t k, const uint64_t* sorted_keys, int32_t use_depth16, int64_t num_tiles,
                              int32_t* tile_ranges, void* stream) {
  GS_REQUIRE(tile_ranges && num_tiles > 0, GS_ERR_INVALID_ARGUMENT, "gs_find_ranges: bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(tile_ranges, 0, size_t(num_tiles) * 8, s) != hipSuccess) {
    gs_set_error("gs_find_ranges: memset failed");
    return GS_ERR_LAUNCH;
  }
  if (k == 0) return GS_OK;
  GS_REQUIRE(sorted_keys, GS_ERR_INVALID_ARGUMENT, "gs_find_ranges: keys is NULL");
  hipLaunchKernelGGL(find_ranges_kernel, dim3(unsigned(gs_div_up(k, 256))), dim3(256), 0, s, k, sorted_keys,
                     use_depth16 ? 16 : 32, tile_ranges);
  GS_CHECK_LAUNCH("gs_find_ranges");
  return GS_OK;
}
