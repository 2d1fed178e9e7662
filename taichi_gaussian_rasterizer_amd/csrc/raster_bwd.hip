// raster_bwd.hip -- gradient backward of the alpha-blend (reference rasterizer/backward.py:53-228,
// pdf gradients taichi_lib/generic.py:321-336 / :371-404).
//
// Same wave-per-16x16-region layout, LDS staging and scalar sub-block masks as raster_fwd.hip.
// What is specific to the backward:
//   * Front-to-back re-traversal with the FINAL image as the "remaining colour"
//     (backward.py:110,177-180).  The remaining colour only ever appears dotted with the pixel's
//     upstream gradient, so each pixel carries ONE scalar R = sum_c rem_c * g_c instead of F
//     channels:  R -= w * (f . g);  dL/dalpha = T (f . g) - R / (1 - alpha).
//   * Lean kernel: the pdf gradients are linear in a few per-pixel moments, so each lane
//     accumulates 6 sums in the ellipse frame (G, G tx, G ty, G tx^2, G tx ty, G ty^2 with
//     G = p * dL/dalpha; tx, ty are O(1) so nothing cancels) + F feature sums over its <= 4 pixels;
//     the 7 splat gradients are formed from the wave totals once per splat, 64 splats at a time
//     (one lane per splat).
//   * Wave reduction by fused v_add_f32_dpp (row_ror 8/4/2/1, row_bcast 15/31): a workgroup is
//     exactly one wave64, so the whole-tile reduction needs no LDS atomics and no barrier
//     (the reference: warp shuffle -> shared atomic -> global atomic, concurrent.py:81-85).
//   * Flush: one 64-byte-aligned row per Gaussian in grad_rows; a wave atomic instruction
//     covers 4 splats x 16 contiguous floats (four 64-B memory-side atomic requests) instead of 64
//     scattered dwords -- the difference between ~1.3 TB/s and ~0.08 TB/s of atomic bytes on
//     MI355X (MI355X_MICROARCH.md, Global float atomics).
//
// Roofline: algorithmic HBM bytes 8T + K(4+28+4F) + 8PF + 4(7+F)K; VALU-bound.

#include "gs_common.h"

namespace {

struct BwdArgs {
  const float* points;
  const float* features;
  const int2* ranges;
  const int* o2p;
  const float* image;
  const float* grad_image;
  float* grad_rows;
  int W, H, F;
  int row_floats;
  int tiles_wide;
  int tile_size;
  int sub_x, sub_y;  // wave regions per tile along x / y
  int num_items;
  int num_tiles;
  const int* heavy;  // optional (device): the first *heavy entries of tile_order get four 8x8 workgroups each
  int heavy_cap;
  const int* tile_order;  // optional launch order of the items (heaviest first)
  float cmax, thr, inv_thr, sat;
  int aa, heur;
  GsShard sh;  // owned tile rows: tile ids are local, H is the full image height, the images hold the owned rows
};

// pixel origin of a (local) tile in the full image, and the row of the image buffers it starts at
__device__ __forceinline__ void tile_origin(const BwdArgs& a, int tile, int& x0, int& y0, int& yout0) {
  const int lty = tile / a.tiles_wide;
  x0 = (tile - lty * a.tiles_wide) * a.tile_size;
  y0 = gs_shard_global_row(a.sh, lty) * a.tile_size;
  yout0 = lty * a.tile_size;
}

__device__ __forceinline__ void s_sig_grad(float x, float inv_sigma, float& s, float& ds_dx, float& ds_dsig) {
  // taichi_lib/generic.py:360-369
  const float z = x * inv_sigma;
  s = gs_rcp_fast(1.0f + gs_exp2_fast((-1.6f * z - 0.07f * z * z * z) * 1.44269504088896341f));
  const float d = (1.6f + 0.21f * z * z) * s * (1.0f - s);
  ds_dx = d * inv_sigma;
  ds_dsig = ds_dx * -z;
}

// The antialiased pdf's sigmoid S(z) = 1 / (1 + exp(-(1.6 z + 0.07 z^3))) (taichi_lib/generic.py:341-369) and its
// derivative in 15 issue slots: a = S(z), d = dS/dz = (1.6 + 0.21 z^2) S (1 - S); the log2(e) factors are folded into
// the polynomial.  (S (1 - S) as a (1 - a), not e a^2: far out in the tail e overflows to inf while a is an exact 0.)
__device__ __forceinline__ void s_sig_parts(float z, float& a, float& d) {
  const float z2 = z * z;
  const float e = gs_exp2_fast(z * __builtin_fmaf(-0.07f * 1.44269504088896341f, z2, -1.6f * 1.44269504088896341f));
  a = gs_rcp_fast(1.0f + e);
  d = __builtin_fmaf(0.21f, z2, 1.6f) * (a * (1.0f - a));
}
// ... in two halves: the value alone decides whether a pixel takes anything from the splat; the derivative is only
// formed for the pixels that do (GS_BWD_HIT_EXEC: under their EXEC mask, skipped when the sub-block has none)
__device__ __forceinline__ float s_sig_value(float z) {
  const float e = gs_exp2_fast(z * __builtin_fmaf(-0.07f * 1.44269504088896341f, z * z, -1.6f * 1.44269504088896341f));
  return gs_rcp_fast(1.0f + e);
}
__device__ __forceinline__ float s_sig_slope(float z, float a) {
  return __builtin_fmaf(0.21f, z * z, 1.6f) * (a * (1.0f - a));
}

#ifndef GS_BWD_WAVES
#define GS_BWD_WAVES 1  // minimum waves per SIMD requested from the register allocator (1 = no constraint)
#endif

// LDS arena of one wave: sized by the 64-splat staging group, not by the wave's pixel region
// MODE 0: lean (6 ellipse-frame moments); 1: lean + the two densification heuristics (training with statistics);
// 2: full (7 gradients + 2 heuristics per pixel; antialiased pdf)
template <int FP, int MODE>
struct BwdShape {
  static constexpr bool FULL = MODE == 2;
  static constexpr int NS = MODE == 2 ? 9 : MODE == 1 ? 8 : 6;
  static constexpr int NACC = NS + FP;           // values reduced per splat
  static constexpr int ROW = ((9 + FP + 15) / 16) * 16;
  static constexpr int GEO_V4 = FULL ? 3 : 2;     // float4s of geometry per staged record ...
  static constexpr int REC_V4 = GEO_V4 + (FP + 3) / 4;  // ... followed by the feature row: one address, b128 reads
  // the per-splat totals are read back one lane per splat, so any ODD stride is conflict-free
  static constexpr int REC_F = 64 * 4 * REC_V4, ACC_STRIDE = NACC | 1, ACC_F = 64 * ACC_STRIDE;
  static constexpr int OUT_STRIDE = ROW + 1, OUT_F = 64 * OUT_STRIDE;
  static constexpr int ARENA_F = (REC_F + ACC_F) > OUT_F ? (REC_F + ACC_F) : OUT_F;
};

// The wave sum of the 9 per-splat values.  0 (shipped): the transposed register butterfly.  1 / 2 (round-3 experiments,
// kept buildable: tools/build_variant.sh lds2 raster_bwd -DGS_BWD_LDS_REDUCE=2): 8 of the values transposed through LDS
// in one 8-row pass / two 4-row passes.  In tools/ubench/mfma_reduce.hip, beside 80 plain v_fma, the LDS form costs
// 57 ns per call against 87 ns for the butterfly -- but in THIS kernel it is slower (0.67 vs 0.64 ms at C3, 6 waves per
// SIMD either way): the kernel already reads three b128 records per splat from LDS, and 8 ds_write_b32 + 2 ds_read_b128
// + 2 result writes more per (tile, splat) put the CU's LDS array at ~170 of the ~200 cycles four SIMDs spend on a
// splat -- the reduction leaves the VALU port only to queue at the LDS.  (v1, one pass with 8 rows, also costs a wave of
// occupancy: 1.167 vs 1.137 ms per frame.)
// The per-splat sums start at zero.  Written as `= 0.0f` the first evaluated sub-block of the unrolled chain initialises
// them for free -- and every splat whose first sub-block is masked out pays 9 v_mov_b32 (about 5 issue slots per
// overlap on average, in a kernel bound by VALU issue).  1: the zeros come from LDS instead -- ceil(NACC / 4) ds_read_b128
// of a 64-byte block of zeros at the head of every splat, on the LDS pipe, their latency under the first sub-block's
// coordinate / exponent arithmetic -- and every sub-block accumulates.  (Kernels with up to 16 sums per splat.)
#ifndef GS_BWD_ZERO_LDS
#define GS_BWD_ZERO_LDS 1
#endif
// The sub-block masks of the staged splats as four 64-bit ballots in scalar registers (bit j of ballot b: splat j reaches
// sub-block b, and b is still live): the walk tests them with scalar bit tests only -- no v_readfirstlane per splat, and
// the branch no longer waits for the splat's LDS record.
// The ninth of the nine per-splat sums does not fit the 8-value butterfly; it went through six fused DPP adds to row 3
// (gs_wave_reduce_transposed<9>).  1: four DPP adds leave every lane with its ROW's sum, and one lane per row adds that
// to the splat's total in LDS (ds_add_f32, four lanes on one address): two DPP adds, a move and a select less on the
// VALU port per (region, splat).  (Ending the 8-value butterfly the same way, two DPP stages early -- 32 lanes adding
// four partial sums per value -- is ruinous: 1.05 against 0.62 ms, profiles/r3/ab_butterfly_quarter_lds_add.txt; an LDS
// float add costs by the lane.)
#ifndef GS_BWD_NINTH_LDS
#define GS_BWD_NINTH_LDS 1
#endif
#ifndef GS_BWD_BPERMUTE
#define GS_BWD_BPERMUTE 0
#endif
#ifndef GS_BWD_HIT_EXEC
#define GS_BWD_HIT_EXEC 1
#endif
#ifndef GS_BWD_MASK_BALLOTS
#define GS_BWD_MASK_BALLOTS 1
#endif
#ifndef GS_BWD_FETCH_AHEAD
#define GS_BWD_FETCH_AHEAD 1
#endif
#ifndef GS_BWD_LDS_REDUCE
#define GS_BWD_LDS_REDUCE 0
#endif
constexpr int TR_STRIDE = GS_BWD_LDS_REDUCE == 1 ? 68 : 64;  // floats per value row of the transposition buffer
constexpr int TR_ROWS = GS_BWD_LDS_REDUCE == 1 ? 8 : 4;
// 3: hybrid -- values 0..3 in ONE 4-row LDS pass, values 4..8 through the 5-value butterfly

template <int NB, int FP, int MODE>
__device__ __forceinline__ void raster_bwd_body(const BwdArgs& a, int tile, int x0, int y0, int yout0, float* smem,
                                                float* s_tr) {
  const int lane = threadIdx.x;
  constexpr bool FULL = MODE == 2, HEUR = MODE == 1;
  constexpr int NS = MODE == 2 ? 9 : MODE == 1 ? 8 : 6;  // sums per splat besides the F feature gradients
  constexpr int NACC = NS + FP;           // values reduced per splat
  constexpr int ROW = ((9 + FP + 15) / 16) * 16;

  // One LDS arena per wave.  The staged records (geo, feat) and the per-splat totals (acc) are dead
  // by the time the gradient rows (out) are written, so `out` aliases them: ~7.4 KB per wave
  // instead of ~11.5 KB, i.e. 21 instead of 13 resident waves per CU.
  typedef BwdShape<FP, MODE> Shape;
  constexpr int GEO_V4 = Shape::GEO_V4, REC_V4 = Shape::REC_V4, ACC_STRIDE = Shape::ACC_STRIDE;
  constexpr int OUT_STRIDE = Shape::OUT_STRIDE;
  float4(*s_geo)[REC_V4] = reinterpret_cast<float4(*)[REC_V4]>(smem);
  float(*s_acc)[ACC_STRIDE] = reinterpret_cast<float(*)[ACC_STRIDE]>(smem + Shape::REC_F);
  float(*s_out)[OUT_STRIDE] = reinterpret_cast<float(*)[OUT_STRIDE]>(smem);

  // which value of the per-splat reduction this lane ends up owning (lane-constant; -1 = none)
  constexpr bool NINTH_LDS = GS_BWD_NINTH_LDS && NACC == 9 && !GS_BWD_LDS_REDUCE;
  constexpr int NBUTTERFLY = NINTH_LDS ? 8 : (NACC <= 16 ? NACC : 1);
  const int my_slot = (NACC <= 16 && (lane & 3) == 0) ? gs_reduce_slot<NBUTTERFLY>(lane) : -1;
  const int lx = lane & 7, ly = lane >> 3;
  // Tr = 1 - (accumulated weight): the transmittance in front of the next splat.  The reference carries the weight W
  // and tests W < saturate_threshold (backward.py:160); Tr > 1 - saturate_threshold is the same test, and every use of
  // W in the gradient is through 1 - W.
  const float tsat = 1.0f - a.sat;
  float Xf[NB], Yf[NB], Tr[NB], R[NB], gpix[NB][FP];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int X = x0 + (b & 1) * 8 + lx, Y = y0 + (b >> 1) * 8 + ly;
    const bool inb = X < a.W && Y < a.H;
    // lean modes: pixel centres relative to the wave's origin, as in the forward (raster_fwd.hip)
    Xf[b] = FULL ? float(X) + 0.5f : float((b & 1) * 8 + lx) + 0.5f;
    Yf[b] = FULL ? float(Y) + 0.5f : float((b >> 1) * 8 + ly) + 0.5f;
    Tr[b] = inb ? 1.0f : 0.0f;  // backward.py:99-112: out-of-image pixels start saturated
    R[b] = 0.0f;
#pragma unroll
    for (int c = 0; c < FP; ++c) gpix[b][c] = 0.0f;
    if (inb) {
      const int64_t pix = int64_t(Y - y0 + yout0) * a.W + X;
#pragma unroll
      for (int c = 0; c < FP; ++c)
        if (c < a.F) {
          gpix[b][c] = a.grad_image[pix * a.F + c];
          R[b] += a.image[pix * a.F + c] * gpix[b][c];
        }
    }
  }

  const int2 range_v = a.ranges[tile];
  // wave-uniform: keep the loop bounds in scalar registers
  const int range_x = __builtin_amdgcn_readfirstlane(range_v.x), range_y = __builtin_amdgcn_readfirstlane(range_v.y);
  // lean kernels stage the ellipse frame scaled by K_EXP so that the pdf is exp2(-(tx^2 + ty^2)) with no further
  // multiply; the moments are accumulated in those coordinates and unscaled once per splat in the epilogue
  constexpr float K_EXP = 0.84932180028801904f;  // sqrt(0.5 * log2(e))
  constexpr float IK = 1.0f / K_EXP, IK2 = IK * IK;

  int zero_off = 0;  // see GS_BWD_ZERO_LDS
  for (int g0 = range_x; g0 < range_y; g0 += 64) {
    // all pixels of the region saturated -> nothing further contributes (backward.py:116-118)
    // ... and a saturated 8x8 sub-block takes no gradient from here on (:160,166): masked out before its alphas
    // are even computed
    int live = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b)
      if (__ballot(Tr[b] > tsat) != 0ull) live |= 1 << b;
    if (live == 0) break;

    const int cnt = __builtin_amdgcn_readfirstlane(min(64, range_y - g0));
    float ax = 0, ay = 0, isx = 0, isy = 0, al = 0;
    int idx = 0;
    int staged_mask = 0;
    if (lane < cnt) {
      idx = a.o2p[g0 + lane];
      const float* p = a.points + int64_t(idx) * 7;
      const float mx = p[0], my = p[1];
      ax = p[2]; ay = p[3];
      isx = gs_rcp_fast(p[4]); isy = gs_rcp_fast(p[5]);  // v_rcp_f32: 1 ulp, far inside the parity tolerance
      al = p[6];
      const float ks = FULL ? 1.0f : K_EXP;
      const float Ax = ax * isx * ks, Ay = ay * isx * ks, Bx = -ay * isy * ks, By = ax * isy * ks;
      int mask = 0;
      if (FULL && a.aa) {
        float s1, s2, u0, u1;
        s_sig_grad(0.5f, isx, s1, u0, u1);
        s_sig_grad(0.5f, isy, s2, u0, u1);
        // D(0; s) = S(0.5 / s) - S(-0.5 / s) = 2 S(0.5 / s) - 1
        mask = gs_sub_block_mask_antialias<NB>(ax, ay, p[4], p[5], al, a.inv_thr, 2.0f * s1 - 1.0f, 2.0f * s2 - 1.0f,
                                               float(x0) + 0.5f - mx, float(y0) + 0.5f - my);
      } else if (al > a.thr) {
        // alpha * pdf > thr  needs  tx^2 + ty^2 < log2(alpha / thr)  (scaled frame; 2 ln(alpha / thr) unscaled)
        const float r2 = __log2f(al * a.inv_thr) * (FULL ? 1.38629436111989f : 1.0f);
        mask = gs_sub_block_mask<NB>(Ax, Ay, Bx, By, r2, float(x0) + 0.5f - mx, float(y0) + 0.5f - my);
      }
      staged_mask = mask;
      if (FULL && a.aa) {
        // antialiased pdf: the per-pixel code works in the splat's frame (ux, uy) and needs the sigmas and the half
        // pixel in sigma units, not the scaled ellipse frame
        s_geo[lane][0] = make_float4(mx, my, p[4], p[5]);
        s_geo[lane][1] = make_float4(0.5f * isx, 0.5f * isy, al, __int_as_float(mask));
      } else {
        // lean: tx = A . (X - origin) + A . (origin - m), the second term formed here (the forward's expression)
        const float ox = float(x0) - mx, oy = float(y0) - my;
        if (FULL) s_geo[lane][0] = make_float4(mx, my, Ax, Ay);
        else s_geo[lane][0] = make_float4(__builtin_fmaf(Ax, ox, Ay * oy), __builtin_fmaf(Bx, ox, By * oy), Ax, Ay);
        // lean modes carry -log2(opacity): it starts the exponent's fma chain, so v_exp_f32 returns alpha itself
        s_geo[lane][1] = make_float4(Bx, By, FULL ? al : -__log2f(al), __int_as_float(mask));
      }
      if (FULL) s_geo[lane][2] = make_float4(ax, ay, isx, isy);
      const float* f = a.features + int64_t(idx) * a.F;
#pragma unroll
      for (int q = 0; q < REC_V4 - GEO_V4; ++q) {
        float fv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) fv[k] = (4 * q + k < FP && 4 * q + k < a.F) ? f[4 * q + k] : 0.0f;
        s_geo[lane][GEO_V4 + q] = make_float4(fv[0], fv[1], fv[2], fv[3]);
      }
    }
#pragma unroll
    for (int c = 0; c < NACC; ++c) s_acc[lane][c] = 0.0f;
    __syncthreads();

    float4 g0v, g1v, g2v = make_float4(0, 0, 0, 0);
    float feat[FP];
    auto fetch_record = [&](int j) {
      g0v = s_geo[j][0];
      g1v = s_geo[j][1];
      if (FULL) g2v = s_geo[j][2];
#pragma unroll
      for (int q = 0; q < REC_V4 - GEO_V4; ++q) {
        const float4 fq = s_geo[j][GEO_V4 + q];
        const float fv[4] = {fq.x, fq.y, fq.z, fq.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (4 * q + k < FP) feat[4 * q + k] = fv[k];
      }
    };
    if (GS_BWD_FETCH_AHEAD) fetch_record(0);
    uint64_t reach[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
      reach[b] = (GS_BWD_MASK_BALLOTS && ((live >> b) & 1)) ? __ballot((staged_mask >> b) & 1) : 0ull;
    for (int j = 0; j < cnt; ++j) {
      if (!GS_BWD_FETCH_AHEAD) fetch_record(j);
      int mask = 0;
      if (GS_BWD_MASK_BALLOTS) {
#pragma unroll
        for (int b = 0; b < NB; ++b) mask |= int((reach[b] >> j) & 1ull) << b;
      } else {
        mask = __builtin_amdgcn_readfirstlane(__float_as_int(g1v.w)) & live;
      }

      float S[NS], gf[FP];
      if (GS_BWD_ZERO_LDS && !GS_BWD_LDS_REDUCE && NACC <= 16) {
        // (an opaque byte OFFSET, carried across the loop: an opaque pointer would lose its address space and become a
        // flat load; re-initialising it per splat would cost the v_mov this is about)
        asm volatile("" : "+v"(zero_off));
        const float4* zp = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_tr) + zero_off);
        float z[16];
#pragma unroll
        for (int q = 0; q < (NACC + 3) / 4; ++q) {
          const float4 zq = zp[q];
          z[4 * q] = zq.x; z[4 * q + 1] = zq.y; z[4 * q + 2] = zq.z; z[4 * q + 3] = zq.w;
        }
#pragma unroll
        for (int c = 0; c < NS; ++c) S[c] = z[c];
#pragma unroll
        for (int c = 0; c < FP; ++c) gf[c] = z[(NS + c) < 16 ? NS + c : 0];
      } else {
#pragma unroll
        for (int c = 0; c < NS; ++c) S[c] = 0.0f;
#pragma unroll
        for (int c = 0; c < FP; ++c) gf[c] = 0.0f;
      }
      bool any_grad = false;  // wave-uniform: some pixel of some sub-block took a gradient from this splat
      uint64_t hit_lanes = 0ull;

#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (!(mask & (1 << b))) continue;  // scalar branch
        const float dx = FULL ? Xf[b] - g0v.x : 0.0f, dy = FULL ? Yf[b] - g0v.y : 0.0f;
        float p, tx = 0, ty = 0;
        float dmx = 0, dmy = 0, dax = 0, day = 0, dsx = 0, dsy = 0;
        float Px = 0, Py = 0;  // antialias: d pdf / d (ux, uy), the splat-frame gradient the mean and axis terms share
        float aa_z[4] = {0, 0, 0, 0}, aa_a[4] = {0, 0, 0, 0};  // antialias: sigmoid arguments and values
        if (FULL && a.aa) {
          // taichi_lib/generic.py:341-404 in the splat's frame: u = R(axis) d, pdf = tau fx(ux) fy(uy) with
          // f(u; s) = s (S((u + .5) / s) - S((u - .5) / s)).  With a_k = S(z_k), d_k = S'(z_k), z_1,2 = (u +- .5) / s:
          //   df/du = d_1 - d_2,   df/ds = (a_1 - a_2) - (z_1 d_1 - z_2 d_2)
          // record: g0 = (mean, sx, sy), g1 = (.5 / sx, .5 / sy, alpha, mask), g2 = (axis, 1 / sx, 1 / sy)
          const float axv = g2v.x, ayv = g2v.y;
          const float ux = dx * axv + dy * ayv, uy = dy * axv - dx * ayv;
          const float zx1 = __builtin_fmaf(ux, g2v.z, g1v.x), zx2 = __builtin_fmaf(ux, g2v.z, -g1v.x);
          const float zy1 = __builtin_fmaf(uy, g2v.w, g1v.y), zy2 = __builtin_fmaf(uy, g2v.w, -g1v.y);
          aa_z[0] = zx1; aa_z[1] = zx2; aa_z[2] = zy1; aa_z[3] = zy2;
#pragma unroll
          for (int k = 0; k < 4; ++k) aa_a[k] = s_sig_value(aa_z[k]);
          const float Dx = aa_a[0] - aa_a[1], Dy = aa_a[2] - aa_a[3];
          const float fx = g0v.z * Dx, fy = g0v.w * Dy;
          p = 6.28318530717958648f * fx * fy;
          // the derivatives follow behind the hit test (aa_gradients below)
        } else {
          tx = FULL ? dx * g0v.z + dy * g0v.w : __builtin_fmaf(g0v.z, Xf[b], __builtin_fmaf(g0v.w, Yf[b], g0v.x));
          ty = FULL ? dx * g1v.x + dy * g1v.y : __builtin_fmaf(g1v.x, Xf[b], __builtin_fmaf(g1v.y, Yf[b], g0v.y));
          // lean: the record carries -log2(opacity), p is alpha itself (same expression as the forward: same bits)
          p = FULL ? gs_exp2_fast(-0.72134752044448170f * (tx * tx + ty * ty))
                   : gs_exp2_fast(-__builtin_fmaf(ty, ty, __builtin_fmaf(tx, tx, g1v.z)));
          if (FULL) {
            // taichi_lib/generic.py:321-336
            const float txs = tx * g2v.z, tys = ty * g2v.w;
            dsx = tx * tx * p * g2v.z;
            dsy = ty * ty * p * g2v.w;
            dax = p * (txs * -dx + tys * -dy);
            day = p * (txs * -dy + tys * dx);
            dmx = p * (txs * g2v.x - tys * g2v.y);
            dmy = p * (txs * g2v.y + tys * g2v.x);
          }
        }
        const float alpha_raw = FULL ? g1v.z * p : p;
        const bool over = alpha_raw > a.thr, open = Tr[b] > tsat;
        const bool hit = over && open;  // backward.py:160,166
        // (two ballots of plain compares and a scalar AND: a ballot of the conjunction is rebuilt through a VGPR)
        if (GS_BWD_HIT_EXEC) {
          hit_lanes |= __ballot(over) & __ballot(open);  // scalar; the EXEC-masked block below skips itself when empty
        } else {
          if ((__ballot(over) & __ballot(open)) == 0ull) continue;
          any_grad = true;
        }
        // Everything below is linear in the pixel's alpha and touches nothing but the lane's own sums and state: it runs
        // under EXEC = the pixels that take something from this splat (a partly empty EXEC costs a wave64 instruction
        // nothing extra), which saves the select that used to zero alpha for the others.  (GS_BWD_HIT_EXEC = 0: the select.)
        if (GS_BWD_HIT_EXEC && !hit) continue;
        const float a_hit = GS_BWD_HIT_EXEC ? alpha_raw : (hit ? alpha_raw : 0.0f);
        if (FULL && a.aa) {  // aa_gradients: see the value half above
          const float d0 = s_sig_slope(aa_z[0], aa_a[0]), d1 = s_sig_slope(aa_z[1], aa_a[1]);
          const float d2 = s_sig_slope(aa_z[2], aa_a[2]), d3 = s_sig_slope(aa_z[3], aa_a[3]);
          const float Dx = aa_a[0] - aa_a[1], Dy = aa_a[2] - aa_a[3];
          const float tau = 6.28318530717958648f;
          const float fxt = tau * (g0v.z * Dx), fyt = tau * (g0v.w * Dy);
          Px = (d0 - d1) * fyt;
          Py = fxt * (d2 - d3);
          dsx = (Dx - __builtin_fmaf(aa_z[0], d0, -(aa_z[1] * d1))) * fyt;
          dsy = fxt * (Dy - __builtin_fmaf(aa_z[2], d2, -(aa_z[3] * d3)));
          dax = __builtin_fmaf(Px, dx, Py * dy);
          day = __builtin_fmaf(Px, dy, -(Py * dx));
          // dmx, dmy: the wave totals of aag Px, aag Py are rotated out of the frame once per splat (epilogue)
        }
        const float alc = __builtin_amdgcn_fmed3f(a_hit, a.cmax, -1.0f);  // min(alpha, cmax), one v_med3_f32 (:169)
        float dot = 0.0f;
#pragma unroll
        for (int c = 0; c < FP; ++c) dot += feat[c] * gpix[b][c];
        // dL/dalpha = sum_c (f_c T - rem_c / (1 - alpha)) g_c   (:180-182) with rem = R - w dot, w = alpha T:
        //           = (T dot - R) / (1 - alpha), R still including this splat's share.  The numerator is formed before
        // T and R are updated so that both updates happen in place (no register copies at the end of the block).
        const float num = Tr[b] * dot - R[b];
        const float w = alc * Tr[b];
        float alpha_grad = num * gs_rcp_fast(1.0f - alc);
        if (!GS_BWD_HIT_EXEC && (FULL || HEUR)) alpha_grad = hit ? alpha_grad : 0.0f;  // used without the a_hit factor
#pragma unroll
        for (int c = 0; c < FP; ++c) gf[c] += w * gpix[b][c];  // :201
        if (FULL) {
          const float aag = g1v.z * alpha_grad;  // :184
          if (a.aa) {
            S[0] += aag * Px; S[1] += aag * Py;  // splat frame; see the epilogue
            if (a.heur) {
              dmx = -Px * g2v.x + Py * g2v.y;
              dmy = -Px * g2v.y - Py * g2v.x;
            }
          } else {
            S[0] += aag * dmx; S[1] += aag * dmy;
          }
          S[2] += aag * dax; S[3] += aag * day;
          S[4] += aag * dsx; S[5] += aag * dsy;
          S[6] += p * alpha_grad;
          if (a.heur) {
            S[7] += aag * aag;                                  // :194-198
            S[8] += fabsf(aag * dmx) + fabsf(aag * dmy);
          }
        } else {
          // moments in the (scaled) ellipse frame (tx, ty are O(1): no cancellation for elongated splats), carrying
          // the splat's opacity: G = alpha_p * pdf * dL/dalpha
          const float G = a_hit * alpha_grad;
          const float Gtx = G * tx, Gty = G * ty;
          S[0] += G;
          S[1] += Gtx; S[2] += Gty;
          S[3] += Gtx * tx; S[4] += Gtx * ty; S[5] += Gty * ty;
          if (HEUR) {
            // backward.py:194-198 from the lean record: dp/dmean = p (tx A + ty B), A = axis / sx, B = perp(axis) / sy
            S[6] += alpha_grad * alpha_grad;  // hit-masked in this mode; the epilogue applies alpha_p^2
            // |alpha_p dL/dalpha dp/dmean|_1 = |G| (|tx A.x + ty B.x| + |tx A.y + ty B.y|) / K_EXP^2 with
            // G = alpha_p pdf dL/dalpha as above (the staged frame and tx, ty both carry K_EXP; the constant is
            // applied once per splat in the epilogue)
            S[7] += fabsf(G) * (fabsf(tx * g0v.z + ty * g1v.x) + fabsf(tx * g0v.w + ty * g1v.y));
          }
        }
        // T -= alpha T and R -= w dot, last and in place: left to the scheduler, the new values are formed early in
        // temporaries (the old ones are still needed for `num`) and copied back with two v_mov_b32 per block
        asm("v_fma_f32 %0, -%0, %1, %0" : "+v"(Tr[b]) : "v"(alc));
        asm("v_fma_f32 %0, -%1, %2, %0" : "+v"(R[b]) : "v"(dot), "v"(w));
      }

      // The record's registers are dead from here to the end of the iteration: the next splat's record is fetched into
      // them NOW, so that its LDS latency passes under the reduction below instead of in front of the next splat's
      // first instruction (no second register set, no copies).
      if (GS_BWD_FETCH_AHEAD) fetch_record(min(j + 1, 63));
      // reduce over the wave only if some pixel took a gradient (backward.py:204)
      if (GS_BWD_HIT_EXEC) any_grad = hit_lanes != 0ull;
      if (any_grad) {
        // transposed butterfly over the wave, sized for the exact number of values; the lane that ends
        // up owning value k stores it (one ds_write_b32 for all values of a chunk)
        if (GS_BWD_LDS_REDUCE == 3 && NACC == 9) {
          float vals[9];
#pragma unroll
          for (int c = 0; c < 9; ++c) vals[c] = c < NS ? S[c < NS ? c : 0] : gf[c >= NS ? c - NS : 0];
#pragma unroll
          for (int c = 0; c < 4; ++c) s_tr[c * TR_STRIDE + lane] = vals[c];
          float w5[5];
#pragma unroll
          for (int c = 0; c < 5; ++c) w5[c] = vals[4 + c];
          const float tot5 = gs_wave_reduce_transposed<5>(w5, lane);
          const int slot5 = (lane & 3) == 0 ? gs_reduce_slot<5>(lane) : -1;
          __syncthreads();
          const float4 u = *reinterpret_cast<const float4*>(s_tr + (lane >> 4) * TR_STRIDE + (lane & 15) * 4);
          float t = (u.x + u.y) + (u.z + u.w);
          t = gs_dpp_add_full<0xB1>(t);
          t = gs_dpp_add_full<0x4E>(t);
          t = gs_dpp_add_full<0x141>(t);
          t = gs_dpp_add_full<0x128>(t);
          if ((lane & 15) == 0) s_acc[j][lane >> 4] = t;
          if (slot5 >= 0) s_acc[j][4 + slot5] = tot5;
          __syncthreads();
        } else if (GS_BWD_LDS_REDUCE == 2 && NACC == 9) {
          // experiment (see GS_BWD_LDS_REDUCE above): in two passes of four values every lane stores its partial sums
          // as rows (value, lane), lane (c = lane >> 4, s = lane & 15) adds columns 4 s .. 4 s + 3 of row c -- one
          // conflict-free ds_read_b128 -- and four DPP adds fold the sixteen lanes of a value.  2 x (3 adds + 4 DPP) +
          // the ninth value's six DPP adds against ~58 issue slots; the wave's own LDS accesses execute in order, so
          // only the compiler needs the fences.
          float vals[9];
#pragma unroll
          for (int c = 0; c < 9; ++c) vals[c] = c < NS ? S[c < NS ? c : 0] : gf[c >= NS ? c - NS : 0];
          float x = vals[8];
          x = gs_dpp_add_full<0x128>(x);
          x = gs_dpp_add_full<0x124>(x);
          x = gs_dpp_add_full<0x122>(x);
          x = gs_dpp_add_full<0x121>(x);
          x = gs_dpp_add_full<0x142>(x);
          x = gs_dpp_add_full<0x143>(x);
          if (lane == 60) s_acc[j][8] = x;
#pragma unroll
          for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int c = 0; c < 4; ++c) s_tr[c * TR_STRIDE + lane] = vals[4 * pass + c];
            __syncthreads();
            const float4 u = *reinterpret_cast<const float4*>(s_tr + (lane >> 4) * TR_STRIDE + (lane & 15) * 4);
            float t = (u.x + u.y) + (u.z + u.w);
            t = gs_dpp_add_full<0xB1>(t);   // quad_perm:[1,0,3,2]
            t = gs_dpp_add_full<0x4E>(t);   // quad_perm:[2,3,0,1]
            t = gs_dpp_add_full<0x141>(t);  // row_half_mirror
            t = gs_dpp_add_full<0x128>(t);  // row_ror:8: all sixteen lanes of the row hold the value's total
            if ((lane & 15) == 0) s_acc[j][4 * pass + (lane >> 4)] = t;
            __syncthreads();
          }
        } else if (GS_BWD_LDS_REDUCE == 1 && NACC == 9) {
          // Transposed through LDS instead of the register butterfly: every lane stores its 8 partial sums (row c =
          // value c, column = lane), then lane (c = lane >> 3, s = lane & 7) adds columns 8 s .. 8 s + 7 of row c and
          // three DPP adds fold the eight lanes of a value.  7 adds + 3 DPP against ~42 issue slots of swaps and DPP;
          // the wave's own LDS accesses execute in order, so nothing but the compiler needs a fence.
          float vals[9];
#pragma unroll
          for (int c = 0; c < 9; ++c) vals[c] = c < NS ? S[c < NS ? c : 0] : gf[c >= NS ? c - NS : 0];
#pragma unroll
          for (int c = 0; c < 8; ++c) s_tr[c * TR_STRIDE + lane] = vals[c];
          float x = vals[8];
          x = gs_dpp_add_full<0x128>(x);
          x = gs_dpp_add_full<0x124>(x);
          x = gs_dpp_add_full<0x122>(x);
          x = gs_dpp_add_full<0x121>(x);
          x = gs_dpp_add_full<0x142>(x);
          x = gs_dpp_add_full<0x143>(x);
          __syncthreads();
          const float4* rowp = reinterpret_cast<const float4*>(s_tr + (lane >> 3) * TR_STRIDE + (lane & 7) * 8);
          const float4 u0 = rowp[0], u1 = rowp[1];
          float t = ((u0.x + u0.y) + (u0.z + u0.w)) + ((u1.x + u1.y) + (u1.z + u1.w));
          t = gs_dpp_add_full<0xB1>(t);   // quad_perm:[1,0,3,2]
          t = gs_dpp_add_full<0x4E>(t);   // quad_perm:[2,3,0,1]
          t = gs_dpp_add_full<0x141>(t);  // row_half_mirror: the other quad of the 8-lane group
          if ((lane & 7) == 0) s_acc[j][lane >> 3] = t;
          if (lane == 60) s_acc[j][8] = x;
          __syncthreads();
        } else if (NINTH_LDS) {
          float w8[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) w8[c] = c < NS ? S[c < NS ? c : 0] : gf[c >= NS ? c - NS : 0];
          float x = gf[FP - 1];  // value 8
          x = gs_dpp_add_full<0x128>(x);  // row_ror:8
          x = gs_dpp_add_full<0x124>(x);  // row_ror:4
          x = gs_dpp_add_full<0x122>(x);  // row_ror:2
          x = gs_dpp_add_full<0x121>(x);  // row_ror:1 -> every lane holds its row's sum
          const float tot = GS_BWD_BPERMUTE ? gs_wave_reduce_transposed8_bpermute(w8, lane)
                                            : gs_wave_reduce_transposed<8>(w8, lane);
          if (my_slot >= 0) s_acc[j][my_slot] = tot;
          if ((lane & 15) == 0)
            __hip_atomic_fetch_add(&s_acc[j][8], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // 0 at staging
        } else if (NACC <= 16) {
          float vals[NACC <= 16 ? NACC : 1];
#pragma unroll
          for (int c = 0; c < NACC && c < 16; ++c) vals[c] = c < NS ? S[c < NS ? c : 0] : gf[c >= NS ? c - NS : 0];
          const float tot = gs_wave_reduce_transposed<(NACC <= 16 ? NACC : 1)>(vals, lane);
          if (my_slot >= 0) s_acc[j][my_slot] = tot;
        } else {
#pragma unroll
          for (int base = 0; base < NACC; base += 16) {
            float vals[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
              const int k = base + c;
              vals[c] = k < NS ? S[k < NS ? k : 0] : (k < NACC ? gf[(k - NS) < FP && k >= NS ? (k - NS) : 0] : 0.0f);
            }
            const float tot = gs_wave_reduce16_transposed(vals, lane);
            const int k = base + (lane >> 2);
            if ((lane & 3) == 0 && k < NACC) s_acc[j][k] = tot;
          }
        }
      }
    }
    __syncthreads();

    // ---- per-splat epilogue: lane j turns splat j's totals into its gradient row
    {
      float row[ROW];
#pragma unroll
      for (int c = 0; c < ROW; ++c) row[c] = 0.0f;
      if (lane < cnt) {
        float t[NACC];
#pragma unroll
        for (int c = 0; c < NACC; ++c) t[c] = s_acc[lane][c];
        if (FULL) {
#pragma unroll
          for (int c = 0; c < 7; ++c) row[c] = t[c];
          if (a.aa) {  // the mean's gradient out of the splat frame: d ux / d mean = -axis, d uy / d mean = -perp(axis)
            row[0] = -t[0] * ax + t[1] * ay;
            row[1] = -t[0] * ay - t[1] * ax;
          }
          if (a.heur) { row[7 + FP] = t[7]; row[8 + FP] = t[8]; }
        } else {
          // t = wave totals of (G, G tx, G ty, G tx^2, G tx ty, G ty^2), tx = d.axis/sx, ty = d.perp(axis)/sy.
          // With d = (sx tx axis + sy ty perp(axis)) / |axis|^2 the position-weighted sums follow:
          //   sum G tx d = (sx Mxx axis + sy Mxy perp) / n2 ,  sum G ty d = (sx Mxy axis + sy Myy perp) / n2
          // and dp/dmean, dp/daxis, dp/dsigma (taichi_lib/generic.py:321-336) become
          // the totals carry alpha_p (G = alpha_p pdf dL/dalpha) and powers of K_EXP (scaled tx, ty): undo both here
          t[1] *= IK; t[2] *= IK;
          t[3] *= IK2; t[4] *= IK2; t[5] *= IK2;
          const float sx = gs_rcp_fast(isx), sy = gs_rcp_fast(isy);
          const float in2 = gs_rcp_fast(ax * ax + ay * ay);
          const float txdx = (sx * ax * t[3] - sy * ay * t[4]) * in2, txdy = (sx * ay * t[3] + sy * ax * t[4]) * in2;
          const float tydx = (sx * ax * t[4] - sy * ay * t[5]) * in2, tydy = (sx * ay * t[4] + sy * ax * t[5]) * in2;
          row[0] = t[1] * isx * ax - t[2] * isy * ay;
          row[1] = t[1] * isx * ay + t[2] * isy * ax;
          row[2] = -(isx * txdx + isy * tydy);
          row[3] = isy * tydx - isx * txdy;
          row[4] = t[3] * isx;
          row[5] = t[5] * isy;
          // G carries the opacity; a listed splat at or below the threshold (a caller's own tile lists may hold one:
          // opacity zeroed after map_to_tiles) blended nothing, and 0 * rcp(0) must not become NaN
          row[6] = al > a.thr ? t[0] * gs_rcp_fast(al) : 0.0f;
          if (HEUR) { row[7 + FP] = t[6] * al * al; row[8 + FP] = t[7] * IK2; }
        }
#pragma unroll
        for (int c = 0; c < FP; ++c) row[7 + c] = t[NS + c];
        // the Gaussian's row index (still in this lane's register from the staging) rides in the row's last, unused
        // word to the flush below: no index array in LDS
        static_assert(9 + FP < ROW, "the gradient row needs a spare word");
        row[ROW - 1] = __int_as_float(idx);
      }
      __syncthreads();  // `out` aliases the totals just read
#pragma unroll
      for (int c = 0; c < ROW; ++c) s_out[lane][c] = row[c];
    }
    __syncthreads();
    // ---- flush: 64/ROW splats x ROW contiguous floats per wave atomic instruction
    for (int e = lane; e < cnt * ROW; e += 64) {
      const int jj = e / ROW, comp = e - jj * ROW;
      const float val = s_out[jj][comp];
      // heuristics live at [7+F, 9+F) of the caller's row; the kernel's padded slot is 7+FP
      int dst = comp;
      if (comp >= 9 + FP) continue;  // padding, and the index word
      if (comp >= 7 + FP) dst = comp - FP + a.F;
      else if (comp >= 7 + a.F) continue;
      if (val != 0.0f)
        atomicAdd(a.grad_rows + int64_t(__float_as_int(s_out[jj][ROW - 1])) * a.row_floats + dst, val);
    }
    __syncthreads();
  }
}

// Block -> work: see raster_fwd_kernel (the mapper's fullest tiles get one workgroup per 8x8 quadrant).
template <int NB, int FP, int MODE>
__global__ __launch_bounds__(64, GS_BWD_WAVES) void raster_bwd_kernel(const BwdArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[BwdShape<FP, MODE>::ARENA_F];
  // transposition buffer of the lean F = 3 reduction (the only shape that uses it)
  // (GS_BWD_ZERO_LDS, butterfly reduction: the same pointer carries the 64-byte block of zeros instead)
  __shared__ __attribute__((aligned(16))) float s_tr_buf[(GS_BWD_LDS_REDUCE && FP == 3 && MODE == 0) ? TR_ROWS * TR_STRIDE + 16 : 16];
  float* s_tr = s_tr_buf;
  if (GS_BWD_ZERO_LDS && !GS_BWD_LDS_REDUCE && threadIdx.x < 16) s_tr_buf[threadIdx.x] = 0.0f;  // read after the first staging barrier
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
      if (x0 < a.W && y0 < a.H) raster_bwd_body<1, FP, MODE>(a, tile, x0, y0, yout0, smem, s_tr);
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
  raster_bwd_body<NB, FP, MODE>(a, tile, x0, y0, yout0, smem, s_tr);
}

template <int NB, int MODE>
int launch_fp(const BwdArgs& a, hipStream_t s) {
  const int grid = 8 * int(gs_div_up(a.num_items + (a.heavy ? 4 * a.heavy_cap : 0), 8));
  if (a.F <= 3) hipLaunchKernelGGL((raster_bwd_kernel<NB, 3, MODE>), dim3(grid), dim3(64), 0, s, a);
  else if (a.F <= 5) hipLaunchKernelGGL((raster_bwd_kernel<NB, 5, MODE>), dim3(grid), dim3(64), 0, s, a);
  else if (a.F <= 8) hipLaunchKernelGGL((raster_bwd_kernel<NB, 8, MODE>), dim3(grid), dim3(64), 0, s, a);
  else hipLaunchKernelGGL((raster_bwd_kernel<NB, 32, MODE>), dim3(grid), dim3(64), 0, s, a);
  GS_CHECK_LAUNCH("gs_raster_bwd");
  return GS_OK;
}

__global__ void unpack_kernel(int64_t v, int F, int row_floats, const float* rows, float* gp, float* gf, float* heur) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int per = 9 + F;
  if (i >= v * per) return;
  const int64_t r = i / per;
  const int c = int(i - r * per);
  const float val = rows[r * row_floats + c];
  if (c < 7) { if (gp) gp[r * 7 + c] = val; }
  else if (c < 7 + F) { if (gf) gf[r * F + (c - 7)] = val; }
  else if (heur) heur[r * 2 + (c - 7 - F)] = val;
}

}  // namespace

extern "C" int32_t gs_grad_row_floats(int32_t num_features) { return int32_t(gs_align_up(9 + num_features, 16)); }

extern "C" int gs_raster_bwd(int64_t v, int32_t num_features, const float* points, const float* features,
                             const int32_t* tile_ranges, const int32_t* overlap_to_point, int64_t k, int32_t width,
                             int32_t height, const GsRasterConfig* cfg, const int32_t* tile_order,
                             const int32_t* heavy_tiles, const float* image, const float* grad_image,
                             float* grad_rows, const GsRowShard* shard, void* stream) {
  if (int rc = gs_check_cfg(cfg)) return rc;
  GS_REQUIRE(width > 0 && height > 0, GS_ERR_INVALID_ARGUMENT, "gs_raster_bwd: image size %dx%d", width, height);
  GS_REQUIRE(num_features >= 1 && num_features <= GS_MAX_FEATURES, GS_ERR_UNSUPPORTED,
             "gs_raster_bwd: feature width %d not in [1,%d]", num_features, GS_MAX_FEATURES);
  GS_REQUIRE(image && grad_image && tile_ranges, GS_ERR_INVALID_ARGUMENT, "gs_raster_bwd: NULL image or ranges");
  GS_REQUIRE(cfg->use_alpha_blending, GS_ERR_UNSUPPORTED,
             "gs_raster_bwd: no gradient is defined without alpha blending (reference tests/test_rasterizer.py:92-101)");
  if (k == 0 || v == 0) return GS_OK;
  GS_REQUIRE(points && features && overlap_to_point && grad_rows, GS_ERR_INVALID_ARGUMENT,
             "gs_raster_bwd: NULL input");
  const int ts = cfg->tile_size;
  BwdArgs a;
  a.points = points; a.features = features; a.ranges = reinterpret_cast<const int2*>(tile_ranges);
  a.o2p = overlap_to_point; a.image = image; a.grad_image = grad_image; a.grad_rows = grad_rows;
  a.W = width; a.H = height; a.F = num_features;
  a.row_floats = gs_grad_row_floats(num_features);
  a.tiles_wide = int(gs_div_up(width, ts));
  a.tile_size = ts;
  if (int rc = gs_make_shard(shard, int(gs_div_up(height, ts)), &a.sh)) return rc;
  const int num_tiles = a.tiles_wide * a.sh.local_rows;
  if (num_tiles == 0) return GS_OK;
  const int nb = gs_raster_sub_blocks(cfg, num_tiles, 1);
  a.sub_x = ts / (nb == 1 ? 8 : 16);
  a.sub_y = ts / (nb == 4 ? 16 : 8);
  a.num_items = num_tiles * a.sub_x * a.sub_y;
  a.tile_order = tile_order;
  a.num_tiles = num_tiles;
  a.heavy = (tile_order && ts == 16 && nb > 1) ? heavy_tiles : nullptr;
  a.heavy_cap = num_tiles / 4;
  if (cfg->tune_no_heavy_split) a.heavy = nullptr;
  a.cmax = cfg->clamp_max_alpha; a.thr = cfg->alpha_threshold; a.sat = cfg->saturate_threshold;
  a.inv_thr = 1.0f / cfg->alpha_threshold;
  a.aa = cfg->antialias; a.heur = cfg->compute_point_heuristic;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int mode = a.aa ? 2 : a.heur ? 1 : 0;
  if (nb == 1) return mode == 2 ? launch_fp<1, 2>(a, s) : mode == 1 ? launch_fp<1, 1>(a, s) : launch_fp<1, 0>(a, s);
  if (nb == 2) return mode == 2 ? launch_fp<2, 2>(a, s) : mode == 1 ? launch_fp<2, 1>(a, s) : launch_fp<2, 0>(a, s);
  return mode == 2 ? launch_fp<4, 2>(a, s) : mode == 1 ? launch_fp<4, 1>(a, s) : launch_fp<4, 0>(a, s);
}

extern "C" int gs_raster_bwd_unpack(int64_t v, int32_t num_features, const float* grad_rows, float* grad_points,
                                    float* grad_features, float* point_heuristic, void* stream) {
  if (v == 0) return GS_OK;
  GS_REQUIRE(grad_rows, GS_ERR_INVALID_ARGUMENT, "gs_raster_bwd_unpack: grad_rows is NULL");
  const int64_t total = v * (9 + num_features);
  hipLaunchKernelGGL(unpack_kernel, dim3(unsigned(gs_div_up(total, 256))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), v, num_features, gs_grad_row_floats(num_features), grad_rows,
                     grad_points, grad_features, point_heuristic);
  GS_CHECK_LAUNCH("gs_raster_bwd_unpack");
  return GS_OK;
}
