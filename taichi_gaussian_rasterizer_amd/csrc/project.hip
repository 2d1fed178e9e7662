// project.hip -- perspective projection + EWA 3D->2D covariance + cull + compaction, and its
// hand-derived adjoint.  Reference: perspective/projection.py:32-80 (project_kernel), :84-118
// (indexed_project_kernel, differentiated by Taichi autodiff at :175-180), math in
// taichi_lib/generic.py:96-158, :217-237, :419-427; ndc depth torch_lib/projection.py:120-123.
//
// MI355X notes: one lane per Gaussian, inputs read once (44 B/Gaussian); the camera (16+4 floats)
// is read through wave-uniform scalar loads instead of the reference's per-point expanded copies
// (projection.py:212-213: +64 B/Gaussian).  Compaction (the reference's torch.nonzero + two
// gathers, :146-149) is a ballot/popcount rank inside each 256-lane block plus one scan of the
// per-block counts; ndc depth and the int64 index list come out of the same pass.
//
// Roofline (HBM): forward reads 44 N, writes 36 N staging + reads it back + 48 V out;
// backward reads 44 N + 36 V, writes 44 N.

#include "gs_common.h"
#include "../../include/gs_detmath.h"
#include "project_math.h"

namespace {

using namespace gs_proj;

// pass 1: project everything, stage rows, count visible per block
// zero_words: the tile mapper's region counters, which the compaction pass adds into when it also does the mapper's
// binning (frame calls): cleared here, one pass earlier, instead of by a memset launch
__global__ __launch_bounds__(256) void project_kernel(ProjArgs a, float4* st_rows, int* block_counts, float* cam_out,
                                                      int* zero_words, int zero_count) {
  __shared__ int s_cnt[4];
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (cam_out && blockIdx.x == 0 && threadIdx.x == 0) camera_position(a.T44, cam_out);
  if (zero_words && blockIdx.x == 0)
    for (int e = threadIdx.x; e < zero_count; e += 256) zero_words[e] = 0;
  bool vis = false;
  if (i < a.n) {
    const Cam c = load_cam(a.T44, a.proj);
    Fwd f;
    forward(a, c, i, f);
    vis = visible(a, f);
    const float z = f.cam[2];
    st_rows[2 * i] = make_float4(f.u, f.v, f.ax, f.ay);
    st_rows[2 * i + 1] = make_float4(f.s1, f.s2, f.alpha, vis ? z : 0.0f);  // depth 0 = culled (:69-70)
  }
  const uint64_t b = __ballot(vis);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// pass 2: stable compaction
// block_offsets == nullptr: every workgroup sums the visible counts of the workgroups before it itself
// (block_counts, a few KB that stay in L2) instead of reading a prefix computed by three scan launches
__global__ __launch_bounds__(256) void compact_kernel(int64_t n, const float4* st_rows, const int* block_offsets,
                                                      const int* block_counts, int num_blocks, float inv_far, float ndc_denom, float* points,
                                                      float* depth, float* ndc, int64_t* indexes, int* slot_of,
                                                      int* num_visible, float* depth_feat, int depth_feat_stride,
                                                      float4* zero_rows, int zero_row_v4) {
  __shared__ int s_cnt[4];
  __shared__ int s_before[4];
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  float4 r0 = make_float4(0, 0, 0, 0), r1 = r0;
  bool vis = false;
  if (i < n) {
    r0 = st_rows[2 * i];
    r1 = st_rows[2 * i + 1];
    vis = r1.w != 0.0f;
  }
  const uint64_t b = __ballot(vis);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) s_cnt[wave] = __popcll(b);
  int before = 0;
  if (!block_offsets) {
    for (int j = threadIdx.x; j < int(blockIdx.x); j += 256) before += block_counts[j];
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off);
    if (lane == 0) s_before[wave] = before;
  }
  __syncthreads();
  int base = block_offsets ? block_offsets[blockIdx.x] : s_before[0] + s_before[1] + s_before[2] + s_before[3];
  const int block_start = base;
  for (int w = 0; w < wave; ++w) base += s_cnt[w];
  if (i < n) {
    int slot = -1;
    if (vis) {
      slot = base + __popcll(b & ((1ull << lane) - 1ull));
      float* p = points + int64_t(slot) * 7;
      p[0] = r0.x; p[1] = r0.y; p[2] = r0.z; p[3] = r0.w; p[4] = r1.x; p[5] = r1.y; p[6] = r1.z;
      depth[slot] = r1.w;
      if (depth_feat) {  // renderer.py:191-193: raster features [z, z^2, ...]
        depth_feat[int64_t(slot) * depth_feat_stride] = r1.w;
        depth_feat[int64_t(slot) * depth_feat_stride + 1] = r1.w * r1.w;
      }
      // fixed f32 op order: the sort key is the bit pattern of this value
      const float inv_d = __fdiv_rn(1.0f, r1.w);
      ndc[slot] = 1.0f - __fdiv_rn(inv_d - inv_far, ndc_denom);
      indexes[slot] = i;
    }
    slot_of[i] = slot;
  }
  // the frame's gradient rows (gs_raster_bwd accumulates into them with atomics): zero-filled here, by the pass that
  // already streams the V compact rows, instead of by a fill launch in front of the backward.  The workgroup's rows are
  // one contiguous range, cleared with consecutive 16-byte stores (a lane clearing its own 64-byte row costs 12 us more)
  if (zero_rows) {
    const int mine = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    float4* dst = zero_rows + int64_t(block_start) * zero_row_v4;
    for (int e = threadIdx.x; e < mine * zero_row_v4; e += 256) dst[e] = make_float4(0, 0, 0, 0);
  }
  if (int(blockIdx.x) == num_blocks - 1 && threadIdx.x == 0)
    *num_visible = block_start + s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// ------------------------------------------------------------------------------- backward
struct BwdArgs {
  ProjArgs f;
  const int* slot_of;
  const float* gpoints;  // row stride gpoints_stride, or null
  const float* gdepth;   // stride gdepth_stride, or null
  const float* gdepth_sq;  // optional gradient of a z^2 feature (adds 2 z g), same stride as gdepth
  int gpoints_stride, gdepth_stride;
  float* d_position;
  float* d_log_scaling;
  float* d_rotation;
  float* d_alpha_logit;
  float* cam_partials;  // (num_blocks,16) or null
};

template <bool CAMERA>
__global__ __launch_bounds__(256) void project_bwd_kernel(BwdArgs a) {
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  float gcam_acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) gcam_acc[k] = 0.0f;
  const int slot = i < a.f.n ? a.slot_of[i] : -1;
  if (i < a.f.n) {
    float dpos[3] = {0, 0, 0}, dls[3] = {0, 0, 0}, dq[4] = {0, 0, 0, 0}, dal = 0;
    if (slot >= 0) {
      const Cam c = load_cam(a.f.T44, a.f.proj);
      Fwd f;
      forward(a.f, c, i, f);
      float g[7] = {0, 0, 0, 0, 0, 0, 0}, gz = 0.0f;
      if (a.gpoints) {
#pragma unroll
        for (int k = 0; k < 7; ++k) g[k] = a.gpoints[int64_t(slot) * a.gpoints_stride + k];
      }
      if (a.gdepth) gz = a.gdepth[int64_t(slot) * a.gdepth_stride];
      if (a.gdepth_sq) gz += 2.0f * f.cam[2] * a.gdepth_sq[int64_t(slot) * a.gdepth_stride];
      // alpha = sigmoid(logit)
      dal = g[6] * f.alpha * (1.0f - f.alpha);
      // sigma = sqrt(lambda)
      float gl1 = g[4] * 0.5f / f.s1, gl2 = g[5] * 0.5f / f.s2;
      // axis = v / |v|
      const float dotag = f.ax * g[2] + f.ay * g[3];
      const float gvx = (g[2] - f.ax * dotag) / f.vn, gvy = (g[3] - f.ay * dotag) / f.vn;
      float gc00 = gvx, gc01 = gvy, gc11 = 0.0f;
      gl2 -= gvx;
      // lambda1,2 = (tr +- sg)/2
      float gtr = 0.5f * (gl1 + gl2);
      const float gsg = 0.5f * (gl1 - gl2);
      // sg = sqrt(max(gap,0)); at gap == 0 the reference's autodiff yields inf/NaN, we return 0
      const float ggap = (f.gap > 0.0f && f.sg > 0.0f) ? gsg * 0.5f / f.sg : 0.0f;
      gtr += 2.0f * f.tr * ggap;
      const float gdet = -4.0f * ggap;
      gc00 += gdet * f.c11 + gtr;
      gc11 += gdet * f.c00 + gtr;
      gc01 += -2.0f * f.c01 * gdet;
      // cov = m m^T
      float gN[2][3], gs[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float gm0 = 2.0f * gc00 * f.m[0][k] + gc01 * f.m[1][k];
        const float gm1 = 2.0f * gc11 * f.m[1][k] + gc01 * f.m[0][k];
        gs[k] = gm0 * f.N[0][k] + gm1 * f.N[1][k];
        gN[0][k] = gm0 * f.s[k];
        gN[1][k] = gm1 * f.s[k];
        dls[k] = gs[k] * f.s[k];  // s = exp(log_scale)
      }
      // N = J M3
      float gJ00 = 0, gJ02 = 0, gJ11 = 0, gJ12 = 0, gM3[3][3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        gJ00 += gN[0][k] * f.M3[0][k];
        gJ02 += gN[0][k] * f.M3[2][k];
        gJ11 += gN[1][k] * f.M3[1][k];
        gJ12 += gN[1][k] * f.M3[2][k];
        gM3[0][k] = f.J00 * gN[0][k];
        gM3[1][k] = f.J11 * gN[1][k];
        gM3[2][k] = f.J02 * gN[0][k] + f.J12 * gN[1][k];
      }
      // M3 = Tr R :  gR = Tr^T gM3 ; gTr = gM3 R^T
      float gR[3][3];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          gR[r][k] = c.T[0 * 4 + r] * gM3[0][k] + c.T[1 * 4 + r] * gM3[1][k] + c.T[2 * 4 + r] * gM3[2][k];
          if (CAMERA) gcam_acc[r * 4 + k] += gM3[r][0] * f.R[k][0] + gM3[r][1] * f.R[k][1] + gM3[r][2] * f.R[k][2];
        }
      // R = quat_to_mat(qn)  (generic.py:407-416)
      const float x = f.qn[0], y = f.qn[1], z = f.qn[2], w = f.qn[3];
      float gq[4];
      gq[0] = 2.0f * (y * gR[0][1] + z * gR[0][2] + y * gR[1][0] - 2.0f * x * gR[1][1] - w * gR[1][2] + z * gR[2][0] +
                      w * gR[2][1] - 2.0f * x * gR[2][2]);
      gq[1] = 2.0f * (-2.0f * y * gR[0][0] + x * gR[0][1] + w * gR[0][2] + x * gR[1][0] + z * gR[1][2] - w * gR[2][0] +
                      z * gR[2][1] - 2.0f * y * gR[2][2]);
      gq[2] = 2.0f * (-2.0f * z * gR[0][0] - w * gR[0][1] + x * gR[0][2] + w * gR[1][0] - 2.0f * z * gR[1][1] +
                      y * gR[1][2] + x * gR[2][0] + y * gR[2][1]);
      gq[3] = 2.0f * (-z * gR[0][1] + y * gR[0][2] + z * gR[1][0] - x * gR[1][2] - y * gR[2][0] + x * gR[2][1]);
      // qn = q / |q|
      const float dotq = f.qn[0] * gq[0] + f.qn[1] * gq[1] + f.qn[2] * gq[2] + f.qn[3] * gq[3];
#pragma unroll
      for (int k = 0; k < 4; ++k) dq[k] = (gq[k] - f.qn[k] * dotq) / f.qlen;
      // J and the projected mean
      const float zc = f.cam[2], iz = 1.0f / zc;
      float gzc = gz;
      float gfx = gJ00 * iz, gfy = gJ11 * iz;
      gzc += -gJ00 * c.fx * iz * iz - gJ11 * c.fy * iz * iz;
      gzc += gJ02 * (f.tx - c.cx) * iz * iz + gJ12 * (f.ty - c.cy) * iz * iz;
      float gcx = gJ02 * iz, gcy = gJ12 * iz;
      const float gu = g[0] + (f.in_x ? -gJ02 * iz : 0.0f);  // clamp: zero gradient outside the margin
      const float gv = g[1] + (f.in_y ? -gJ12 * iz : 0.0f);
      gfx += gu * f.cam[0] * iz;
      gfy += gv * f.cam[1] * iz;
      gcx += gu; gcy += gv;
      const float gcamx = gu * c.fx * iz, gcamy = gv * c.fy * iz;
      gzc += -gu * c.fx * f.cam[0] * iz * iz - gv * c.fy * f.cam[1] * iz * iz;
      // cam = Tr p + t
      const float gcamv[3] = {gcamx, gcamy, gzc};
      const float px = a.f.position[3 * i], py = a.f.position[3 * i + 1], pz = a.f.position[3 * i + 2];
#pragma unroll
      for (int k = 0; k < 3; ++k) dpos[k] = c.T[0 * 4 + k] * gcamv[0] + c.T[1 * 4 + k] * gcamv[1] + c.T[2 * 4 + k] * gcamv[2];
      if (CAMERA) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          gcam_acc[r * 4 + 0] += gcamv[r] * px;
          gcam_acc[r * 4 + 1] += gcamv[r] * py;
          gcam_acc[r * 4 + 2] += gcamv[r] * pz;
          gcam_acc[r * 4 + 3] += gcamv[r];
        }
        gcam_acc[12] = gfx; gcam_acc[13] = gfy; gcam_acc[14] = gcx; gcam_acc[15] = gcy;
      }
    }
#pragma unroll
    // gradients are written once and read by the optimizer later: keep them out of the caches the next frame's
    // gathers live in
    for (int k = 0; k < 3; ++k) __builtin_nontemporal_store(dpos[k], a.d_position + 3 * i + k);
#pragma unroll
    for (int k = 0; k < 3; ++k) __builtin_nontemporal_store(dls[k], a.d_log_scaling + 3 * i + k);
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(dq[k], a.d_rotation + 4 * i + k);
    __builtin_nontemporal_store(dal, a.d_alpha_logit + i);
  }
  if (CAMERA) {
    __shared__ float s_part[4][16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float tot = gs_wave_sum_to_lane63(gcam_acc[k]);
      if ((threadIdx.x & 63) == 63) s_part[threadIdx.x >> 6][k] = tot;
    }
    __syncthreads();
    if (threadIdx.x < 16)
      a.cam_partials[int64_t(blockIdx.x) * 16 + threadIdx.x] =
          s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
  }
}

// deterministic final reduction of the per-block camera partials (one block, fixed order)
__global__ __launch_bounds__(256) void cam_reduce_kernel(int num_blocks, const float* partials, float* dT44, float* dproj) {
  __shared__ double s[256];
  for (int k = 0; k < 16; ++k) {
    double acc = 0.0;
    for (int b = threadIdx.x; b < num_blocks; b += 256) acc += double(partials[int64_t(b) * 16 + k]);
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      if (k < 12) { if (dT44) dT44[k] = float(s[0]); }
      else if (dproj) dproj[k - 12] = float(s[0]);
    }
    __syncthreads();
  }
  if (threadIdx.x < 4 && dT44) dT44[12 + threadIdx.x] = 0.0f;
}

__global__ void camera_position_kernel(const float* T, float* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) camera_position(T, out);
}

int fill(ProjArgs& a, int64_t n, const float* position, const float* log_scaling, const float* rotation,
         const float* alpha_logit, const float* T, const float* proj, int width, int height, double near_p,
         double far_p, const GsRasterConfig* cfg) {
  if (int rc = gs_check_cfg(cfg)) return rc;
  GS_REQUIRE(n >= 0 && n < (int64_t(1) << 31), GS_ERR_INVALID_ARGUMENT, "projection: %lld gaussians", (long long)n);
  GS_REQUIRE(width > 0 && height > 0, GS_ERR_INVALID_ARGUMENT, "projection: image size %dx%d", width, height);
  GS_REQUIRE(n == 0 || (position && log_scaling && rotation && alpha_logit), GS_ERR_INVALID_ARGUMENT,
             "projection: NULL gaussian tensor");
  GS_REQUIRE(T && proj, GS_ERR_INVALID_ARGUMENT, "projection: NULL camera");
  a.position = position; a.log_scaling = log_scaling; a.rotation = rotation; a.alpha_logit = alpha_logit;
  a.T44 = T; a.proj = proj; a.n = n;
  a.width = float(width); a.height = float(height);
  a.near_p = float(near_p); a.far_p = float(far_p);
  a.inv_far = float(1.0 / far_p);
  a.ndc_denom = float(1.0 / near_p - 1.0 / far_p);
  a.clamp_margin = cfg->clamp_margin; a.blur_cov = cfg->blur_cov; a.alpha_thr = cfg->alpha_threshold;
  return GS_OK;
}

}  // namespace

extern "C" int64_t gs_project_scratch_bytes(int64_t n) {
  const int64_t nb = gs_div_up(n, 256);
  return gs_align_up(n * 32, 256) + gs_align_up((nb + 1) * 4, 256) * 2 + gs_cumsum_scratch_bytes(nb) + 256;
}

extern "C" int gs_project_fwd(int64_t n, const float* position, const float* log_scaling, const float* rotation,
                              const float* alpha_logit, const float* T_camera_world, const float* projection,
                              int32_t width, int32_t height, double near_plane, double far_plane,
                              const GsRasterConfig* cfg, float* points, float* depth, float* ndc_depth,
                              int64_t* indexes, int32_t* slot_of, int32_t* num_visible, float* depth_features,
                              int32_t depth_features_stride, float* camera_pos, void* scratch,
                              int64_t scratch_bytes, void* stream) {
  return gs_project_fwd_ex(n, position, log_scaling, rotation, alpha_logit, T_camera_world, projection, width, height,
                           near_plane, far_plane, cfg, points, depth, ndc_depth, indexes, slot_of, num_visible,
                           depth_features, depth_features_stride, camera_pos, scratch, scratch_bytes, nullptr, 0, nullptr,
                           stream);
}

// gs_project_fwd + (library-internal, used by gs_frame_fwd) zero_rows: a (V, zero_row_floats) buffer, 16-byte aligned
// rows, whose first V rows the compaction pass zero-fills; bin: the compaction pass also does the tile mapper's region
// binning (gs_common.h: GsMapBinPlan)
int gs_project_fwd_ex(int64_t n, const float* position, const float* log_scaling, const float* rotation,
                      const float* alpha_logit, const float* T_camera_world, const float* projection, int32_t width,
                      int32_t height, double near_plane, double far_plane, const GsRasterConfig* cfg, float* points,
                      float* depth, float* ndc_depth, int64_t* indexes, int32_t* slot_of, int32_t* num_visible,
                      float* depth_features, int32_t depth_features_stride, float* camera_pos, void* scratch,
                      int64_t scratch_bytes, float* zero_rows, int32_t zero_row_floats, const GsMapBinPlan* bin,
                      void* stream) {
  ProjArgs a;
  if (int rc = fill(a, n, position, log_scaling, rotation, alpha_logit, T_camera_world, projection, width, height,
                    near_plane, far_plane, cfg))
    return rc;
  GS_REQUIRE(near_plane > 0 && far_plane > near_plane, GS_ERR_INVALID_ARGUMENT, "projection: depth range");
  GS_REQUIRE(num_visible, GS_ERR_INVALID_ARGUMENT, "gs_project_fwd: num_visible is NULL");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n == 0) {
    if (hipMemsetAsync(num_visible, 0, 4, s) != hipSuccess) { gs_set_error("gs_project_fwd: memset failed"); return GS_ERR_LAUNCH; }
    if (camera_pos) return gs_camera_position(T_camera_world, camera_pos, stream);
    return GS_OK;
  }
  GS_REQUIRE(points && depth && ndc_depth && indexes && slot_of && scratch, GS_ERR_INVALID_ARGUMENT,
             "gs_project_fwd: NULL output");
  GS_REQUIRE(scratch_bytes >= gs_project_scratch_bytes(n), GS_ERR_SCRATCH_TOO_SMALL,
             "gs_project_fwd: scratch %lld < %lld", (long long)scratch_bytes, (long long)gs_project_scratch_bytes(n));
  const int nb = int(gs_div_up(n, 256));
  char* base = static_cast<char*>(scratch);
  float4* st_rows = reinterpret_cast<float4*>(base);
  int* counts = reinterpret_cast<int*>(base + gs_align_up(n * 32, 256));
  int* offsets = counts + gs_align_up(int64_t(nb + 1) * 4, 256) / 4;
  void* scan_scratch = offsets + gs_align_up(int64_t(nb + 1) * 4, 256) / 4;
#ifdef GS_PROJECT_ONE_PASS
  if (bin) {
    GsCompactArgs c;
    c.n = n; c.st_rows = nullptr; c.block_counts = nullptr;
    c.block_offsets = GS_PROJECT_ONE_PASS == 2 ? reinterpret_cast<const int*>(1) : nullptr;  // 2: count + recompute
    c.num_blocks = nb; c.inv_far = a.inv_far; c.ndc_denom = a.ndc_denom;
    c.points = points; c.depth = depth; c.ndc = ndc_depth; c.indexes = indexes; c.slot_of = slot_of;
    c.num_visible = num_visible; c.depth_feat = depth_features; c.depth_feat_stride = depth_features_stride;
    c.zero_rows = zero_rows; c.zero_row_v4 = zero_row_floats / 4;
    GS_REQUIRE(scratch_bytes >= gs_map_one_pass_scratch_bytes(n), GS_ERR_SCRATCH_TOO_SMALL, "gs_project_fwd: scratch");
    return gs_map_project_compact_bin(bin, &a, &c, camera_pos, scratch, stream);
  }
#endif
  int32_t* zero_words = nullptr;
  int32_t zero_count = 0;
  if (bin)
    if (int rc = gs_map_bin_counters(bin, n, &zero_words, &zero_count)) return rc;
  hipLaunchKernelGGL(project_kernel, dim3(nb), dim3(256), 0, s, a, st_rows, counts, camera_pos, zero_words, zero_count);
  GS_CHECK_LAUNCH("gs_project_fwd/project");
  // up to 16384 workgroups (4M Gaussians) each workgroup adds up the counts in front of it (<= 64 KB out of
  // L2); beyond that the quadratic read volume loses to a proper scan
  const bool self_offsets = nb <= 16384;
  if (!self_offsets)
    if (int rc = gs_full_cumsum_i32(nb, counts, offsets, scan_scratch, gs_cumsum_scratch_bytes(nb), s)) return rc;
  if (bin) {
    GsCompactArgs c;
    c.n = n; c.st_rows = st_rows; c.block_offsets = self_offsets ? nullptr : offsets; c.block_counts = counts;
    c.num_blocks = nb; c.inv_far = a.inv_far; c.ndc_denom = a.ndc_denom;
    c.points = points; c.depth = depth; c.ndc = ndc_depth; c.indexes = indexes; c.slot_of = slot_of;
    c.num_visible = num_visible; c.depth_feat = depth_features; c.depth_feat_stride = depth_features_stride;
    c.zero_rows = zero_rows; c.zero_row_v4 = zero_row_floats / 4;
    return gs_map_compact_bin(bin, &c, stream);
  }
  hipLaunchKernelGGL(compact_kernel, dim3(nb), dim3(256), 0, s, n, st_rows, self_offsets ? nullptr : offsets, counts, nb, a.inv_far, a.ndc_denom,
                     points, depth, ndc_depth, indexes, slot_of, num_visible, depth_features, depth_features_stride,
                     reinterpret_cast<float4*>(zero_rows), zero_row_floats / 4);
  GS_CHECK_LAUNCH("gs_project_fwd/compact");
  return GS_OK;
}

extern "C" int64_t gs_project_bwd_scratch_bytes(int64_t n) { return gs_align_up(gs_div_up(n, 256) * 64, 256) + 256; }

extern "C" int gs_project_bwd(int64_t n, int64_t v, const float* position, const float* log_scaling,
                              const float* rotation, const float* alpha_logit, const float* T_camera_world,
                              const float* projection, int32_t width, int32_t height, const GsRasterConfig* cfg,
                              const int32_t* slot_of, const float* grad_points, int32_t grad_points_stride,
                              const float* grad_depth, const float* grad_depth_sq, int32_t grad_depth_stride,
                              float* d_position, float* d_log_scaling, float* d_rotation, float* d_alpha_logit,
                              float* d_T_camera_world, float* d_projection, void* scratch, int64_t scratch_bytes,
                              void* stream) {
  (void)v;
  BwdArgs b;
  if (int rc = fill(b.f, n, position, log_scaling, rotation, alpha_logit, T_camera_world, projection, width, height,
                    1.0, 2.0, cfg))
    return rc;
  if (n == 0) return GS_OK;
  GS_REQUIRE(slot_of && d_position && d_log_scaling && d_rotation && d_alpha_logit, GS_ERR_INVALID_ARGUMENT,
             "gs_project_bwd: NULL buffer");
  const bool camera = d_T_camera_world != nullptr || d_projection != nullptr;
  const int nb = int(gs_div_up(n, 256));
  GS_REQUIRE(!camera || (scratch && scratch_bytes >= int64_t(nb) * 64), GS_ERR_SCRATCH_TOO_SMALL,
             "gs_project_bwd: camera gradients need %lld bytes of scratch", (long long)(int64_t(nb) * 64));
  b.slot_of = slot_of; b.gpoints = grad_points; b.gdepth = grad_depth; b.gdepth_sq = grad_depth_sq;
  b.gpoints_stride = grad_points_stride > 0 ? grad_points_stride : 7;
  b.gdepth_stride = grad_depth_stride > 0 ? grad_depth_stride : 1;
  b.d_position = d_position; b.d_log_scaling = d_log_scaling; b.d_rotation = d_rotation;
  b.d_alpha_logit = d_alpha_logit;
  b.cam_partials = static_cast<float*>(scratch);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (camera) {
    hipLaunchKernelGGL(project_bwd_kernel<true>, dim3(nb), dim3(256), 0, s, b);
    hipLaunchKernelGGL(cam_reduce_kernel, dim3(1), dim3(256), 0, s, nb, b.cam_partials, d_T_camera_world,
                       d_projection);
  } else {
    hipLaunchKernelGGL(project_bwd_kernel<false>, dim3(nb), dim3(256), 0, s, b);
  }
  GS_CHECK_LAUNCH("gs_project_bwd");
  return GS_OK;
}

extern "C" int gs_camera_position(const float* T_camera_world, float* camera_pos, void* stream) {
  GS_REQUIRE(T_camera_world && camera_pos, GS_ERR_INVALID_ARGUMENT, "gs_camera_position: NULL buffer");
  hipLaunchKernelGGL(camera_position_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), T_camera_world,
                     camera_pos);
  GS_CHECK_LAUNCH("gs_camera_position");
  return GS_OK;
}
