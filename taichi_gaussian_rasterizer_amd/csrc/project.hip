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

namespace {

struct Cam {
  float T[12];  // rows 0..2 of T_camera_world
  float fx, fy, cx, cy;
};

__device__ __forceinline__ Cam load_cam(const float* T44, const float* proj) {
  Cam c;
#pragma unroll
  for (int i = 0; i < 12; ++i) c.T[i] = T44[i];
  c.fx = proj[0]; c.fy = proj[1]; c.cx = proj[2]; c.cy = proj[3];
  return c;
}

struct ProjArgs {
  const float* position;
  const float* log_scaling;
  const float* rotation;
  const float* alpha_logit;
  const float* T44;
  const float* proj;
  int64_t n;
  float width, height, near_p, far_p;
  float inv_far, ndc_denom;
  float clamp_margin, blur_cov, alpha_thr;
};

// Everything the forward produces plus the intermediates the adjoint needs.
struct Fwd {
  float qn[4], qlen, s[3];
  float cam[3];
  float u, v, tx, ty;
  bool in_x, in_y;
  float J00, J02, J11, J12;
  float R[3][3], M3[3][3], N[2][3], m[2][3];
  float c00, c01, c11, tr, gap, sg, l1, l2, vx, vy, vn;
  float ax, ay, s1, s2, alpha;
};

__device__ __forceinline__ void forward(const ProjArgs& a, const Cam& c, int64_t i, Fwd& f) {
  const float* q = a.rotation + 4 * i;
  f.qlen = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
#pragma unroll
  for (int k = 0; k < 4; ++k) f.qn[k] = q[k] / f.qlen;
#pragma unroll
  for (int k = 0; k < 3; ++k) f.s[k] = expf(a.log_scaling[3 * i + k]);
  const float px = a.position[3 * i], py = a.position[3 * i + 1], pz = a.position[3 * i + 2];
#pragma unroll
  for (int r = 0; r < 3; ++r) f.cam[r] = c.T[r * 4] * px + c.T[r * 4 + 1] * py + c.T[r * 4 + 2] * pz + c.T[r * 4 + 3];
  const float z = f.cam[2];
  f.u = (c.fx * f.cam[0]) / z + c.cx;
  f.v = (c.fy * f.cam[1]) / z + c.cy;
  const float lox = -a.width * a.clamp_margin, hix = (a.width - 1.0f) * (1.0f + a.clamp_margin);
  const float loy = -a.height * a.clamp_margin, hiy = (a.height - 1.0f) * (1.0f + a.clamp_margin);
  f.in_x = f.u >= lox && f.u <= hix;
  f.in_y = f.v >= loy && f.v <= hiy;
  f.tx = fminf(fmaxf(f.u, lox), hix);
  f.ty = fminf(fmaxf(f.v, loy), hiy);
  f.J00 = c.fx / z; f.J02 = -(f.tx - c.cx) / z;
  f.J11 = c.fy / z; f.J12 = -(f.ty - c.cy) / z;
  const float x = f.qn[0], y = f.qn[1], zq = f.qn[2], w = f.qn[3];
  const float x2 = x * x, y2 = y * y, z2 = zq * zq;
  f.R[0][0] = 1 - 2 * y2 - 2 * z2; f.R[0][1] = 2 * x * y - 2 * w * zq; f.R[0][2] = 2 * x * zq + 2 * w * y;
  f.R[1][0] = 2 * x * y + 2 * w * zq; f.R[1][1] = 1 - 2 * x2 - 2 * z2; f.R[1][2] = 2 * y * zq - 2 * w * x;
  f.R[2][0] = 2 * x * zq - 2 * w * y; f.R[2][1] = 2 * y * zq + 2 * w * x; f.R[2][2] = 1 - 2 * x2 - 2 * y2;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int k = 0; k < 3; ++k)
      f.M3[r][k] = c.T[r * 4] * f.R[0][k] + c.T[r * 4 + 1] * f.R[1][k] + c.T[r * 4 + 2] * f.R[2][k];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    f.N[0][k] = f.J00 * f.M3[0][k] + f.J02 * f.M3[2][k];
    f.N[1][k] = f.J11 * f.M3[1][k] + f.J12 * f.M3[2][k];
    f.m[0][k] = f.N[0][k] * f.s[k];
    f.m[1][k] = f.N[1][k] * f.s[k];
  }
  f.c00 = f.m[0][0] * f.m[0][0] + f.m[0][1] * f.m[0][1] + f.m[0][2] * f.m[0][2] + a.blur_cov;
  f.c01 = f.m[0][0] * f.m[1][0] + f.m[0][1] * f.m[1][1] + f.m[0][2] * f.m[1][2];
  f.c11 = f.m[1][0] * f.m[1][0] + f.m[1][1] * f.m[1][1] + f.m[1][2] * f.m[1][2] + a.blur_cov;
  f.tr = f.c00 + f.c11;
  const float det = f.c00 * f.c11 - f.c01 * f.c01;
  f.gap = f.tr * f.tr - 4.0f * det;
  f.sg = sqrtf(fmaxf(f.gap, 0.0f));
  f.l1 = (f.tr + f.sg) * 0.5f;
  f.l2 = (f.tr - f.sg) * 0.5f;
  f.vx = f.c00 - f.l2; f.vy = f.c01;
  f.vn = sqrtf(f.vx * f.vx + f.vy * f.vy);
  f.ax = f.vx / f.vn; f.ay = f.vy / f.vn;
  f.s1 = sqrtf(f.l1); f.s2 = sqrtf(f.l2);
  f.alpha = 1.0f / (1.0f + expf(-a.alpha_logit[i]));
}

// camera position = -R^-1 t of the (affine) camera matrix, on the device: CameraParams.camera_position
// (params.py:76-78) without the host round trip of a 4x4 torch.inverse
__device__ __forceinline__ void camera_position(const float* T, float* out) {
  const float a = T[0], b = T[1], c = T[2], d = T[4], e = T[5], f = T[6], g = T[8], h = T[9], i = T[10];
  const float tx = T[3], ty = T[7], tz = T[11];
  const float A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  const float det = a * A + b * B + c * C;
  const float inv = 1.0f / det;
  // inverse(R) rows
  const float r00 = A * inv, r01 = -(b * i - c * h) * inv, r02 = (b * f - c * e) * inv;
  const float r10 = B * inv, r11 = (a * i - c * g) * inv, r12 = -(a * f - c * d) * inv;
  const float r20 = C * inv, r21 = -(a * h - b * g) * inv, r22 = (a * e - b * d) * inv;
  out[0] = -(r00 * tx + r01 * ty + r02 * tz);
  out[1] = -(r10 * tx + r11 * ty + r12 * tz);
  out[2] = -(r20 * tx + r21 * ty + r22 * tz);
}

// pass 1: project everything, stage rows, count visible per block
__global__ __launch_bounds__(256) void project_kernel(ProjArgs a, float4* st_rows, int* block_counts, float* cam_out) {
  __shared__ int s_cnt[4];
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (cam_out && blockIdx.x == 0 && threadIdx.x == 0) camera_position(a.T44, cam_out);
  bool vis = false;
  if (i < a.n) {
    const Cam c = load_cam(a.T44, a.proj);
    Fwd f;
    forward(a, c, i, f);
    // projection.py:60-67 (NaN from alpha < threshold fails every comparison)
    const float gs = sqrtf(2.0f * gs_det_logf(f.alpha / a.alpha_thr));
    const float sx = f.s1 * gs, sy = f.s2 * gs;
    const float v1x = f.ax * sx, v1y = f.ay * sx, v2x = -f.ay * sy, v2y = f.ax * sy;
    const float ex = sqrtf(v1x * v1x + v2x * v2x), ey = sqrtf(v1y * v1y + v2y * v2y);
    const float z = f.cam[2];
    vis = (z > a.near_p) && (z < a.far_p) && (f.u + ex > 0.0f) && (f.v + ey > 0.0f) && (f.u - ex < a.width) &&
          (f.v - ey < a.height);
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
  hipLaunchKernelGGL(project_kernel, dim3(nb), dim3(256), 0, s, a, st_rows, counts, camera_pos);
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
