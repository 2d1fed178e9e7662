// project_math.h -- the projection's per-Gaussian device arithmetic (perspective/projection.py:32-80, math in
// taichi_lib/generic.py:96-158, :217-237, :419-427), shared by project.hip (the two-pass projection and the adjoint) and
// by mapper.hip's one-pass project + compact + bin kernel (frame calls).
//
// Both must produce the SAME BITS -- tests compare the frame calls with the composed operators bit for bit -- while
// mapper.hip is compiled with -ffp-contract=off (its grid query matches the CPU oracle op for op) and project.hip with
// hipcc's default (fast).  The functions below therefore pin their own contraction mode: `#pragma clang fp contract(fast)`
// at the head of each body, whatever the translation unit's flag says.
#pragma once
#include "gs_common.h"
#include "../../include/gs_detmath.h"

#ifdef __HIPCC__
namespace gs_proj {

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
#pragma clang fp contract(fast)
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
#pragma clang fp contract(fast)
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

// projection.py:60-67: does the Gaussian's alpha_threshold contour reach the image, between the depth planes?  (NaN from
// alpha < threshold fails every comparison)
__device__ __forceinline__ bool visible(const ProjArgs& a, const Fwd& f) {
#pragma clang fp contract(fast)
  const float gs = sqrtf(2.0f * gs_det_logf(f.alpha / a.alpha_thr));
  const float sx = f.s1 * gs, sy = f.s2 * gs;
  const float v1x = f.ax * sx, v1y = f.ay * sx, v2x = -f.ay * sy, v2y = f.ax * sy;
  const float ex = sqrtf(v1x * v1x + v2x * v2x), ey = sqrtf(v1y * v1y + v2y * v2y);
  const float z = f.cam[2];
  return (z > a.near_p) && (z < a.far_p) && (f.u + ex > 0.0f) && (f.v + ey > 0.0f) && (f.u - ex < a.width) &&
         (f.v - ey < a.height);
}

}  // namespace gs_proj
#endif
